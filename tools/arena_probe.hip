// Lease probe: which growth patterns of a reserved address range does this HIP stack accept?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/arena_probe tools/arena_probe.hip && /tmp/arena_probe
// (the engine's arena maps physical memory behind a growing high-water mark; a 10 GiB first step followed by any further
// step was refused by hipMemSetAccess -- "invalid argument" -- while 1 GiB steps worked)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static const size_t GiB = (size_t)1 << 30;
struct Range {
    char *base = nullptr; size_t va = 0, size = 0;
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks;
};
static bool reserve(Range &r, size_t va) {
    void *b = nullptr;
    if (hipMemAddressReserve(&b, va, GiB, nullptr, 0) != hipSuccess) return false;
    r.base = (char *)b; r.va = va; return true;
}
// mode 0: access set on the new piece only; mode 1: access set on [base, new end)
static const char *grow(Range &r, size_t want, int mode) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, want, &prop, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); return "create"; }
    e = hipMemMap(r.base + r.size, want, 0, h, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return "map"; }
    hipMemAccessDesc ad = {};
    ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    e = mode == 0 ? hipMemSetAccess(r.base + r.size, want, &ad, 1) : hipMemSetAccess(r.base, r.size + want, &ad, 1);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipMemUnmap(r.base + r.size, want); (void)hipMemRelease(h); return "set_access"; }
    r.chunks.emplace_back(h, want); r.size += want;
    return nullptr;
}
static void drop(Range &r) {
    size_t off = 0;
    for (auto &c : r.chunks) { (void)hipMemUnmap(r.base + off, c.second); (void)hipMemRelease(c.first); off += c.second; }
    (void)hipMemAddressFree(r.base, r.va);
    r = Range();
}
__global__ void touch(unsigned long long *p, size_t n, unsigned long long v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i * 512] = v + i;
}
__global__ void check(const unsigned long long *p, size_t n, unsigned long long v, unsigned *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i * 512] != v + i) atomicAdd(bad, 1u);
}
static unsigned verify(Range &r) {
    size_t n = r.size / 4096;
    unsigned *bad; (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
    touch<<<(unsigned)((n + 255) / 256), 256>>>((unsigned long long *)r.base, n, 77);
    check<<<(unsigned)((n + 255) / 256), 256>>>((const unsigned long long *)r.base, n, 77, bad);
    unsigned h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); (void)hipFree(bad);
    if (hipDeviceSynchronize() != hipSuccess) { printf("    kernel fault: %s\n", hipGetErrorString(hipGetLastError())); return ~0u; }
    return h;
}
static void pattern(const char *name, std::vector<size_t> steps_gib, int mode, size_t piece_gib /*0: one piece per step*/) {
    Range r;
    if (!reserve(r, 288 * GiB)) { printf("%s: reserve failed\n", name); return; }
    auto t0 = std::chrono::steady_clock::now();
    const char *fail = nullptr; size_t at = 0;
    for (size_t s : steps_gib) {
        size_t left = s;
        while (left && !fail) {
            size_t p = piece_gib && left > piece_gib ? piece_gib : left;
            fail = grow(r, p * GiB, mode);
            if (fail) at = r.size / GiB;
            left -= p;
        }
        if (fail) break;
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (fail) printf("%-44s mode %d piece %zu: FAILED in %s at %zu GiB mapped\n", name, mode, piece_gib, fail, at);
    else printf("%-44s mode %d piece %zu: ok, %zu GiB in %.3f s, %u bad words\n", name, mode, piece_gib, r.size / GiB, dt, verify(r));
    drop(r);
}
int main(int argc, char **argv) {
    // one pattern per process (unmapping and mapping again in ONE process leaves stale translations behind on this stack:
    // "bad words" in every pattern that follows a drop())
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    (void)hipSetDevice(0);
    int k = 0;
#define PAT(...) do { if (which < 0 || which == k) pattern(__VA_ARGS__); k++; } while (0)
    PAT("1 x 8", {1, 1, 1, 1, 1, 1, 1, 1}, 0, 0);
    PAT("10 then 1", {10, 1}, 0, 0);
    PAT("10 then 1 (access over the whole range)", {10, 1}, 1, 0);
    PAT("4 then 1", {4, 1}, 0, 0);
    PAT("5 then 1", {5, 1}, 0, 0);
    PAT("8 then 1", {8, 1}, 0, 0);
    PAT("1 then 10 then 1", {1, 10, 1}, 0, 0);
    PAT("10 then 1, pieces of 1 GiB", {10, 1}, 0, 1);
    PAT("10 then 1, pieces of 2 GiB", {10, 1}, 0, 2);
    PAT("10 then 1, pieces of 4 GiB", {10, 1}, 0, 4);
    PAT("10, 12, 16, 16, 20, 30 pieces of 1 GiB", {10, 12, 16, 16, 20, 30}, 0, 1);
    PAT("10, 12, 16, 16, 20, 30 pieces of 4 GiB", {10, 12, 16, 16, 20, 30}, 0, 4);
    PAT("10, 12, 16, 16, 20, 30 one piece each", {10, 12, 16, 16, 20, 30}, 0, 0);
    PAT("1, 2, 4, 8, 16, 32 one piece each", {1, 2, 4, 8, 16, 32}, 0, 0);
    PAT("2 then 2 then 2", {2, 2, 2}, 0, 0);
    PAT("3 then 1", {3, 1}, 0, 0);
    return 0;
}
