// Lease probe: what does device memory cost to obtain?  (sizing of the slab reservation and of the CLI's upload path)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_probe tools/alloc_probe.hip && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); } } while (0)

__global__ void touch(char *p, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096;
    if (i < n) p[i] = 1;
}

int main() {
    CK(hipSetDevice(0));
    CK(hipFree(0));
    for (size_t gb : {1, 4, 16, 64, 128, 200}) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, gb << 30);
        double t1 = now();
        if (e != hipSuccess) { printf("hipMalloc %zu GB failed\n", gb); (void)hipGetLastError(); continue; }
        touch<<<(unsigned)(((gb << 30) / 4096 + 255) / 256), 256>>>((char *)p, gb << 30);
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        double t3 = now();
        printf("hipMalloc %4zu GB: alloc %.3f s, first touch (1 B / 4 KiB) %.3f s, free %.3f s\n", gb, t1 - t0, t2 - t1, t3 - t2);
    }
    // virtual memory management: reserve a large VA range, back it chunk by chunk
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("VMM granularity %zu\n", gran);
        const size_t total = (size_t)256 << 30, chunk = (size_t)2 << 30;
        void *va = nullptr;
        double t0 = now();
        hipError_t e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
        printf("hipMemAddressReserve 256 GB: %s %.4f s\n", hipGetErrorString(e), now() - t0);
        if (e == hipSuccess) {
            std::vector<hipMemGenericAllocationHandle_t> hs;
            double tc = 0, tm = 0;
            for (int k = 0; k < 32; k++) {
                hipMemGenericAllocationHandle_t h;
                double a = now();
                if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %d\n", k); break; }
                double b = now();
                CK(hipMemMap((char *)va + k * chunk, chunk, 0, h, 0));
                hipMemAccessDesc ad = {};
                ad.location = prop.location;
                ad.flags = hipMemAccessFlagsProtReadWrite;
                CK(hipMemSetAccess((char *)va + k * chunk, chunk, &ad, 1));
                double c = now();
                tc += b - a; tm += c - b;
                hs.push_back(h);
            }
            printf("VMM: %zu chunks of 2 GB: create %.3f s, map+access %.3f s\n", hs.size(), tc, tm);
            double a = now();
            touch<<<(unsigned)((hs.size() * chunk / 4096 + 255) / 256), 256>>>((char *)va, hs.size() * chunk);
            CK(hipDeviceSynchronize());
            printf("VMM: touch %.3f s\n", now() - a);
            for (size_t k = 0; k < hs.size(); k++) { CK(hipMemUnmap((char *)va + k * chunk, chunk)); CK(hipMemRelease(hs[k])); }
            CK(hipMemAddressFree(va, total));
        }
    }
    // pinned host memory + copies
    {
        const size_t sz = (size_t)1 << 30;
        void *h = nullptr, *d = nullptr;
        double t0 = now();
        CK(hipHostMalloc(&h, sz, hipHostMallocDefault));
        double t1 = now();
        memset(h, 1, sz);
        double t2 = now();
        CK(hipMalloc(&d, sz));
        double t3 = now();
        CK(hipMemcpy(d, h, sz, hipMemcpyHostToDevice));
        double t4 = now();
        CK(hipMemcpy(h, d, sz, hipMemcpyDeviceToHost));
        double t5 = now();
        std::vector<char> pg(sz);
        memset(pg.data(), 2, sz);
        double t6 = now();
        CK(hipMemcpy(d, pg.data(), sz, hipMemcpyHostToDevice));
        double t7 = now();
        printf("hipHostMalloc 1 GB %.3f s, memset %.3f s, H2D pinned %.1f GB/s, D2H pinned %.1f GB/s, H2D pageable %.1f GB/s\n", t1 - t0,
               t2 - t1, 1.0737 / (t4 - t3), 1.0737 / (t5 - t4), 1.0737 / (t7 - t6));
        CK(hipHostFree(h));
        CK(hipFree(d));
    }
    return 0;
}
