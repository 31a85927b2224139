// membench.hip -- random-access rates of MI355X as a function of the footprint (design input for the hash / gather kernels).
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
// For windows of 1 MiB .. 4 GiB: random 8-byte loads (plain and L1-bypassing), 16-byte loads, 4/8-byte stores, agent-scope
// u32 atomic adds (no return), u64 atomic adds with return, u64 CAS; ops/s chip-wide with 4 independent ops in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
enum { LD8, LD8_SC1, LD16, ST4, ST8, AT32, AT64R, CAS64, NKIND };
const char *kname[] = {"load8", "load8_sc1", "load16", "store4", "store8", "atomic32_noret", "atomic64_ret", "cas64"};
template <int KIND>
__global__ void __launch_bounds__(256) k(u64 *buf, u64 mask8 /* window in 8-byte words - 1 */, int iters, u64 *sink) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 acc = 0;
    for (int it = 0; it < iters; it++) {
        u64 i0 = mix(tid * 4 + 0 + (u64)it * 0x9E3779B97F4A7C15ull) & mask8, i1 = mix(tid * 4 + 1 + (u64)it * 0x9E3779B97F4A7C15ull) & mask8,
            i2 = mix(tid * 4 + 2 + (u64)it * 0x9E3779B97F4A7C15ull) & mask8, i3 = mix(tid * 4 + 3 + (u64)it * 0x9E3779B97F4A7C15ull) & mask8;
        if (KIND == LD8) { acc += buf[i0] + buf[i1] + buf[i2] + buf[i3]; }
        else if (KIND == LD8_SC1) {
            acc += __hip_atomic_load(&buf[i0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&buf[i1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                   __hip_atomic_load(&buf[i2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&buf[i3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (KIND == LD16) {
            const uint4 *b = (const uint4 *)buf;
            uint4 a = b[i0 >> 1], c = b[i1 >> 1], d = b[i2 >> 1], e = b[i3 >> 1];
            acc += a.x + a.w + c.x + c.w + d.x + d.w + e.x + e.w;
        } else if (KIND == ST4) { u32 *b = (u32 *)buf; b[i0 * 2] = (u32)it; b[i1 * 2] = (u32)it; b[i2 * 2] = (u32)it; b[i3 * 2] = (u32)it; }
        else if (KIND == ST8) { buf[i0] = it; buf[i1] = it; buf[i2] = it; buf[i3] = it; }
        else if (KIND == AT32) { u32 *b = (u32 *)buf; atomicAdd(&b[i0 * 2], 1u); atomicAdd(&b[i1 * 2], 1u); atomicAdd(&b[i2 * 2], 1u); atomicAdd(&b[i3 * 2], 1u); }
        else if (KIND == AT64R) {
            acc += atomicAdd((unsigned long long *)&buf[i0], 1ull) + atomicAdd((unsigned long long *)&buf[i1], 1ull) + atomicAdd((unsigned long long *)&buf[i2], 1ull) +
                   atomicAdd((unsigned long long *)&buf[i3], 1ull);
        } else if (KIND == CAS64) {
            acc += atomicCAS((unsigned long long *)&buf[i0], 0ull, 5ull) + atomicCAS((unsigned long long *)&buf[i1], 0ull, 5ull) + atomicCAS((unsigned long long *)&buf[i2], 0ull, 5ull) +
                   atomicCAS((unsigned long long *)&buf[i3], 0ull, 5ull);
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// dependent chain: one load feeds the next address (latency-bound walks), 1 and 4 chains per lane
template <int CH>
__global__ void __launch_bounds__(256) kchain(const u64 *buf, u64 mask8, int iters, u64 *sink) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 p[CH];
    for (int c = 0; c < CH; c++) p[c] = mix(tid * CH + c) & mask8;
    for (int it = 0; it < iters; it++)
        for (int c = 0; c < CH; c++) p[c] = mix(buf[p[c]] + p[c] + it) & mask8;
    u64 acc = 0;
    for (int c = 0; c < CH; c++) acc += p[c];
    if (acc == 0x1234567) sink[0] = acc;
}
template <int KIND> double run(u64 *buf, u64 words, int blocks, int iters, u64 *sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, buf, words - 1, 2, sink);      // warm
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, buf, words - 1, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return (double)blocks * 256 * 4 * iters / (ms * 1e-3) / 1e9;
}
template <int CH> double runchain(u64 *buf, u64 words, int blocks, int iters, u64 *sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kchain<CH>, dim3(blocks), dim3(256), 0, 0, buf, words - 1, 2, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kchain<CH>, dim3(blocks), dim3(256), 0, 0, buf, words - 1, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return (double)blocks * 256 * CH * iters / (ms * 1e-3) / 1e9;
}
int main() {
    u64 *buf, *sink;
    const u64 maxb = 4ull << 30;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, maxb));
    const int blocks = 256 * 8, iters = 64;
    printf("%-16s", "window");
    for (int kd = 0; kd < NKIND; kd++) printf(" %14s", kname[kd]);
    printf(" %10s %10s   (G ops/s, chip-wide; %d lanes x 4 ops in flight)\n", "chain1", "chain4", blocks * 256);
    for (u64 mb : {1ull, 2ull, 4ull, 16ull, 32ull, 64ull, 128ull, 256ull, 512ull, 1024ull, 4096ull}) {
        u64 words = (mb << 20) / 8;
        printf("%6llu MiB      ", (unsigned long long)mb);
        printf(" %14.2f", run<LD8>(buf, words, blocks, iters, sink));
        printf(" %14.2f", run<LD8_SC1>(buf, words, blocks, iters, sink));
        printf(" %14.2f", run<LD16>(buf, words, blocks, iters, sink));
        printf(" %14.2f", run<ST4>(buf, words, blocks, iters, sink));
        printf(" %14.2f", run<ST8>(buf, words, blocks, iters, sink));
        printf(" %14.2f", run<AT32>(buf, words, blocks, iters, sink));
        printf(" %14.2f", run<AT64R>(buf, words, blocks, iters, sink));
        CK(hipMemset(buf, 0, mb << 20));
        printf(" %14.2f", run<CAS64>(buf, words, blocks, iters, sink));
        printf(" %10.2f %10.2f\n", runchain<1>(buf, words, blocks, 32, sink), runchain<4>(buf, words, blocks, 32, sink));
        fflush(stdout);
    }
    return 0;
}
