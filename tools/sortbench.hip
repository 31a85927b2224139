// sortbench.hip -- the radix-sort primitives of prim_hip.hpp alone, on synthetic keys (kernel work of the round: the
// in-wave ranking shared by k_rs_scatter / k_xs_scatter / k_rs_unscatter, digit plans, record widths).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I grlbwt_amd/csrc -o sortbench tools/sortbench.hip && ./sortbench [n] [kbits] [reps] [skew]
// Sorts n (u64 key, u32 value) pairs and n u64 keys by key bits [0, kbits); prints ms per sort, ms per launch site, the
// rate of a pass in its own bytes, and checks order + stability + that the values are a permutation.
// skew = 0: uniform keys; skew = s > 0: every digit is the AND of s+1 uniform draws (few bins take most keys: the first
// symbols of a dictionary's suffixes are like that).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <type_traits>
#include <algorithm>
#include "prim_hip.hpp"
using prim::u32; using prim::u64;

__device__ __forceinline__ u64 mix(u64 z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ void k_fill(u64 *k, u32 *v, u64 n, int kbits, int skew) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 x = mix(i);
    for (int s = 0; s < skew; s++) x &= mix(i * 7919u + (u64)s + 1u);
    k[i] = kbits >= 64 ? x : (x & ((1ull << kbits) - 1ull));
    if (v) v[i] = (u32)i;
}
// bad[0] += out-of-order neighbours, bad[1] += unstable neighbours (equal keys, values descending), bad[2] ^= / += checksums
__global__ void k_check(const u64 *k, const u32 *v, u64 n, u64 mask, unsigned long long *bad) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i + 1 < n) {
        u64 a = k[i] & mask, b = k[i + 1] & mask;
        if (a > b) atomicAdd(&bad[0], 1ull);
        if (v && a == b && v[i] > v[i + 1]) atomicAdd(&bad[1], 1ull);
    }
    if (v) { if (k[i] != (mix(v[i]) & (mask == ~0ull ? mask : mask)) && false) atomicAdd(&bad[3], 1ull); atomicAdd(&bad[2], (unsigned long long)v[i]); }
}
__global__ void k_check_pairs(const u64 *k, const u32 *v, u64 n, int kbits, int skew, unsigned long long *bad) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 j = v[i];
    u64 x = mix(j);
    for (int s = 0; s < skew; s++) x &= mix(j * 7919u + (u64)s + 1u);
    x = kbits >= 64 ? x : (x & ((1ull << kbits) - 1ull));
    if (x != k[i]) atomicAdd(&bad[3], 1ull);
}

static void report(const char *what, double ms, u64 n, int passes, int bytes_per_pass) {
    printf("%-28s %9.3f ms   %d passes   %7.1f GB/s per pass (read + write + histogram read)\n", what, ms, passes, passes * (double)n * bytes_per_pass / ms / 1e6);
}

int main(int argc, char **argv) {
    u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : (u64)256 << 20;
    int kbits = argc > 2 ? atoi(argv[2]) : 56, reps = argc > 3 ? atoi(argv[3]) : 3, skew = argc > 4 ? atoi(argv[4]) : 0;
    try {
        prim::init(0);
        u64 *ka = (u64 *)prim::dev_alloc(n * 8), *kb = (u64 *)prim::dev_alloc(n * 8);
        u32 *va = (u32 *)prim::dev_alloc(n * 4), *vb = (u32 *)prim::dev_alloc(n * 4);
        unsigned long long *bad = (unsigned long long *)prim::dev_alloc(64);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const unsigned grid = (unsigned)((n + 255) / 256);
        const u64 mask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
        int widths[16];
        const int passes = prim::rs_plan(kbits, widths);
        printf("n = %llu, kbits = %d, passes = %d, skew = %d\n", (unsigned long long)n, kbits, passes, skew);
        for (int mode = 0; mode < 2; mode++) {       // 0: pairs, 1: keys only
            double best = 1e30;
            for (int r = 0; r < reps + 1; r++) {
                hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, prim::rt().stream, ka, mode ? nullptr : va, n, kbits, skew);
                prim::rt().profile = (r == reps);
                (void)hipEventRecord(e0, prim::rt().stream);
                int res = mode ? prim::sort_keys<u64>(ka, kb, n, 0, kbits, "sort") : prim::sort_pairs<u64, u32>(ka, va, kb, vb, n, 0, kbits, "sort");
                (void)hipEventRecord(e1, prim::rt().stream);
                prim::sync();
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r > 0 && r < reps && ms < best) best = ms;
                if (r == reps) {
                    prim::dev_memset(bad, 0, 64);
                    const u64 *ks = res ? kb : ka; const u32 *vs = mode ? nullptr : (res ? vb : va);
                    hipLaunchKernelGGL(k_check, dim3(grid), dim3(256), 0, prim::rt().stream, ks, vs, n, mask, bad);
                    if (!mode) hipLaunchKernelGGL(k_check_pairs, dim3(grid), dim3(256), 0, prim::rt().stream, ks, vs, n, kbits, skew, bad);
                    unsigned long long h[8];
                    prim::d2h(h, bad, 64);
                    const unsigned long long want = (unsigned long long)n * (n - 1) / 2;
                    const bool ok = h[0] == 0 && h[1] == 0 && h[3] == 0 && (mode || h[2] == want);
                    printf("  check %s: out of order %llu, unstable %llu, wrong pairs %llu, value sum %s\n", ok ? "OK" : "FAILED", h[0], h[1], h[3], (mode || h[2] == want) ? "ok" : "WRONG");
                    if (!ok) return 1;
                }
            }
            report(mode ? "sort_keys<u64>" : "sort_pairs<u64,u32>", best, n, passes, mode ? 24 : 32);
            for (auto &kv : prim::rt().prof) if (kv.second.ms > 0) printf("    %-24s %4llu launches %9.3f ms\n", kv.first.c_str(), (unsigned long long)kv.second.launches, kv.second.ms);
            prim::rt().prof.clear();
        }
    } catch (const std::exception &e) { printf("error: %s\n", e.what()); return 2; }
    return 0;
}
