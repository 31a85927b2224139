// launch_floor.hip -- what a launch and a host synchronisation cost on this box, whatever the kernel does (the floor of
// small inputs -- BASELINE configs[1], 101 MB: 755 launches and 101 synchronisations per build).
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor tools/launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nothing(unsigned *p) { if (p && threadIdx.x == 1024) *p = 1; }
__global__ void k_touch(unsigned *p, unsigned n) { unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1; }
int main() {
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *d; (void)hipMalloc(&d, 1 << 22);
    unsigned *pinned; (void)hipHostMalloc((void **)&pinned, 4096, hipHostMallocMapped);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int N = 2000;
    for (int variant = 0; variant < 3; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            (void)hipStreamSynchronize(s);
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipEventRecord(a, s);
            for (int i = 0; i < N; i++) {
                if (variant == 0) hipLaunchKernelGGL(k_nothing, dim3(1), dim3(256), 0, s, (unsigned *)nullptr);
                else if (variant == 1) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, d, 1u << 20);
                else { hipLaunchKernelGGL(k_nothing, dim3(1), dim3(256), 0, s, (unsigned *)nullptr); (void)hipStreamSynchronize(s); }
            }
            (void)hipEventRecord(b, s);
            (void)hipStreamSynchronize(s);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
            if (rep) printf("%-58s %7.2f us per launch on the stream, %7.2f us of wall\n",
                            variant == 0 ? "empty kernel, back to back" : (variant == 1 ? "1 M-element kernel (4 MB read + written), back to back" : "empty kernel + host synchronisation"),
                            ms * 1e3 / N, wall * 1e3 / N);
        }
    }
    return 0;
}
