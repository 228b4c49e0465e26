// Host-side cost of hipLaunchKernelGGL as a function of the kernel-argument size (the conv-GEMM descriptors are
// passed by value): enqueue N launches without waiting, time the host loop only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int N> struct Arg { float v[N]; };
template <int N> __global__ void k(Arg<N> a, float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a.v[0] + a.v[N - 1]; }
template <int N> int run(hipStream_t st, float* d, int grid) {
    Arg<N> a; for (int i = 0; i < N; ++i) a.v[i] = (float)i;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, st, a, d);
    CK(hipStreamSynchronize(st));
    const int n = 4000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, st, a, d);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    printf("arg %4d B grid %4d: host enqueue %.2f us/launch, total incl. drain %.2f us/launch\n", N * 4, grid,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    return 0;
}
int main() {
    float* d; CK(hipMalloc(&d, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int grid : {1, 256}) {
        if (run<4>(st, d, grid)) return 1;
        if (run<50>(st, d, grid)) return 1;
        if (run<160>(st, d, grid)) return 1;
        if (run<512>(st, d, grid)) return 1;
    }
    return 0;
}
