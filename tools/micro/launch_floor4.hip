// Dispatch-rate probe: S streams, each replaying a hipGraph of dependent small kernels, concurrently.  If the per-kernel time of a
// chain grows with the number of concurrently replaying chains, the chains share a serial resource in front of the CUs (command
// processor / dispatcher), which is what a four-chain pipelined step would then be bound by.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void small_k(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 64, per_graph = 200, reps = 20;
    const int SMAX = 6;
    hipStream_t st[SMAX]; float* buf[SMAX]; hipGraphExec_t ge[SMAX];
    for (int s = 0; s < SMAX; ++s) {
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        CK(hipMalloc(&buf[s], 4 * 256 * blocks)); CK(hipMemset(buf[s], 0, 4 * 256 * blocks));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < per_graph; ++i) hipLaunchKernelGGL(small_k, dim3(blocks), dim3(256), 0, st[s], buf[s], 256 * blocks);
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
    }
    for (int S = 1; S <= SMAX; ++S) {
        for (int s = 0; s < S; ++s) CK(hipGraphLaunch(ge[s], st[s]));          // warm
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (int s = 0; s < S; ++s) CK(hipGraphLaunch(ge[s], st[s]));
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%d blocks x 256 threads per kernel, %d concurrent chains: %.2f us per kernel of a chain, %.2f us per kernel overall\n", blocks, S,
               us / (reps * per_graph), us / (reps * per_graph * S));
    }
    return 0;
}
