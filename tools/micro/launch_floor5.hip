// Which stream subsets avoid the dispatch cliff of four concurrently replaying chains?  (8 streams created in order; subsets by index)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void small_k(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int blocks = 32, per_graph = 200, reps = 20, NS = 8;
    hipStream_t st[NS]; float* buf[NS]; hipGraphExec_t ge[NS];
    for (int s = 0; s < NS; ++s) {
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        CK(hipMalloc(&buf[s], 4 * 256 * blocks)); CK(hipMemset(buf[s], 0, 4 * 256 * blocks));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < per_graph; ++i) hipLaunchKernelGGL(small_k, dim3(blocks), dim3(256), 0, st[s], buf[s], 256 * blocks);
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
    }
    const std::vector<std::vector<int>> sets = {{0, 1, 2}, {0, 1, 2, 3}, {0, 2, 4, 6}, {0, 1, 4, 5}, {0, 3, 5, 6}, {1, 2, 3, 4}, {4, 5, 6, 7}, {0, 4}, {0, 1, 2, 3, 4, 5, 6, 7}};
    for (const auto& set : sets) {
        for (int s : set) CK(hipGraphLaunch(ge[s], st[s]));
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (int s : set) CK(hipGraphLaunch(ge[s], st[s]));
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("streams {");
        for (int s : set) printf(" %d", s);
        printf(" }: %.2f us per kernel of a chain\n", us / (reps * per_graph));
    }
    return 0;
}
