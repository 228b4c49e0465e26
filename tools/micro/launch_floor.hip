// Launch-floor probe: dependent trivial kernels on one stream, eager vs hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny256(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    int* d; float* f;
    CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    CK(hipMalloc(&f, 4 * 256 * 256)); CK(hipMemset(f, 0, 4 * 256 * 256));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager tiny(1 block): %.2f us/launch\n", ms * 1e3 / N);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny256, dim3(256), dim3(256), 0, st, f, 65536);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager tiny256(256 blocks): %.2f us/launch\n", ms * 1e3 / N);
    }
    // graph of 500 dependent kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(tiny256, dim3(256), dim3(256), 0, st, f, 65536);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph tiny256: %.2f us/kernel\n", ms * 1e3 / 2000);
    }
    return 0;
}
