// Kernel unit-test hooks exported through the C ABI (include/sva.h: sva_test_*).
#include "../../include/sva.h"
#include "kernels.h"
#include <vector>

using namespace sva;

extern "C" int sva_test_gemm(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C) {
    SVA_HIP(hipSetDevice(device));
    float *dA, *dW, *dB = nullptr, *dC;
    SVA_HIP(hipMalloc(&dA, sizeof(float) * (size_t)M * K));
    SVA_HIP(hipMalloc(&dW, sizeof(float) * (size_t)N * K));
    SVA_HIP(hipMalloc(&dC, sizeof(float) * (size_t)M * N));
    SVA_HIP(hipMemcpy(dA, A, sizeof(float) * (size_t)M * K, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dW, W, sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice));
    if (bias) {
        SVA_HIP(hipMalloc(&dB, sizeof(float) * N));
        SVA_HIP(hipMemcpy(dB, bias, sizeof(float) * N, hipMemcpyHostToDevice));
    }
    ConvGemm g;
    g.A = dA; g.a_bstride = (long)M * K; g.lda = K; g.T = M; g.M = M; g.Cin = K; g.taps = 1;
    g.W = dW; g.N = N; g.bias = dB; g.C = dC; g.c_bstride = (long)M * N; g.ldc = N;
    int rc = launch_conv_gemm(g, 0);
    if (rc) return rc;
    SVA_HIP(hipDeviceSynchronize());
    SVA_HIP(hipMemcpy(C, dC, sizeof(float) * (size_t)M * N, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dW); hipFree(dC); if (dB) hipFree(dB);
    return 0;
}
