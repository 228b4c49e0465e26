// Kernel unit-test hooks exported through the C ABI (include/sva.h: sva_test_*).
#include "../../include/sva.h"
#include "kernels.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

using namespace sva;

#define SVA_TRY(expr)            \
    do {                         \
        const int _rc = (expr);  \
        if (_rc) return _rc;     \
    } while (0)

static int test_gemm_impl(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C, const int* choice);
extern "C" int sva_test_gemm(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C) {
    return test_gemm_impl(device, M, N, K, A, W, bias, C, nullptr);
}
extern "C" int sva_test_gemm_choice(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C, int kind,
                                    int a, int b, int c) {
    const int ch[4] = {kind, a, b, c};
    return test_gemm_impl(device, M, N, K, A, W, bias, C, ch);
}
static int test_gemm_impl(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C, const int* choice) {
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
    // kind 2: small-M kernel with a grid-level K split, c = column tiles + 16 * splits
    int rc = !choice ? launch_conv_gemm(g, 0)
             : choice[0] == 3 ? launch_conv_gemm_choice(g, 0, 2, choice[1], 0, 0)       // the register-staged pipelined kernel (gemm_pipe.hip), tile variant choice[1]
             : choice[0] == 4 ? launch_conv_gemm_choice(g, 0, 4, choice[1], 0, 0)       // the split-bf16 kernel, tile variant choice[1]
             : choice[0] == 2 ? launch_conv_gemm_choice_z(g, 0, choice[1], choice[2], choice[3] & 15, choice[3] >> 4)
                              : launch_conv_gemm_choice(g, 0, choice[0], choice[1], choice[2], choice[3]);
    if (!rc && choice && choice[0] == 2) rc = launch_conv_gemm_choice_z(g, 0, choice[1], choice[2], choice[3] & 15, choice[3] >> 4);   // twice: the counters re-arm
    if (rc) return rc;
    SVA_HIP(hipDeviceSynchronize());
    SVA_HIP(hipMemcpy(C, dC, sizeof(float) * (size_t)M * N, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); if (dB) (void)hipFree(dB);
    return 0;
}

// fp16-weight GEMM of the batched fp16 AR (gemm_f16w.hip): W is rounded to fp16 here; mode bits: 1 = RMSNorm prologue (rms_w [K]),
// 2 = residual add (res [M][N]), 4 = SwiGLU over 16-row interleaved (gate, up) weights (C is [M][N/2]).  iters > 0 also times it.
extern "C" int sva_test_gemm_f16w(int device, int M, int N, int K, const float* A, const float* W, const float* bias, const float* rms_w,
                                  const float* res, int mode, float* C, int iters, float* out_us) {
    SVA_HIP(hipSetDevice(device));
    const int NC = (mode & 4) ? N / 2 : N;
    float *dA, *dC, *dB = nullptr, *dN = nullptr, *dR = nullptr;
    uint16_t* dW;
    std::vector<uint16_t> hb((size_t)N * K);
    for (size_t i = 0; i < hb.size(); ++i) { const _Float16 h = (_Float16)W[i]; memcpy(&hb[i], &h, 2); }
    SVA_HIP(hipMalloc(&dA, sizeof(float) * (size_t)M * K));
    SVA_HIP(hipMalloc(&dW, 2 * (size_t)N * K));
    SVA_HIP(hipMalloc(&dC, sizeof(float) * (size_t)M * NC));
    SVA_HIP(hipMemcpy(dA, A, sizeof(float) * (size_t)M * K, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dW, hb.data(), 2 * (size_t)N * K, hipMemcpyHostToDevice));
    if (bias) { SVA_HIP(hipMalloc(&dB, sizeof(float) * N)); SVA_HIP(hipMemcpy(dB, bias, sizeof(float) * N, hipMemcpyHostToDevice)); }
    if (mode & 1) { SVA_HIP(hipMalloc(&dN, sizeof(float) * K)); SVA_HIP(hipMemcpy(dN, rms_w, sizeof(float) * K, hipMemcpyHostToDevice)); }
    if (mode & 2) { SVA_HIP(hipMalloc(&dR, sizeof(float) * (size_t)M * N)); SVA_HIP(hipMemcpy(dR, res, sizeof(float) * (size_t)M * N, hipMemcpyHostToDevice)); }
    ConvGemm g;
    g.A = dA; g.a_bstride = (long)M * K; g.lda = K; g.T = M; g.M = M; g.Cin = K; g.taps = 1;
    g.Wh = dW; g.N = N; g.bias = dB; g.C = dC; g.c_bstride = (long)M * NC; g.ldc = NC;
    g.rms_w = dN; g.res = dR; g.r_bstride = (long)M * N; g.ldr = N; g.w13 = (mode & 4) ? 1 : 0;
    SVA_CHECK(f16w_gemm_supported(g), "sva_test_gemm_f16w: unsupported shape");
    { const int rc = launch_f16w_gemm(g, 0); if (rc) return rc; }
    SVA_HIP(hipDeviceSynchronize());
    SVA_HIP(hipMemcpy(C, dC, sizeof(float) * (size_t)M * NC, hipMemcpyDeviceToHost));
    if (iters > 0 && out_us) {
        hipEvent_t e0, e1;
        SVA_HIP(hipEventCreate(&e0)); SVA_HIP(hipEventCreate(&e1));
        SVA_HIP(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) if (launch_f16w_gemm(g, 0)) return -1;
        SVA_HIP(hipEventRecord(e1, 0));
        SVA_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
        *out_us = ms * 1000.f / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC);
    if (dB) (void)hipFree(dB); if (dN) (void)hipFree(dN); if (dR) (void)hipFree(dR);
    return 0;
}

// Prefill attention: M query rows at positions pos0 .. pos0 + M - 1 of one slot against a cache holding `keys` [pos0 + M][H*64] / `vals`
// (fp32, or rounded to fp16 when half_kv): out_ref = the per-row kernel (ar_attention_kernel), out_mfma = the flash-style MFMA kernel
// (ar_prefill_attention_kernel); us[0], us[1] = their average launch times over `iters`.
extern "C" int sva_test_prefill_attention(int device, int M, int H, int pos0, int S, const float* q, const float* keys, const float* vals,
                                          int half_kv, float* out_ref, float* out_mfma, int iters, float* us) {
    SVA_HIP(hipSetDevice(device));
    const int D = H * 64, L = pos0 + M;
    SVA_CHECK(L <= S, "sva_test_prefill_attention: pos0 + M must fit the cache");
    std::vector<float> qkv((size_t)M * 3 * D, 0.f);
    for (int m = 0; m < M; ++m) memcpy(&qkv[(size_t)m * 3 * D], q + (size_t)m * D, sizeof(float) * D);
    const size_t cache_elems = (size_t)2 * H * S * 64;
    std::vector<float> cf(cache_elems, 0.f);
    for (int j = 0; j < L; ++j)
        for (int h = 0; h < H; ++h)
            for (int dd = 0; dd < 64; ++dd) {
                cf[((size_t)h * S + j) * 64 + dd] = keys[(size_t)j * D + h * 64 + dd];
                cf[(size_t)H * S * 64 + ((size_t)h * S + j) * 64 + dd] = vals[(size_t)j * D + h * 64 + dd];
            }
    std::vector<int> slot(M, 0), pos(M);
    for (int m = 0; m < M; ++m) pos[m] = pos0 + m;
    float *dq, *dc, *do1, *do2;
    int *ds, *dp;
    void* dch = nullptr;
    SVA_HIP(hipMalloc(&dq, sizeof(float) * qkv.size()));
    SVA_HIP(hipMalloc(&dc, sizeof(float) * cache_elems));
    SVA_HIP(hipMalloc(&do1, sizeof(float) * (size_t)M * D));
    SVA_HIP(hipMalloc(&do2, sizeof(float) * (size_t)M * D));
    SVA_HIP(hipMalloc(&ds, sizeof(int) * M));
    SVA_HIP(hipMalloc(&dp, sizeof(int) * M));
    SVA_HIP(hipMemcpy(dq, qkv.data(), sizeof(float) * qkv.size(), hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dc, cf.data(), sizeof(float) * cache_elems, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(ds, slot.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dp, pos.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    if (half_kv) {
        std::vector<uint16_t> ch(cache_elems);
        for (size_t i = 0; i < cache_elems; ++i) { const _Float16 hv = (_Float16)cf[i]; memcpy(&ch[i], &hv, 2); }
        SVA_HIP(hipMalloc(&dch, 2 * cache_elems));
        SVA_HIP(hipMemcpy(dch, ch.data(), 2 * cache_elems, hipMemcpyHostToDevice));
    }
    const long slot_stride = (long)cache_elems;
    auto run = [&](int which) -> int {
        if (half_kv) {
            const __half* c16 = reinterpret_cast<const __half*>(dch);
            return which ? launch_ar_prefill_attention<__half>(dq, M, H, 64, 0, pos0, c16, slot_stride, S, do2, 0)
                         : launch_ar_attention<__half>(dq, M, H, 64, ds, dp, c16, slot_stride, S, do1, 0);
        }
        return which ? launch_ar_prefill_attention<float>(dq, M, H, 64, 0, pos0, dc, slot_stride, S, do2, 0)
                     : launch_ar_attention<float>(dq, M, H, 64, ds, dp, dc, slot_stride, S, do1, 0);
    };
    for (int which = 0; which < 2; ++which) {
        if (run(which)) return -1;
        SVA_HIP(hipDeviceSynchronize());
        if (iters > 0 && us) {
            hipEvent_t e0, e1;
            SVA_HIP(hipEventCreate(&e0)); SVA_HIP(hipEventCreate(&e1));
            SVA_HIP(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) if (run(which)) return -1;
            SVA_HIP(hipEventRecord(e1, 0));
            SVA_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
            us[which] = ms * 1000.f / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
    SVA_HIP(hipMemcpy(out_ref, do1, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToHost));
    SVA_HIP(hipMemcpy(out_mfma, do2, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToHost));
    (void)hipFree(dq); (void)hipFree(dc); (void)hipFree(do1); (void)hipFree(do2); (void)hipFree(ds); (void)hipFree(dp);
    if (dch) (void)hipFree(dch);
    return 0;
}

// The same rows through the decode frame's PAIRED attention kernel (rows 2 i, 2 i + 1 = consecutive positions of one slot; M even): out_mfma = its result
extern "C" int sva_test_pair_attention(int device, int M, int H, int pos0, int S, const float* q, const float* keys, const float* vals,
                                          int half_kv, float* out_ref, float* out_mfma, int iters, float* us) {
    SVA_HIP(hipSetDevice(device));
    const int D = H * 64, L = pos0 + M;
    SVA_CHECK(L <= S, "sva_test_pair_attention: pos0 + M must fit the cache");
    std::vector<float> qkv((size_t)M * 3 * D, 0.f);
    for (int m = 0; m < M; ++m) memcpy(&qkv[(size_t)m * 3 * D], q + (size_t)m * D, sizeof(float) * D);
    const size_t cache_elems = (size_t)2 * H * S * 64;
    std::vector<float> cf(cache_elems, 0.f);
    for (int j = 0; j < L; ++j)
        for (int h = 0; h < H; ++h)
            for (int dd = 0; dd < 64; ++dd) {
                cf[((size_t)h * S + j) * 64 + dd] = keys[(size_t)j * D + h * 64 + dd];
                cf[(size_t)H * S * 64 + ((size_t)h * S + j) * 64 + dd] = vals[(size_t)j * D + h * 64 + dd];
            }
    std::vector<int> slot(M, 0), pos(M);
    for (int m = 0; m < M; ++m) pos[m] = pos0 + m;
    float *dq, *dc, *do1, *do2;
    int *ds, *dp;
    void* dch = nullptr;
    SVA_HIP(hipMalloc(&dq, sizeof(float) * qkv.size()));
    SVA_HIP(hipMalloc(&dc, sizeof(float) * cache_elems));
    SVA_HIP(hipMalloc(&do1, sizeof(float) * (size_t)M * D));
    SVA_HIP(hipMalloc(&do2, sizeof(float) * (size_t)M * D));
    SVA_HIP(hipMalloc(&ds, sizeof(int) * M));
    SVA_HIP(hipMalloc(&dp, sizeof(int) * M));
    SVA_HIP(hipMemcpy(dq, qkv.data(), sizeof(float) * qkv.size(), hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dc, cf.data(), sizeof(float) * cache_elems, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(ds, slot.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dp, pos.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    if (half_kv) {
        std::vector<uint16_t> ch(cache_elems);
        for (size_t i = 0; i < cache_elems; ++i) { const _Float16 hv = (_Float16)cf[i]; memcpy(&ch[i], &hv, 2); }
        SVA_HIP(hipMalloc(&dch, 2 * cache_elems));
        SVA_HIP(hipMemcpy(dch, ch.data(), 2 * cache_elems, hipMemcpyHostToDevice));
    }
    const long slot_stride = (long)cache_elems;
    auto run = [&](int which) -> int {
        if (half_kv) {
            const __half* c16 = reinterpret_cast<const __half*>(dch);
            return which ? launch_ar_attention_pairs<__half>(dq, M, H, 64, ds, dp, c16, slot_stride, S, do2, 0)
                         : launch_ar_attention<__half>(dq, M, H, 64, ds, dp, c16, slot_stride, S, do1, 0);
        }
        return which ? launch_ar_attention_pairs<float>(dq, M, H, 64, ds, dp, dc, slot_stride, S, do2, 0)
                     : launch_ar_attention<float>(dq, M, H, 64, ds, dp, dc, slot_stride, S, do1, 0);
    };
    for (int which = 0; which < 2; ++which) {
        if (run(which)) return -1;
        SVA_HIP(hipDeviceSynchronize());
        if (iters > 0 && us) {
            hipEvent_t e0, e1;
            SVA_HIP(hipEventCreate(&e0)); SVA_HIP(hipEventCreate(&e1));
            SVA_HIP(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) if (run(which)) return -1;
            SVA_HIP(hipEventRecord(e1, 0));
            SVA_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
            us[which] = ms * 1000.f / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
    SVA_HIP(hipMemcpy(out_ref, do1, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToHost));
    SVA_HIP(hipMemcpy(out_mfma, do2, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToHost));
    (void)hipFree(dq); (void)hipFree(dc); (void)hipFree(do1); (void)hipFree(do2); (void)hipFree(ds); (void)hipFree(dp);
    if (dch) (void)hipFree(dch);
    return 0;
}

// microbenchmark of the conv-GEMM dispatcher on device-resident random data:
//   out_us[0] = average microseconds per launch over `iters` back-to-back launches (hipEvents)
// mode bits: 1 = GELU epilogue, 2 = residual + gamma, 4 = silu-on-load, 8 = w13
extern "C" int sva_bench_gemm(int device, int B, int T, int N, int Cin, int taps, int dil, int mode, int iters, float* out_us) {
    SVA_HIP(hipSetDevice(device));
    const int H = (taps - 1) * dil;
    const long rows = H + T;
    float *dA, *dW, *dB, *dC, *dR;
    const long K = (long)taps * Cin;
    const int Nout = (mode & 8) ? N / 2 : N;
    SVA_HIP(hipMalloc(&dA, sizeof(float) * (size_t)B * rows * Cin));
    SVA_HIP(hipMalloc(&dW, sizeof(float) * (size_t)N * K));
    SVA_HIP(hipMalloc(&dB, sizeof(float) * N));
    SVA_HIP(hipMalloc(&dC, sizeof(float) * (size_t)B * T * Nout));
    SVA_HIP(hipMalloc(&dR, sizeof(float) * (size_t)B * T * Nout));
    std::vector<float> h((size_t)std::max<long>((long)B * rows * Cin, (long)N * K));
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    SVA_HIP(hipMemcpy(dA, h.data(), sizeof(float) * (size_t)B * rows * Cin, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dW, h.data(), sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice));
    SVA_HIP(hipMemset(dB, 0, sizeof(float) * N));
    SVA_HIP(hipMemset(dR, 0, sizeof(float) * (size_t)B * T * Nout));
    ConvGemm g;
    g.A = dA; g.a_bstride = rows * Cin; g.a_off = 0; g.lda = Cin; g.T = T; g.M = B * T; g.Cin = Cin; g.taps = taps; g.dil = dil;
    g.W = dW; g.N = N; g.bias = (mode & 8) ? nullptr : dB; g.C = dC; g.c_bstride = (long)T * Nout; g.ldc = Nout;
    if (mode & 1) g.act = ACT_GELU;
    if (mode & 2) { g.res = dR; g.r_bstride = (long)T * Nout; g.ldr = Nout; g.gamma = dB; }
    if (mode & 4) g.a_silu = 1;
    if (mode & 8) g.w13 = 1;
    hipStream_t st;
    SVA_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    SVA_HIP(hipEventCreate(&e0));
    SVA_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { int rc = launch_conv_gemm(g, st); if (rc) return rc; }
    SVA_HIP(hipStreamSynchronize(st));
    SVA_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) { int rc = launch_conv_gemm(g, st); if (rc) return rc; }
    SVA_HIP(hipEventRecord(e1, st));
    SVA_HIP(hipStreamSynchronize(st));
    float ms = 0;
    SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
    out_us[0] = ms * 1e3f / iters;
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dR);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return 0;
}

// One dispatch choice of the conv-GEMM family timed on device-resident random data, `nrot` weight copies rotated launch by launch (large nrot:
// every launch streams its weights from HBM, as inside a step that touches 0.8 GB of weights; 1: the weights stay in L2 / MALL).
//   kind -1 = the dispatcher; 0 = small-M kernel (a = rows / 16, b = K-split waves, c = column tiles + 16 * grid-level K split); 1 tiled (a);
//   2 = pipelined (a); 4 = split-bf16 (a); 6 = weight-streaming kernel (gemm_stream.hip: a = mt + 16 * nt, b = kw, c = wmode + 16 * probe)
//   mode bits: 1 GELU, 2 residual + gamma, 4 SiLU on load, 8 SwiGLU (w13), 16 fused RMSNorm of the rows
//   out[0] = microseconds per launch, eager back-to-back; out[1] = the same launches replayed as one hipGraph; out[2] = max |C - C_dispatcher|;
//   out[3] = max |C_dispatcher|
extern "C" int sva_bench_gemm_choice(int device, int B, int T, int N, int Cin, int taps, int dil, int mode, int kind, int a, int b, int c, int nrot,
                                     int iters, float* out) {
    SVA_HIP(hipSetDevice(device));
    SVA_CHECK(nrot >= 1 && iters >= 1 && out, "bench_gemm_choice: arguments");
    const int H = (taps - 1) * dil;
    const long rows = H + T;
    const long K = (long)taps * Cin;
    const int Nout = (mode & 8) ? N / 2 : N;
    const bool packed = kind == 6 && ((c & 15) & 2);
    float *dA, *dB, *dC, *dC2, *dR, *dG;
    std::vector<float*> dW(nrot, nullptr), dWp(nrot, nullptr);
    SVA_HIP(hipMalloc(&dA, sizeof(float) * (size_t)B * rows * Cin));
    SVA_HIP(hipMalloc(&dB, sizeof(float) * N));
    SVA_HIP(hipMalloc(&dG, sizeof(float) * Cin));
    SVA_HIP(hipMalloc(&dC, sizeof(float) * (size_t)B * T * Nout));
    SVA_HIP(hipMalloc(&dC2, sizeof(float) * (size_t)B * T * Nout));
    SVA_HIP(hipMalloc(&dR, sizeof(float) * (size_t)B * T * Nout));
    std::vector<float> hA((size_t)B * rows * Cin), hW((size_t)N * K), hB(N), hG(Cin), hR((size_t)B * T * Nout);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd() * 0.05f;
    for (auto& v : hB) v = rnd() * 0.1f;
    for (auto& v : hG) v = 1.f + 0.1f * rnd();
    for (auto& v : hR) v = rnd();
    SVA_HIP(hipMemcpy(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dB, hB.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dG, hG.data(), sizeof(float) * Cin, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dR, hR.data(), sizeof(float) * hR.size(), hipMemcpyHostToDevice));
    std::vector<float> hWp;
    if (packed) {           // fragment-major: [N / 16][K / 16][lane = (n & 15) + 16 * (k4)][4]
        const long nkb = K / 16, nt16 = (N + 15) / 16;
        hWp.assign((size_t)nt16 * nkb * 256, 0.f);
        for (long t = 0; t < nt16; ++t)
            for (long kb = 0; kb < nkb; ++kb)
                for (int l = 0; l < 64; ++l) {
                    const long n = t * 16 + (l & 15);
                    if (n >= N) continue;
                    for (int e = 0; e < 4; ++e) hWp[((t * nkb + kb) * 64 + l) * 4 + e] = hW[(size_t)n * K + kb * 16 + 4 * (l >> 4) + e];
                }
    }
    for (int r = 0; r < nrot; ++r) {
        SVA_HIP(hipMalloc(&dW[r], sizeof(float) * hW.size()));
        SVA_HIP(hipMemcpy(dW[r], hW.data(), sizeof(float) * hW.size(), hipMemcpyHostToDevice));
        if (packed) {
            SVA_HIP(hipMalloc(&dWp[r], sizeof(float) * hWp.size()));
            SVA_HIP(hipMemcpy(dWp[r], hWp.data(), sizeof(float) * hWp.size(), hipMemcpyHostToDevice));
        }
    }
    ConvGemm g;
    g.A = dA; g.a_bstride = rows * Cin; g.a_off = 0; g.lda = Cin; g.T = T; g.M = B * T; g.Cin = Cin; g.taps = taps; g.dil = dil;
    g.W = dW[0]; g.N = N; g.bias = (mode & 8) ? nullptr : dB; g.C = dC; g.c_bstride = (long)T * Nout; g.ldc = Nout;
    if (mode & 1) g.act = ACT_GELU;
    if (mode & 2) { g.res = dR; g.r_bstride = (long)T * Nout; g.ldr = Nout; g.gamma = dB; }
    if (mode & 4) g.a_silu = 1;
    if (mode & 8) g.w13 = 1;
    if (mode & 16) g.rms_w = dG;
    hipStream_t st;
    SVA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    SVA_TRY(conv_gemm_prepare_stream(st));
    auto run = [&](int r) -> int {
        ConvGemm q = g;
        q.W = dW[r];
        if (kind < 0) return launch_conv_gemm(q, st);
        if (kind == 6) return launch_stream_gemm(q, packed ? dWp[r] : dW[r], a & 15, a >> 4, b, c & 15, c >> 4, st);
        if (kind == 0 && (c >> 4) > 1) return launch_conv_gemm_choice_z(q, st, a, b, c & 15, c >> 4);
        return launch_conv_gemm_choice(q, st, kind, a, b, c & 15);
    };
    // reference result: the dispatcher's own choice
    {
        ConvGemm q = g;
        q.C = dC2;
        SVA_TRY(launch_conv_gemm(q, st));
    }
    for (int i = 0; i < 3; ++i) SVA_TRY(run(i % nrot));
    SVA_HIP(hipStreamSynchronize(st));
    {
        std::vector<float> c1((size_t)B * T * Nout), c2(c1.size());
        SVA_HIP(hipMemcpy(c1.data(), dC, sizeof(float) * c1.size(), hipMemcpyDeviceToHost));
        SVA_HIP(hipMemcpy(c2.data(), dC2, sizeof(float) * c2.size(), hipMemcpyDeviceToHost));
        float md = 0.f, mx = 0.f;
        for (size_t i = 0; i < c1.size(); ++i) { md = std::max(md, std::fabs(c1[i] - c2[i])); mx = std::max(mx, std::fabs(c2[i])); }
        out[2] = md; out[3] = mx;
    }
    hipEvent_t e0, e1;
    SVA_HIP(hipEventCreate(&e0));
    SVA_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        SVA_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) SVA_TRY(run(i % nrot));
        SVA_HIP(hipEventRecord(e1, st));
        SVA_HIP(hipStreamSynchronize(st));
        float ms = 0;
        SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    out[0] = best * 1e3f / iters;
    // the same chain as one graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    SVA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int i = 0; i < iters && !rc; ++i) rc = run(i % nrot);
    SVA_HIP(hipStreamEndCapture(st, &graph));
    if (rc) return rc;
    SVA_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    SVA_HIP(hipGraphLaunch(exec, st));
    SVA_HIP(hipStreamSynchronize(st));
    best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        SVA_HIP(hipEventRecord(e0, st));
        SVA_HIP(hipGraphLaunch(exec, st));
        SVA_HIP(hipEventRecord(e1, st));
        SVA_HIP(hipStreamSynchronize(st));
        float ms = 0;
        SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    out[1] = best * 1e3f / iters;
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
    for (int r = 0; r < nrot; ++r) { (void)hipFree(dW[r]); if (dWp[r]) (void)hipFree(dWp[r]); }
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dG); (void)hipFree(dC); (void)hipFree(dC2); (void)hipFree(dR);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return 0;
}

// A5 sampler through one explicit implementation (see launch_sampler_variant); noise = the Exp(1) draws [rows, V].
// us_out (optional) = average microseconds per launch over `iters` back-to-back launches.
extern "C" int sva_test_sampler(int device, int variant, int rows, int V, const float* logits, const float* noise, float temperature,
                                float top_p, int* tok_out, int iters, float* us_out) {
    SVA_HIP(hipSetDevice(device));
    float *dL, *dN;
    int* dT;
    const size_t n = (size_t)rows * V;
    SVA_HIP(hipMalloc(&dL, sizeof(float) * n));
    SVA_HIP(hipMalloc(&dN, sizeof(float) * n));
    SVA_HIP(hipMalloc(&dT, sizeof(int) * rows));
    SVA_HIP(hipMemcpy(dL, logits, sizeof(float) * n, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dN, noise, sizeof(float) * n, hipMemcpyHostToDevice));
    int rc = launch_sampler_variant(variant, dL, rows, V, V, dN, V, nullptr, nullptr, temperature, top_p, dT, 0);
    if (!rc) {
        SVA_HIP(hipDeviceSynchronize());
        SVA_HIP(hipMemcpy(tok_out, dT, sizeof(int) * rows, hipMemcpyDeviceToHost));
        if (us_out && iters > 0) {
            hipEvent_t e0, e1;
            SVA_HIP(hipEventCreate(&e0));
            SVA_HIP(hipEventCreate(&e1));
            SVA_HIP(hipEventRecord(e0, 0));
            for (int i = 0; i < iters && !rc; ++i) rc = launch_sampler_variant(variant, dL, rows, V, V, dN, V, nullptr, nullptr, temperature, top_p, dT, 0);
            SVA_HIP(hipEventRecord(e1, 0));
            SVA_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
            us_out[0] = ms * 1000.f / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
    (void)hipFree(dL); (void)hipFree(dN); (void)hipFree(dT);
    return rc;
}

// Host cost of enqueueing one kernel from the calling thread (microseconds), measured with `iters` launches of a one-element
// kernel into an otherwise idle stream.  A single-stream step is ~430 launches, so the launch rate of the enqueueing thread
// bounds the step rate, and on a two-socket host it depends on which core the thread runs on (measured 3.7 vs 4.8 us);
// engine.py pin_enqueue_thread() uses this to place the thread.
extern "C" int sva_host_launch_cost(int device, int iters, float* us_per_launch) {
    SVA_CHECK(iters > 0 && us_per_launch, "host_launch_cost: iters > 0 and an output pointer");
    SVA_HIP(hipSetDevice(device));
    static std::mutex mu;
    static std::map<int, std::pair<hipStream_t, int*>> res;
    std::lock_guard<std::mutex> lk(mu);
    auto& r = res[device];
    if (!r.first) {
        SVA_HIP(hipStreamCreateWithFlags(&r.first, hipStreamNonBlocking));
        SVA_HIP(hipMalloc((void**)&r.second, 64 * sizeof(int)));
    }
    for (int i = 0; i < 32; ++i) if (int rc = launch_fill_i32(r.second, 1, i, r.first)) return rc;
    SVA_HIP(hipStreamSynchronize(r.first));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) if (int rc = launch_fill_i32(r.second, 1, i, r.first)) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    SVA_HIP(hipStreamSynchronize(r.first));
    *us_per_launch = (float)(std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    return 0;
}

// The planes GEMM (gemm_planes.hip) on one problem: weights split on the device as the engine does at finalize.  mode = PlanesMode,
// variant = tile variant; flags: 1 = the A operand as planes (to_planes pass first), 2 = the result leaves as planes only (summed back
// to fp32 here), 4 = GELU epilogue, 8 = SiLU on load (fp32 A only).  iters > 0 also times it (out_us[0] = microseconds per launch).
extern "C" int sva_test_gemm_planes(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C, int mode,
                                    int variant, int flags, int iters, float* out_us) {
    SVA_HIP(hipSetDevice(device));
    SVA_CHECK((mode == PLANES_H3 || mode == PLANES_H1) && K % 32 == 0 && N % 4 == 0, "test_gemm_planes: bad arguments");
    const int npl = planes_count(mode);
    float *dA, *dW, *dB = nullptr, *dC;
    unsigned short *dWp, *dAp = nullptr, *dCp = nullptr;
    SVA_HIP(hipMalloc(&dA, sizeof(float) * (size_t)M * K));
    SVA_HIP(hipMalloc(&dW, sizeof(float) * (size_t)N * K));
    SVA_HIP(hipMalloc(&dC, sizeof(float) * (size_t)M * N));
    SVA_HIP(hipMalloc(&dWp, 2 * (size_t)npl * N * K));
    SVA_HIP(hipMemcpy(dA, A, sizeof(float) * (size_t)M * K, hipMemcpyHostToDevice));
    SVA_HIP(hipMemcpy(dW, W, sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice));
    if (bias) {
        SVA_HIP(hipMalloc(&dB, sizeof(float) * N));
        SVA_HIP(hipMemcpy(dB, bias, sizeof(float) * N, hipMemcpyHostToDevice));
    }
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)N * K; ++i) mx = std::max(mx, fabsf(W[i]));
    ConvGemm g;
    g.A = dA; g.a_bstride = (long)M * K; g.lda = K; g.T = M; g.M = M; g.Cin = K; g.taps = 1;
    g.W = dW; g.N = N; g.bias = dB; g.C = dC; g.c_bstride = (long)M * N; g.ldc = N;
    g.Wp = dWp; g.wp_pstride = (long)N * K; g.pmode = mode;
    SVA_TRY(make_weight_planes(dW, N, K, mx, mode, dWp, &g.wp_inv, 0));
    if (flags & 1) {
        SVA_HIP(hipMalloc(&dAp, 2 * (size_t)npl * M * K));
        SVA_TRY(launch_to_planes(dA, M, K, K, dAp, (long)M * K, mode, 1.f, 0, 0));
        g.Ap = dAp; g.ap_pstride = (long)M * K; g.ap_rows = M; g.A = nullptr;
    }
    if (flags & 2) {
        SVA_HIP(hipMalloc(&dCp, 2 * (size_t)npl * M * N));
        g.Cp = dCp; g.cp_pstride = (long)M * N; g.cp_rows = M; g.C = nullptr;
    }
    if (flags & 4) g.act = ACT_GELU;
    if ((flags & 8) && !(flags & 1)) g.a_silu = 1;
    // flags 32: SwiGLU (W rows interleave w1 | w3 in groups of 16; C is [M][N / 2]); 64: gamma (= bias vector reversed) and a residual (= a
    // deterministic pattern) in front of the store; 128: rows t in [T / 3, T / 3 + 6) of every batch item of T = 170 rows are not stored
    // (M % 170 == 0), C is pre-filled with a marker there
    const int Nout = (flags & 32) ? N / 2 : N;
    float *dG = nullptr, *dR = nullptr;
    if (flags & 32) {
        g.w13 = 1; g.ldc = Nout; g.c_bstride = (long)M * Nout; g.cp_pstride = (long)M * Nout;
    }
    if (flags & 64) {
        std::vector<float> hg(N), hr((size_t)M * N);
        for (int i = 0; i < N; ++i) hg[i] = 0.5f + 0.001f * (float)((i * 37) % 101);
        for (size_t i = 0; i < hr.size(); ++i) hr[i] = 0.01f * (float)((i * 13) % 257) - 1.f;
        SVA_HIP(hipMalloc(&dG, sizeof(float) * N));
        SVA_HIP(hipMalloc(&dR, sizeof(float) * (size_t)M * N));
        SVA_HIP(hipMemcpy(dG, hg.data(), sizeof(float) * N, hipMemcpyHostToDevice));
        SVA_HIP(hipMemcpy(dR, hr.data(), sizeof(float) * hr.size(), hipMemcpyHostToDevice));
        g.gamma = dG; g.res = dR; g.r_bstride = (long)M * N; g.ldr = N;
    }
    if (flags & 128) {
        SVA_CHECK(M % 170 == 0 && !(flags & 2), "test_gemm_planes: the skip-rows case takes M = 170 b and an fp32 C");
        g.T = 170; g.a_bstride = (long)170 * K; g.c_bstride = (long)170 * Nout; g.r_bstride = (long)170 * N; g.skip_lo = 56; g.skip_hi = 62;
        std::vector<float> mark((size_t)M * Nout, -77.f);
        SVA_HIP(hipMemcpy(dC, mark.data(), sizeof(float) * mark.size(), hipMemcpyHostToDevice));
    }
    int* h_ovf = nullptr;
    if (flags & 16) {       // the range check of the fp16 formats: a non-finite output is an error of the call
        SVA_HIP(hipHostMalloc((void**)&h_ovf, sizeof(int), hipHostMallocMapped));
        *h_ovf = 0;
        SVA_HIP(hipHostGetDevicePointer((void**)&g.ovf, h_ovf, 0));
    }
    SVA_TRY(launch_conv_gemm_choice(g, 0, 6, variant, 0, 0));
    SVA_HIP(hipDeviceSynchronize());
    if (h_ovf) {
        const int o = *reinterpret_cast<volatile int*>(h_ovf);
        (void)hipHostFree(h_ovf);
        g.ovf = nullptr;
        if (o) { set_error("planes GEMM: non-finite output (an operand outside the fp16 range)"); return -3; }
    }
    if (flags & 2) {
        std::vector<uint16_t> hp((size_t)npl * M * N);
        SVA_HIP(hipMemcpy(hp.data(), dCp, hp.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)M * Nout; ++i) {
            float s = 0.f;
            const size_t row = i / Nout, col = i % Nout;
            const size_t bo = ((col >> 5) * (size_t)M + row) * 32 + (col & 31);          // K-blocked (planes_split.h)
            for (int p = npl - 1; p >= 0; --p) {
                const uint16_t bits = hp[(size_t)p * M * Nout + bo];
                _Float16 h; memcpy(&h, &bits, 2); s += (float)h;
            }
            C[i] = s;
        }
    } else {
        SVA_HIP(hipMemcpy(C, dC, sizeof(float) * (size_t)M * Nout, hipMemcpyDeviceToHost));
    }
    if (iters > 0 && out_us) {
        hipEvent_t e0, e1;
        SVA_HIP(hipEventCreate(&e0));
        SVA_HIP(hipEventCreate(&e1));
        SVA_HIP(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) SVA_TRY(launch_conv_gemm_choice(g, 0, 6, variant, 0, 0));
        SVA_HIP(hipEventRecord(e1, 0));
        SVA_HIP(hipDeviceSynchronize());
        float ms = 0;
        SVA_HIP(hipEventElapsedTime(&ms, e0, e1));
        out_us[0] = ms * 1e3f / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dWp);
    if (dB) (void)hipFree(dB);
    if (dAp) (void)hipFree(dAp);
    if (dCp) (void)hipFree(dCp);
    if (dG) (void)hipFree(dG);
    if (dR) (void)hipFree(dR);
    return 0;
}
