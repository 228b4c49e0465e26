// f32 MFMA conv-GEMM for gfx950 (v_mfma_f32_16x16x4_f32: exact f32, 32-cycle issue).
//
// Why f32 MFMA: BSQ content codes must be bit-exact against the fp32 reference
// (SURVEY.md §7 hard part 2), so the encoder cannot use fp16/bf16 matrix cores; the f32-input
// MFMA is bitwise an fmaf chain at the f32 vector peak (157 TF) with one VGPR per operand.
//
// Tiling: 256 threads = 4 waves in a WM x WN grid, block tile BM x BN, BK = 16.  A and W tiles
// are staged global -> registers -> LDS (k-major, row stride +16 floats so the two k-rows a
// 32-lane ds_read_b32 group touches fall on disjoint bank halves) and double-buffered: the
// next tile's global loads are issued before the current tile's MFMAs and written to the other
// LDS buffer after them (one barrier per K tile).
#include "sva_common.h"

namespace sva {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemm g) {
    constexpr int BK = 16;
    constexpr int TM = BM / WM, TN = BN / WN;     // wave tile
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int LDA_S = BM + 16, LDB_S = BN + 16;
    constexpr int A_LD = (BM * 4 + 255) / 256;    // float4 loads per thread per tile
    constexpr int B_LD = (BN * 4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ float As[2][BK][LDA_S];
    __shared__ float Bs[2][BK][LDB_S];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;
    const int kq = tid & 3;                       // which float4 of the 16-wide k tile
    const int lrow = tid >> 2;                    // 0..63

    const float* a_ptr[A_LD];
    bool a_on[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int r = lrow + i * 64;
        a_on[i] = r < BM;
        int m = bm0 + r;
        if (m > g.M - 1) m = g.M - 1;
        int b = m / g.T, t = m - b * g.T;
        a_ptr[i] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + kq * 4;
    }
    const float* b_ptr[B_LD];
    bool b_on[B_LD];
    const long Kt = (long)g.taps * g.Cin;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        int r = lrow + i * 64;
        b_on[i] = r < BN;
        int n = bn0 + r;
        if (n > g.N - 1) n = g.N - 1;
        b_ptr[i] = g.W + (long)n * Kt + kq * 4;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kc_tiles = g.Cin / BK;
    const int nk = g.taps * kc_tiles;
    float4 ra[A_LD], rb[B_LD];

    auto gload = [&](int kt) {
        int tap = kt / kc_tiles;
        int kc = (kt - tap * kc_tiles) * BK;
        long aoff = (long)tap * g.dil * g.lda + kc;
        long boff = (long)tap * g.Cin + kc;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            if (a_on[i]) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + aoff);
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            if (b_on[i]) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + boff);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            if (a_on[i]) {
                float4 v = ra[i];
                if (g.a_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                int r = lrow + i * 64;
                As[buf][kq * 4 + 0][r] = v.x;
                As[buf][kq * 4 + 1][r] = v.y;
                As[buf][kq * 4 + 2][r] = v.z;
                As[buf][kq * 4 + 3][r] = v.w;
            }
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            if (b_on[i]) {
                float4 v = rb[i];
                int r = lrow + i * 64;
                Bs[buf][kq * 4 + 0][r] = v.x;
                Bs[buf][kq * 4 + 1][r] = v.y;
                Bs[buf][kq * 4 + 2][r] = v.z;
                Bs[buf][kq * 4 + 3][r] = v.w;
            }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            float af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = As[buf][ks + fk][wm * TM + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = Bs[buf][ks + fk][wn * TN + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg ----
    const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = bm0 + wm * TM + i * 16 + rq + r;
            if (m >= g.M) continue;
            const int b = m / g.T, t = m - b * g.T;
            float* crow = g.C + (long)b * g.c_bstride + g.c_off + (long)t * g.ldc;
            const float* rrow = g.res ? g.res + (long)b * g.r_bstride + g.r_off + (long)t * g.ldr : nullptr;
            if (g.w13) {
                if constexpr (NI >= 2) {
#pragma unroll
                    for (int j = 0; j < NI; j += 2) {
                        const int n = bn0 + wn * TN + j * 16 + col;       // w1 column (even 16-group)
                        if (n + 16 < g.N + 16 && n < g.N) {
                            float a = acc[i][j][r], bb = acc[i][j + 1][r];
                            const int no = ((bn0 + wn * TN + j * 16) >> 1) + col;
                            crow[no] = silu_f(a) * bb;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = bn0 + wn * TN + j * 16 + col;
                    if (n >= g.N) continue;
                    float v = acc[i][j][r];
                    if (g.bias) v += g.bias[n];
                    if (g.act == ACT_GELU) v = gelu_erf(v);
                    else if (g.act == ACT_LOGCLAMP) v = __logf(fmaxf(v, 1e-5f));
                    if (g.gamma) v *= g.gamma[n];
                    if (rrow) v += rrow[n];
                    v *= g.scale;
                    if (g.accumulate) v += crow[n];
                    crow[n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static void launch_t(const ConvGemm& g, hipStream_t st) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, st, g);
}

int launch_conv_gemm(const ConvGemm& g, hipStream_t st) {
    SVA_CHECK(g.Cin % 16 == 0 && g.Cin > 0, "conv_gemm: Cin must be a multiple of 16");
    SVA_CHECK(g.lda % 4 == 0 && (g.a_off % 4) == 0 && (g.a_bstride % 4) == 0, "conv_gemm: A must be float4-aligned");
    SVA_CHECK(g.M > 0 && g.N > 0 && g.T > 0, "conv_gemm: empty problem");
    if (g.w13) SVA_CHECK(g.N % 32 == 0, "conv_gemm: w13 needs N % 32 == 0");
    // tile selection: fill >= ~256 workgroups where the problem allows it
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (g.N <= 16 && !g.w13) {
        launch_t<256, 16, 4, 1>(g, st);
    } else if (g.N <= 32) {
        launch_t<128, 32, 4, 1>(g, st);
    } else if (g.M <= 16) {
        launch_t<16, 128, 1, 4>(g, st);
    } else if (g.M <= 32) {
        launch_t<32, 128, 1, 4>(g, st);
    } else if (big >= 256) {
        launch_t<128, 128, 2, 2>(g, st);
    } else {
        launch_t<64, 64, 2, 2>(g, st);
    }
    SVA_HIP(hipGetLastError());
    return 0;
}

}  // namespace sva
