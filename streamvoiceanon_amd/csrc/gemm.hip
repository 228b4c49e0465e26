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
#include <array>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>
#include <stdlib.h>
#include <stdio.h>

namespace sva {

constexpr int KS_ERR_WORD = (1 << 16) - 1;      // last word of a stream's split-K arrival counters: set when a partial never arrived

#define SVA_TRY_RC(expr)     \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_kernel(const ConvGemmGroup gg) {
    constexpr int NTH = 64 * WM * WN;             // 4 or 8 waves
    const ConvGemm& g = gg.g[blockIdx.z];
    constexpr int TM = BM / WM, TN = BN / WN;     // wave tile
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int LS = BK + 4;                    // LDS row stride: 16-byte aligned rows; row*LS mod 64 banks is a permutation of
                                                  // the multiples of 4 over 16 rows, so the ds_read_b128 fragments are conflict-free
    constexpr int F4R = BK / 4;                   // float4 per tile row
    constexpr int RPP = NTH / F4R;                // rows covered per pass of the workgroup's threads
    constexpr int A_LD = (BM + RPP - 1) / RPP;
    constexpr int B_LD = (BN + RPP - 1) / RPP;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                             // [2][BM][LS]
    float* Bs = smem + 2 * BM * LS;               // [2][BN][LS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tbx = blockIdx.x, tby = blockIdx.y;
    xcd_tile(gg.xcd_swz, gridDim.x, gridDim.y, tbx, tby);
    const int bm0 = tby * BM, bn0 = tbx * BN;
    const int kq = tid % F4R;                     // which float4 of the BK-wide k tile
    const int lrow = tid / F4R;

    const float* a_ptr[A_LD];
    bool a_on[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int r = lrow + i * RPP;
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
        int r = lrow + i * RPP;
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
                *reinterpret_cast<float4*>(As + ((buf * BM + lrow + i * RPP) * LS + kq * 4)) = v;
            }
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            if (b_on[i]) {
                float4 v = rb[i];
                *reinterpret_cast<float4*>(Bs + ((buf * BN + lrow + i * RPP) * LS + kq * 4)) = v;
            }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        // One ds_read_b128 per fragment and 16 k: lane (fr, fk) supplies k = 16*blk + 4*fk + j to MFMA step j of the block --
        // a permutation of k inside the block, identical on both operands (same trick as the small-M kernel).
        const float* Ab = As + (buf * BM + wm * TM + fr) * LS + fk * 4;
        const float* Bb = Bs + (buf * BN + wn * TN + fr) * LS + fk * 4;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            float4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 16 * LS + ks);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 16 * LS + ks);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    // The accumulators (C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg) are staged
    // through LDS so that bias / residual reads and the C stores are whole 16-byte, row-contiguous accesses
    // (a lane-per-element epilogue touches a 64-byte segment per row per instruction and costs up to 30 % of a
    // K = 512 GEMM).  The last loop iteration ended with a barrier, so the A/B buffers are free to reuse.
    constexpr int CS = BN + 4;
    float* Cs = smem;                              // [BM][CS]
    const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wm * TM + i * 16 + rq + r) * CS + wn * TN + j * 16 + col] = acc[i][j][r];
    __syncthreads();
    if (g.w13) {
        // SwiGLU: tile columns alternate 16 x w1 | 16 x w3; output column (n0 >> 1) + c
        constexpr int OC4 = BN / 8;                // float4 chunks of output per row
        for (int idx = tid; idx < BM * OC4; idx += NTH) {
            const int row = idx / OC4, q = idx - row * OC4;
            const int m = bm0 + row;
            const int grp = q >> 2, c4 = (q & 3) * 4;       // 16-wide group, offset inside it
            const int n = bn0 + grp * 32 + c4;              // w1 column
            if (m >= g.M || n >= g.N) continue;
            const int b = m / g.T, t = m - b * g.T;
            if (t >= g.skip_lo && t < g.skip_hi) continue;
            const float4 a = *reinterpret_cast<const float4*>(&Cs[row * CS + grp * 32 + c4]);
            const float4 w = *reinterpret_cast<const float4*>(&Cs[row * CS + grp * 32 + 16 + c4]);
            float4 o;
            o.x = silu_f(a.x) * w.x; o.y = silu_f(a.y) * w.y; o.z = silu_f(a.z) * w.z; o.w = silu_f(a.w) * w.w;
            float* crow = g.C + (long)b * g.c_bstride + g.c_off + (long)t * g.ldc;
            *reinterpret_cast<float4*>(crow + ((bn0 + grp * 32) >> 1) + c4) = o;
        }
        return;
    }
    constexpr int C4 = BN / 4;
    for (int idx = tid; idx < BM * C4; idx += NTH) {
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        const int m = bm0 + row, n = bn0 + c4;
        if (m >= g.M || n >= g.N) continue;
        const int b = m / g.T, t = m - b * g.T;
        if (t >= g.skip_lo && t < g.skip_hi) continue;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * CS + c4]);
        if (g.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (g.act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        else if (g.act == ACT_LOGCLAMP) { v.x = __logf(fmaxf(v.x, 1e-5f)); v.y = __logf(fmaxf(v.y, 1e-5f)); v.z = __logf(fmaxf(v.z, 1e-5f)); v.w = __logf(fmaxf(v.w, 1e-5f)); }
        if (g.gamma) {
            const float4 gg = *reinterpret_cast<const float4*>(g.gamma + n);
            v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
        }
        if (g.res) {
            const float4 rr = *reinterpret_cast<const float4*>(g.res + (long)b * g.r_bstride + g.r_off + (long)t * g.ldr + n);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        v.x *= g.scale; v.y *= g.scale; v.z *= g.scale; v.w *= g.scale;
        float* cp = g.C + (long)b * g.c_bstride + g.c_off + (long)t * g.ldc + n;
        if (g.accumulate) {
            const float4 cc = *reinterpret_cast<const float4*>(cp);
            v.x += cc.x; v.y += cc.y; v.z += cc.z; v.w += cc.w;
        }
        *reinterpret_cast<float4*>(cp) = v;
    }
}

// ------------------------------------------------------------------------------------------
// small-M ("skinny") path: M <= 64 rows -- the AR decode GEMMs (M = 2B / B tokens) and the streaming
// vocoder's first levels.  These are weight-streaming problems: the tiled kernel above would run a
// handful of workgroups through a long serial K loop (one 16-wide tile in flight per barrier) and sit on
// HBM/L2 latency.  Here a workgroup owns 16*NT output columns and ALL rows; its KW waves split the K axis
// (interleaved 16-wide blocks), every lane streams one float4 of W and one float4 of A per block straight
// from global memory into the MFMA operands (no LDS, no barrier in the loop, loads software-pipelined by
// unrolling), and the KW partial accumulators are reduced through LDS once at the end.
// Operand mapping: lane l supplies, for MFMA step j of a block, W[n0 + (l&15)][k0 + 4*(l>>4) + j] and
// A[m0 + (l&15)][k0 + 4*(l>>4) + j] -- a permutation of k inside the block, identical on both operands.
// ------------------------------------------------------------------------------------------
// AOP: what happens to A on load -- 0 nothing, 1 SiLU (HiFiGAN), 2 RMSNorm of the row (weight folded into the operand,
// row statistics accumulated on the fly and applied to the accumulators in the epilogue).
template <int MT, int NT, int KW, int D, int AOP>
__global__ __launch_bounds__(64 * KW) void skinny_gemm_kernel(const ConvGemmGroup gg) {
    // blockIdx.z: member of a group, or (single problem) the K split this workgroup owns
    const ConvGemm& g = gg.g[gg.n > 1 ? blockIdx.z : 0];
    const int Z = gg.n > 1 ? 1 : g.ksplit, ks = gg.n > 1 ? 0 : (int)blockIdx.z;
    constexpr bool SILU = AOP == 1, RMS = AOP == 2, DWLN = AOP == 3;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [KW][MT*NT][64][4] (+ [KW][MT][16] row sums of squares)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * (16 * NT);
    const int m_base = blockIdx.y * (16 * MT);
    const int fr = lane & 15, fg = lane >> 4;
    const long Kt = (long)g.taps * g.Cin;
    const float* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + j * 16 + fr;
        if (n > g.N - 1) n = g.N - 1;
        wp[j] = g.W + (long)n * Kt + 4 * fg;
    }
    const float* ap[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m_base + i * 16 + fr;
        if (m > g.M - 1) m = g.M - 1;
        const int b = m / g.T, t = m - b * g.T;
        ap[i] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + 4 * fg;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // AOP 3: ConvNeXt prologue -- rows of this tile = LayerNorm(depthwise k7 conv) of x, one wave per row, kept in LDS with a
    // padded row stride (the K loop then reads its A fragments from there instead of from global memory)
    const int a_ld = g.Cin + 4;
    float* a_s = red + KW * MT * NT * 256;                 // [16*MT][Cin + 4]
    if constexpr (DWLN) {
        const int C = g.Cin;
        for (int r = wave; r < 16 * MT; r += KW) {
            const int m = m_base + r;
            float* dst = a_s + r * a_ld;
            if (m >= g.M) {
                for (int c = lane; c < C; c += 64) dst[c] = 0.f;
                continue;
            }
            const int b = m / g.T, t = m - b * g.T;
            const float* xr = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.lda;      // tap 0 row
            float v[8], sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                float acc = 0.f;
                if (c < C) {
                    acc = g.dw_b[c];
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc = fmaf(g.dw_wT[j * C + c], xr[(long)j * g.lda + c], acc);
                    sum += acc;
                }
                v[i] = acc;
            }
            sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (lane + 64 * i < C) { const float d = v[i] - mean; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            q += __shfl_xor(q, 8, 64); q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const float inv = 1.f / sqrtf(q / (float)C + g.ln_eps);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                if (c < C) dst[c] = (v[i] - mean) * inv * g.ln_w[c] + g.ln_b[c];
            }
        }
        __syncthreads();
    }
    const int kc_tiles = g.Cin / 16;
    const int nk = g.taps * kc_tiles;
    // software pipeline, D K-blocks in flight per wave.  No branch around any load (a conditional load makes
    // hipcc drain vmcnt(0) at the join): out-of-range blocks re-load the wave's last valid block and are masked.
    // this workgroup's share of the K blocks (grid-level split), interleaved over its waves
    const int kb_lo = (int)((long)ks * nk / Z), kb_hi = (int)((long)(ks + 1) * nk / Z);
    const int nk_loc = kb_hi - kb_lo;
    const int my_n = nk_loc > wave ? (nk_loc - wave + KW - 1) / KW : 0;          // K blocks owned by this wave
    const int last_kb = my_n > 0 ? kb_lo + wave + (my_n - 1) * KW : kb_lo;
    auto kbq = [&](int q) { return kb_lo + wave + q * KW; };
    float4 wv[D][NT], av[D][MT], nv[D];
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ssq[i] = 0.f;
    auto issue = [&](float4 (&w)[NT], float4 (&a)[MT], float4& nw, int kb) {
        kb = kb < kb_hi ? kb : last_kb;
        const int tap = kb / kc_tiles;
        const int kc = (kb - tap * kc_tiles) * 16;
        const long aoff = (long)tap * g.dil * g.lda + kc;
        const long woff = (long)tap * g.Cin + kc;
#pragma unroll
        for (int j = 0; j < NT; ++j) w[j] = *reinterpret_cast<const float4*>(wp[j] + woff);
        if constexpr (DWLN) {            // A fragments come from the LDS tile written by the prologue (no latency to hide)
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(a_s + (i * 16 + fr) * a_ld + kc + 4 * fg);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(ap[i] + aoff);
        }
        if (RMS) nw = *reinterpret_cast<const float4*>(g.rms_w + kc + 4 * fg);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(wv[d], av[d], nv[d], kbq(d));
    for (int it = 0; it < my_n; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kb = kbq(it + d);
            const float keep = kb < kb_hi ? 1.f : 0.f;
            float4 w[NT], a[MT];
#pragma unroll
            for (int j = 0; j < NT; ++j) w[j] = wv[d][j];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                a[i] = av[d][i];
                if (SILU) { a[i].x = silu_f(a[i].x); a[i].y = silu_f(a[i].y); a[i].z = silu_f(a[i].z); a[i].w = silu_f(a[i].w); }
                a[i].x *= keep; a[i].y *= keep; a[i].z *= keep; a[i].w *= keep;
                if (RMS) {
                    // explicit sequential fmas: under -ffp-contract the pairwise form was fused differently per unrolled row tile (gemm_stream.hip), so a row's statistic depended on its position
                    ssq[i] = __builtin_fmaf(a[i].x, a[i].x, ssq[i]);
                    ssq[i] = __builtin_fmaf(a[i].y, a[i].y, ssq[i]);
                    ssq[i] = __builtin_fmaf(a[i].z, a[i].z, ssq[i]);
                    ssq[i] = __builtin_fmaf(a[i].w, a[i].w, ssq[i]);
                    a[i].x *= nv[d].x; a[i].y *= nv[d].y; a[i].z *= nv[d].z; a[i].w *= nv[d].w;
                }
            }
            issue(wv[d], av[d], nv[d], kbq(it + d + D));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, w[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, w[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, w[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, w[j].w, acc[i][j], 0, 0, 0);
        }
    }
    // cross-wave reduction of the K slices
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + j) * 64 + lane) * 4]) = acc[i][j];
    float* redss = red + KW * MT * NT * 256;                          // [KW][MT][16]
    if (RMS) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fg == 0) redss[(wave * MT + i) * 16 + fr] = v;
        }
    }
    __syncthreads();
    // tail: the MT row tiles are spread over the waves (each sums the KW partials of its tiles straight from LDS and runs
    // the epilogue for them) instead of leaving all of it to wave 0
    const int col = lane & 15, rq = (lane >> 4) * 4;
    auto epilogue = [&](int i, f32x4 (&t)[NT]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + i * 16 + rq + r;
            if (m >= g.M) continue;
            const int b = m / g.T, tt = m - b * g.T;
            if (tt >= g.skip_lo && tt < g.skip_hi) continue;
            float* crow = g.C + (long)b * g.c_bstride + g.c_off + (long)tt * g.ldc;
            const float* rrow = g.res ? g.res + (long)b * g.r_bstride + g.r_off + (long)tt * g.ldr : nullptr;
            if (g.w13) {
                if constexpr (NT % 2 == 0) {          // column tiles come in (gate, up) pairs
#pragma unroll
                    for (int jp = 0; jp < NT / 2; ++jp) {
                        const int n = n0 + jp * 32 + col;
                        if (n < g.N) crow[(n0 >> 1) + jp * 16 + col] = silu_f(t[2 * jp][r]) * t[2 * jp + 1][r];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = n0 + j * 16 + col;
                    if (n >= g.N) continue;
                    float v = t[j][r];
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
    };
    const int tile = blockIdx.y * gridDim.x + blockIdx.x, n_tiles = gridDim.x * gridDim.y;
    for (int i = wave; i < MT; i += KW) {
        f32x4 t[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 s = *reinterpret_cast<const f32x4*>(&red[((i * NT + j) * 64 + lane) * 4]);
#pragma unroll
            for (int w = 1; w < KW; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * (MT * NT) + i * NT + j) * 64 + lane) * 4]);
            t[j] = s;
        }
        if (RMS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < KW; ++w) tot += redss[(w * MT + i) * 16 + rq + r];
                const float inv = 1.f / sqrtf(tot / (float)Kt + g.rms_eps);
#pragma unroll
                for (int j = 0; j < NT; ++j) t[j][r] *= inv;
            }
        }
        if (Z == 1) {
            epilogue(i, t);
        } else {          // raw partial tile of this K split as self-validating granules {tag = 1, value} (register layout)
            unsigned long long* gw = reinterpret_cast<unsigned long long*>(g.ks_ws);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __hip_atomic_store(gw + ((((long)ks * n_tiles + tile) * MT + i) * NT + j) * 256 + e * 64 + lane,
                                       (1ull << 32) | (unsigned long long)__float_as_uint(t[j][e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (Z == 1) return;
    // The last workgroup to arrive for this output tile sums the Z partials in split order (deterministic) and runs the epilogue.
    // Hand-off without fences (guide G16, form R2 -- as in ar_decode.hip): every datum is an 8-byte granule {tag, value} written
    // with one relaxed agent-scope atomic store, the reader polls each granule until its tag is set.  The arrival counter only
    // ELECTS the reader, it carries no visibility promise: a writer's granules may land after its counter increment, the poll
    // covers that.  The reader clears the tags it consumed (the next launch on this stream starts behind a kernel boundary).
    // An agent-scope release per writer -- what a counter-validated hand-off needs across XCDs -- cost more than the split won.
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(g.ks_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(Z - 1)) __hip_atomic_store(g.ks_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        reinterpret_cast<unsigned*>(red)[0] = old;
    }
    __syncthreads();
    if (reinterpret_cast<const unsigned*>(red)[0] != (unsigned)(Z - 1)) return;
    unsigned long long* gw = reinterpret_cast<unsigned long long*>(g.ks_ws);
    for (int i = wave; i < MT; i += KW) {
        f32x4 t[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) t[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < Z; ++z) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                unsigned long long* gp = gw + ((((long)z * n_tiles + tile) * MT + i) * NT + j) * 256 + lane;
                unsigned long long x[4];
                unsigned pending = 0xfu;
                for (int spin = 0; pending && spin < (1 << 20); ++spin) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (pending & (1u << e)) {
                            x[e] = __hip_atomic_load(gp + e * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(x[e] >> 32) == 1u) pending &= ~(1u << e);
                        }
                }
                // (every writer finished its stores before it took its arrival ticket, so the poll only bridges their flight time;
                // a partial that never shows up is a hardware / protocol fault: flag it -- sva_sync reports it -- and leave its tag alone)
                if (pending) __hip_atomic_store(g.ks_cnt + KS_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (pending & (1u << e)) continue;
                    t[j][e] += __uint_as_float((unsigned)x[e]);
                    __hip_atomic_store(gp + e * 64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        epilogue(i, t);
    }
}

// set by launch_conv_gemm_group around the dispatch: the launchers then send the whole group (grid.z = members)
static thread_local const ConvGemmGroup* t_group = nullptr;
// grid-level K split requested by the dispatch choice for the next small-M launch
static thread_local int t_ksplit = 1;
// split-K scratch (partial tiles + arrival counters), one per stream: launches on one stream are ordered, so they can
// share it; concurrent streams must not
struct KsScratch { float* ws = nullptr; unsigned* cnt = nullptr; };
static std::mutex g_ks_mu;
static std::unordered_map<hipStream_t, KsScratch> g_ks;
constexpr size_t KS_WS_FLOATS = (size_t)4 << 20;      // granules of partial tiles (8 bytes each: 32 MiB)
constexpr int KS_CNT = 1 << 16;
static_assert(KS_ERR_WORD == KS_CNT - 1, "error word = last counter");
// sva_sync: has any split-K reader on any stream of this process flagged a missing partial?
int conv_gemm_check_errors() {
    std::lock_guard<std::mutex> lk(g_ks_mu);
    for (auto& kv : g_ks) {
        if (!kv.second.cnt) continue;
        unsigned v = 0;
        SVA_HIP(hipMemcpy(&v, kv.second.cnt + KS_ERR_WORD, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (v != 0) {
            // reported once (the word used to stay set, so every later sva_sync of every batch failed): the fault word and the
            // whole hand-off scratch of that stream are cleared -- a partial that arrives late must not be consumed as a stale tile
            SVA_HIP(hipDeviceSynchronize());
            SVA_HIP(hipMemset(kv.second.ws, 0, KS_WS_FLOATS * 8));
            SVA_HIP(hipMemset(kv.second.cnt, 0, KS_CNT * sizeof(unsigned)));
            SVA_CHECK(false, "split-K GEMM: a partial tile never arrived (hand-off fault); the results of the steps since the last sva_sync are invalid");
        }
    }
    return 0;
}
static int ks_scratch(hipStream_t st, KsScratch* out) {
    std::lock_guard<std::mutex> lk(g_ks_mu);
    auto it = g_ks.find(st);
    if (it != g_ks.end()) { *out = it->second; return 0; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { *out = KsScratch(); return 0; }   // no allocation inside a capture
    KsScratch k;
    // UNCACHED device memory: partial tiles and counters bypass the (per-XCD, mutually incoherent) L2s, so the hand-off
    // needs no agent-scope fence -- an agent-scope release / acquire per workgroup writes back / invalidates the whole
    // L2 and serialises (measured: + 0.37 us per workgroup)
    SVA_HIP(hipExtMallocWithFlags((void**)&k.ws, KS_WS_FLOATS * 8, hipDeviceMallocUncached));
    SVA_HIP(hipMemset(k.ws, 0, KS_WS_FLOATS * 8));
    SVA_HIP(hipExtMallocWithFlags((void**)&k.cnt, KS_CNT * sizeof(unsigned), hipDeviceMallocUncached));
    SVA_HIP(hipMemset(k.cnt, 0, KS_CNT * sizeof(unsigned)));
    g_ks[st] = k;
    *out = k;
    return 0;
}

// split-K scratch of a stream, allocated ahead of time: a launch inside a stream capture cannot allocate and would silently fall
// back to an unsplit launch -- a different summation order from the eager launch of the same shape
int conv_gemm_prepare_stream(hipStream_t st) {
    KsScratch k;
    SVA_TRY_RC(ks_scratch(st, &k));
    return 0;
}

template <int MT, int NT, int KW, int D, int AOP>
static int launch_skinny_op(const ConvGemm& g, hipStream_t st) {
    const size_t smem = ((size_t)KW * MT * NT * 256 + (AOP == 2 ? KW * MT * 16 : 0) + (AOP == 3 ? (size_t)16 * MT * (g.Cin + 4) : 0)) * sizeof(float);
    ConvGemmGroup gg;
    if (t_group) gg = *t_group; else gg.g[0] = g;
    dim3 grid((g.N + 16 * NT - 1) / (16 * NT), (g.M + 16 * MT - 1) / (16 * MT), gg.n);
    if (gg.n == 1) {
        int Z = AOP == 2 ? 1 : t_ksplit;                  // (the fused RMSNorm needs the whole row in one workgroup)
        const long nkb = (long)g.taps * g.Cin / 16;
        if (Z > nkb) Z = (int)nkb;
        const size_t tiles = (size_t)grid.x * grid.y;
        KsScratch k;
        if (Z > 1 && (tiles >= (size_t)KS_ERR_WORD || (size_t)Z * tiles * MT * NT * 256 > KS_WS_FLOATS)) Z = 1;
        if (Z > 1) { SVA_TRY_RC(ks_scratch(st, &k)); if (!k.ws) Z = 1; }
        gg.g[0].ksplit = Z; gg.g[0].ks_ws = k.ws; gg.g[0].ks_cnt = k.cnt;
        grid.z = Z;
    }
    static DeviceOnce attr;
    if (attr.needed() && smem > 48 * 1024) {
        SVA_HIP(hipFuncSetAttribute((const void*)skinny_gemm_kernel<MT, NT, KW, D, AOP>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));
        attr.done();
    }
    hipLaunchKernelGGL((skinny_gemm_kernel<MT, NT, KW, D, AOP>), grid, dim3(64 * KW), smem, st, gg);
    return 0;
}
template <int MT, int NT, int KW, int D>
static int launch_skinny(const ConvGemm& g, hipStream_t st) {
    if (g.dw_wT) {
        if constexpr (MT == 1 && NT == 1) return launch_skinny_op<MT, NT, KW, D, 3>(g, st);
        else { set_error("conv_gemm: the fused ConvNeXt prologue runs on one 16-row tile"); return -1; }
    }
    if (g.rms_w) return launch_skinny_op<MT, NT, KW, D, 2>(g, st);
    if (g.a_silu) return launch_skinny_op<MT, NT, KW, D, 1>(g, st);
    return launch_skinny_op<MT, NT, KW, D, 0>(g, st);
}

#ifndef SVA_D18
#define SVA_D116 4
#define SVA_D18 6
#define SVA_D14 8
#define SVA_D28 4
#define SVA_D24 6
#define SVA_D48 3
#define SVA_D44 4
#endif
template <int NT>
static int launch_cfg(const ConvGemm& g, hipStream_t st, int mt, int kw) {
    if constexpr (NT == 4) {      // 64-column workgroups: rows x K-split waves in {16, 32, 64} x {4, 8}
        switch (mt) {
            case 1: return kw == 8 ? launch_skinny<1, 4, 8, 4>(g, st) : launch_skinny<1, 4, 4, 4>(g, st);
            case 2: return kw == 8 ? launch_skinny<2, 4, 8, 3>(g, st) : launch_skinny<2, 4, 4, 3>(g, st);
            default: return kw == 8 ? launch_skinny<4, 4, 8, 2>(g, st) : launch_skinny<4, 4, 4, 2>(g, st);
        }
    }
    switch (mt) {
        case 1:
            if (kw == 16) return launch_skinny<1, NT, 16, SVA_D116>(g, st);
            if (kw == 8) return launch_skinny<1, NT, 8, SVA_D18>(g, st);
            return launch_skinny<1, NT, 4, SVA_D14>(g, st);
        case 2:
            if (kw == 8) return launch_skinny<2, NT, 8, SVA_D28>(g, st);
            return launch_skinny<2, NT, 4, SVA_D24>(g, st);
        case 3:
            if (kw == 8) return launch_skinny<3, NT, 8, SVA_D28>(g, st);
            return launch_skinny<3, NT, 4, SVA_D44>(g, st);
        default:
            if (kw == 8) return launch_skinny<4, NT, 8, SVA_D48>(g, st);
            return launch_skinny<4, NT, 4, SVA_D44>(g, st);
    }
}
// Heuristic choice of (rows per workgroup = 16*MT, K-split waves KW) for the small-M kernel.
static void skinny_heuristic(const ConvGemm& g, int NT, int* mt_out, int* kw_out) {
    const int mt_total = (g.M + 15) / 16;
    const long nk = (long)g.taps * g.Cin / 16;
    const long cols = (g.N + 16 * NT - 1) / (16 * NT);
    int mt = mt_total < 4 ? mt_total : 4;
    auto blocks = [&](int m) { return cols * ((mt_total + m - 1) / m); };
    // Every row tile of a column block re-reads that block's weights.  Weight-heavy problems (AR layers at M = 64..128:
    // the panel comes from HBM) keep the tallest workgroup; light ones (encoder at M = 128..160: the panel sits in L2)
    // trade re-reads for >= ~1.5 workgroups per CU.  Tuned with tools/gemm_sweep4.py.
    const bool heavy = (long)g.N * g.taps * g.Cin * 4 > (8L << 20);
    if (heavy) { while (mt > 1 && blocks(mt) * 4 < 512) mt = mt > 2 ? 2 : 1; }
    else       { while (mt > 1 && blocks(mt) < 384) mt = mt > 2 ? 2 : 1; }
    int kw = 4;
    while (kw < 8 && blocks(mt) * kw < 2048 && nk / (2 * kw) >= 2) kw *= 2;
    if (kw == 8 && blocks(mt) * 8 < 512 && nk / 32 >= 2 && mt == 1) kw = 16;     // a handful of column blocks: split K deeper
    *mt_out = mt; *kw_out = kw;
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_t(const ConvGemm& g, hipStream_t st) {
    constexpr size_t smem_ab = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    constexpr size_t smem_c = (size_t)BM * (BN + 4) * sizeof(float);          // epilogue staging tile reuses the buffers
    constexpr size_t smem = smem_ab > smem_c ? smem_ab : smem_c;
    static DeviceOnce attr_set;
    if (attr_set.needed() && smem > 48 * 1024) {
        SVA_HIP(hipFuncSetAttribute((const void*)conv_gemm_kernel<BM, BN, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    ConvGemmGroup gg;
    if (t_group) gg = *t_group; else gg.g[0] = g;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, gg.n);
    gg.xcd_swz = xcd_swizzle_for(grid.x, grid.y);
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, BK>), grid, dim3(64 * WM * WN), smem, st, gg);
    return 0;
}

// One dispatch decision: kind 0 = small-M K-split kernel (a = rows/16 per workgroup, b = K-split waves, c = 16-column
// tiles per wave); kind 1 = LDS-tiled kernel (a: 0 = 64x64, 1 = 128x128, 2 = 128x32, 3 = 256x16, 4 = 128x64, 5 = 64x128,
// 6 = 256x64, 7 = 256x128 on 8 waves; 4..7 are reached through the autotuner only); kind 6 = weight-streaming kernel (gemm_stream.hip:
// a = 16-row tiles, b = K-split waves, c = 16-column tiles per workgroup; reads the fragment-major weight copy when the problem carries one).
struct Choice { int kind, a, b, c; int z = 1; };      // z: grid-level K split of the small-M kernel
static thread_local int t_planes_mode = -1;              // set by launch_choice when the planes kernel took a kind-4 choice
static thread_local int t_last_kind = -1;                // kernel family of this thread's latest dispatch (bench.py: per-pipe roofline)

// Tile variant of the planes kernel (gemm_planes.hip) for a problem.  The candidates that ever win on the encoder's / vocoder's shapes
// (tools/planes_bench.py, profiles/r04_planes_bench.txt) are 128 x 128 (two workgroups per CU) and 256 x 128 (eight waves, one per CU,
// two thirds of the operand traffic); which one is a matter of how the tile count quantises over the 256 CUs.  In units of the time
// T a CU needs for one 128 x 128 tile's worth of work when it is full: a round of 512 small tiles costs 2 T, a last round of <= 256
// of them (one per CU) 1.3 T, a round of 256 large tiles 1.7 T.  The rule reproduces the measured winner of the two on all ten shapes.
struct PlanesRow { int M, N, K, variant; };
static const PlanesRow g_planes_table[] = {
#include "planes_table.inc"
};
static int planes_variant(const ConvGemm& g, int group_n) {
    // conv taps over A planes (the HiFiGAN levels' ResBlock convs, three branches per launch): only the LDS-DMA form reads them; its tile by
    // the output width and by whether 128 x 128 tiles would fill the chip
    if (g.Ap && (g.taps > 1 || group_n > 1 || g.cp_silu) && debug_options().planes_dma != 0) {
        bool ok = planes_dma_conv_supported(g);
        if (t_group) for (int i = 0; i < t_group->n; ++i) ok = ok && planes_dma_conv_supported(t_group->g[i]);
        if (ok) {
            const int dv = debug_options().voc_dma_variant;
            if (g.N % 128 == 0 && (long)((g.M + 127) / 128) * (g.N / 128) * group_n >= 192) return dv >= 9 && dv <= 14 && dv != 12 ? dv : 11;       // (its loader-wave form: 66 / 61 -> 60 / 55 us per launch at C = 128, 64 streams)
            return g.N == 64 && g.M * (long)group_n >= 3 * 8192 ? 13 : 14;
        }
    }
    bool dma_ok = planes_dma_gemm_supported(g) && debug_options().planes_dma != 0;
    if (group_n > 1) dma_ok = false;
    // measured winners for the encoder's shapes (tools/planes_tune.py -> planes_table.inc; every variant computes the same accumulation per
    // output, so the table is a speed choice only): the row with this (N, K) whose M is nearest, if within a quarter of it
    if (group_n == 1 && g.taps == 1) {
        const PlanesRow* best = nullptr;
        for (const PlanesRow& r : g_planes_table)
            if (r.N == g.N && r.K == g.Cin && (r.variant < 8 || dma_ok) &&
                (!best || std::abs(r.M - g.M) < std::abs(best->M - g.M))) best = &r;
        if (best && std::abs(best->M - g.M) * 4 <= g.M && (best->variant != 6 || g.M >= 256) &&
            (g.M >= 128 || best->variant == 2 || best->variant == 3 || best->variant >= 9))
            return best->variant >= 11 && debug_options().planes_lw == 0 ? 10 : best->variant;       // (A/B: the loader-wave forms off)
    }
    // both operands as planes, whole 128-column tiles, a shape outside the table: the persistent LDS-DMA form with 128 x 128 tiles and TWO
    // workgroups per CU (variant 10: one workgroup's epilogue runs under the other's K steps) -- it wins or ties on every encoder shape at 64
    // streams with the real epilogues (profiles/r05_planes_dma_bench.txt); 256 x 128 (8) never wins.  (Table vs this rule for the shapes
    // the table holds: encoder stage 4.41 / 5.13 / 8.50 ms against 4.67 / 5.28 / 8.68 at 48 / 64 / 128 streams, pipelined frames/s equal.)
    if (dma_ok && (long)((g.M + 127) / 128) * (g.N / 128) >= 64) return 10;
    if (g.N < 128) return g.M >= 128 ? 1 : 3;
    if (g.M < 128) return 2;
    const long wg0 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * group_n, wg6 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128) * group_n;
    if (wg0 < 224) return 3;                // too few 128 x 128 tiles for the chip: 64 x 64 (the mid-size shapes of profiles/r04_planes_bench_small.txt)
    if (g.M < 256) return 0;
    const long rem = wg0 % 512;
    const double t0 = (double)(wg0 / 512) * 2.0 + (rem == 0 ? 0.0 : rem <= 256 ? 1.3 : 2.0);
    const double t6 = (double)((wg6 + 255) / 256) * 1.7;
    if (t6 < t0) return 6;
    return wg0 <= 768 ? 7 : 0;              // up to a round and a half of tiles: the 8-wave form of the same tile (four waves per SIMD overlap its phases better)
}

static int launch_choice(const ConvGemm& g, hipStream_t st, const Choice& ch) {
    if (ch.kind == 4) {                 // fp32 on the bf16 matrix pipes, six-product split (gemm_split.hip), a = tile variant
        ConvGemmGroup gg;
        if (t_group) gg = *t_group; else gg.g[0] = g;
        if (ch.a >= 8) {                // the kernel fed from pre-split operand planes (gemm_planes.hip), its tile variant a - 8
            t_planes_mode = g.pmode + (ch.a - 8 >= 8 ? 2 : 0);          // (+ 2: its persistent LDS-DMA form)
            return launch_planes_gemm(gg, ch.a - 8, st);
        }
        SVA_CHECK(!g.Ap && !g.Cp, "conv_gemm: operand planes need the planes kernel");
        return launch_split_gemm(gg, ch.a, st);
    }
    if (ch.kind == 6) {                 // weight-streaming kernel (gemm_stream.hip): a = row tiles, b = K-split waves, c = column tiles per workgroup
        SVA_CHECK(!t_group && stream_gemm_supported(g), "conv_gemm: the weight-streaming kernel takes single problems");
        return launch_stream_gemm(g, g.Wk ? g.Wk : g.W, ch.a, ch.c, ch.b, g.Wk ? 2 : 0, 0, st);
    }
    if (ch.kind == 2) {                 // LDS-DMA ring kernel (gemm_pipe.hip), a = tile variant
        ConvGemmGroup gg;
        if (t_group) gg = *t_group; else gg.g[0] = g;
        return launch_pipe_gemm(gg, ch.a, st);
    }
    if (ch.kind == 0) {
        t_ksplit = ch.z;
        const int rc = ch.c == 4 ? launch_cfg<4>(g, st, ch.a, ch.b) : ch.c == 2 ? launch_cfg<2>(g, st, ch.a, ch.b) : launch_cfg<1>(g, st, ch.a, ch.b);
        t_ksplit = 1;
        return rc;
    }
    const int bk = g.Cin % 64 == 0 ? 64 : (g.Cin % 32 == 0 ? 32 : 16);
    switch (ch.a) {
        case 3: return launch_t<256, 16, 4, 1, 16>(g, st);
        case 2: return bk >= 32 ? launch_t<128, 32, 4, 1, 32>(g, st) : launch_t<128, 32, 4, 1, 16>(g, st);
        case 1: return bk >= 32 ? launch_t<128, 128, 2, 2, 32>(g, st) : launch_t<128, 128, 2, 2, 16>(g, st);
        case 4: return bk >= 32 ? launch_t<128, 64, 2, 2, 32>(g, st) : launch_t<128, 64, 2, 2, 16>(g, st);
        case 5: return bk >= 32 ? launch_t<64, 128, 2, 2, 32>(g, st) : launch_t<64, 128, 2, 2, 16>(g, st);
        case 6: return bk >= 32 ? launch_t<256, 64, 4, 1, 32>(g, st) : launch_t<256, 64, 4, 1, 16>(g, st);
        case 7: return bk >= 32 ? launch_t<256, 128, 4, 2, 32>(g, st) : launch_t<256, 128, 4, 2, 16>(g, st);       // 8 waves, 64x64 per wave
        default:
            if (bk == 64) return launch_t<64, 64, 2, 2, 64>(g, st);
            if (bk == 32) return launch_t<64, 64, 2, 2, 32>(g, st);
            return launch_t<64, 64, 2, 2, 16>(g, st);
    }
}

// tile variant of the ring kernel for an under-filled grid: the largest tile that still gives every CU of a partition work
static int pipe_variant(const ConvGemm& g) {
    auto tiles = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
    if (tiles(128, 64) >= 192) return 2;
    if (tiles(64, 64) >= 160) return 1;
    if (tiles(32, 64) >= 96 || g.N % 64 == 0) return 0;
    return 6;
}

static Choice heuristic_choice(const ConvGemm& g, bool c_vec) {
    // MFMA-bound problems outside the tuned table (batch sizes the tuning runs did not visit): the split-bf16 kernel, tile shape by
    // the rule the table shows -- wave-specialised 128x128 for narrow outputs with a long K, plain 128x128 otherwise, 64x64 for N < 128
    if (c_vec && split_gemm_supported(g) && g.M >= 2048 && g.N >= 64) {
        if (g.N < 128) return Choice{4, 3, 0, 0};
        const long K = (long)g.taps * g.Cin;
        return Choice{4, (g.N <= 512 && K >= 1024) ? 4 : 0, 0, 0};
    }
    {
        const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
        if (c_vec && pipe_gemm_supported(g) && g.M >= 32 && g.N >= 32 && t64 < 1024) return Choice{2, pipe_variant(g), 0, 0};
    }
    // the K tile is as deep as Cin allows (bytes in flight per workgroup hide the L2/HBM latency of the register-staged
    // pipeline); 128x128 tiles only when they still fill the 256 CUs
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    // under-filled grids (fewer than ~1 tiled workgroup per CU): the barrier-free K-split kernel keeps far more
    // loads in flight per CU than the LDS-staged one and pays for it with extra L2 reads, which are cheap there
    const long tiles64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
    if (g.M <= 64 || (tiles64 < 256 && g.N >= 64) || !c_vec) {      // (the tiled epilogue needs 16-byte aligned C rows)
        // two 16-column tiles per wave halve the A re-reads; worth it once the A panel dominates the L2 traffic
        const bool nt2 = g.w13 || (g.N % 32 == 0 && g.M >= 512 && (long)g.M * g.N >= 256L * 1024);
        Choice ch{0, 1, 4, nt2 ? 2 : 1};
        skinny_heuristic(g, ch.c, &ch.a, &ch.b);
        // A workgroup ingests 16*(MT + NT) rows of K floats and a CU sustains only ~40 GB/s of loads (tools/gemm_kscale.py:
        // time grows with K alone), so when the tiles do not cover the 256 CUs the K axis is split over more workgroups
        const long wgs = (long)((g.N + 16 * ch.c - 1) / (16 * ch.c)) * (((g.M + 15) / 16 + ch.a - 1) / ch.a);
        const long nkb = (long)g.taps * g.Cin / 16;
        while (ch.z < 8 && wgs * ch.z * 2 <= 256 && nkb / (2L * ch.z * ch.b) >= 2) ch.z *= 2;
        return ch;
    }
    if (g.N <= 16 && !g.w13) return Choice{1, 3, 0, 0};
    if (g.N <= 32) return Choice{1, 2, 0, 0};
    // 128x128 tiles only when their last (partial) round over the 256 CUs does not cost more than the lower operand
    // reuse of 64x64 tiles (e.g. 320 big tiles = 2 rounds for 1.25 rounds of work)
    if (big >= 256 && ((big + 255) / 256) * 4.0 <= ((tiles64 + 255) / 256) * 1.25) return Choice{1, 1, 0, 0};
    return Choice{1, 0, 0, 0};
}

static std::mutex g_tune_mu;
static std::map<std::array<int, 6>, Choice> g_tune;      // (M, N, K, taps, epilogue / prologue flags, stride)
// compiled-in per-shape choices: the outcome of an offline tuning run (tools/make_tune_table.py -> tune_table.inc), so the
// default dispatch is a pure function of the problem shape
struct TuneRow { int key[6]; int kind, a, b, c, z; };
static const TuneRow kTuneTable[] = {
#include "tune_table.inc"
    {{0, 0, 0, 0, 0, 0}, -1, 0, 0, 0, 1}};
static const std::map<std::array<int, 6>, Choice>& static_table() {
    static const std::map<std::array<int, 6>, Choice> m = [] {
        std::map<std::array<int, 6>, Choice> t;
        if (!debug_options().tune_table) return t;
        for (const TuneRow& r : kTuneTable) {
            if (r.kind < 0 || !((debug_options().tune_kinds >> r.kind) & 1)) continue;
            Choice c{r.kind, r.a, r.b, r.c};
            c.z = r.z;
            t[{r.key[0], r.key[1], r.key[2], r.key[3], r.key[4], r.key[5]}] = c;
        }
        return t;
    }();
    return m;
}
// SVA_DEBUG=tune_dump=<file>: the shapes tuned by this process (autotune=1) are appended as table rows when the library unloads
static void dump_tune_table() {
    if (debug_options().tune_dump.empty()) return;
    FILE* f = fopen(debug_options().tune_dump.c_str(), "a");
    if (!f) return;
    for (const auto& kv : g_tune)
        fprintf(f, "{{%d, %d, %d, %d, %d, %d}, %d, %d, %d, %d, %d},\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.first[4], kv.first[5],
                kv.second.kind, kv.second.a, kv.second.b, kv.second.c, kv.second.z);
    fclose(f);
}
static const int g_tune_dump_registered = (atexit(dump_tune_table), 0);
static float* g_tune_c = nullptr;
static size_t g_tune_elems = 0;

static int launch_conv_gemm_impl(const ConvGemm& g, hipStream_t st, int group_n);
int launch_conv_gemm(const ConvGemm& g, hipStream_t st) { return launch_conv_gemm_impl(g, st, 1); }

int launch_conv_gemm_group(const ConvGemm* gs, int n, hipStream_t st) {
    SVA_CHECK(n >= 1 && n <= 3, "conv_gemm_group: 1..3 members");
    if (n == 1) return launch_conv_gemm_impl(gs[0], st, 1);
    ConvGemmGroup gg;
    gg.n = n;
    int lead = 0;
    for (int i = 0; i < n; ++i) {
        const ConvGemm& a = gs[i];
        const ConvGemm& r = gs[0];
        SVA_CHECK(a.M == r.M && a.T == r.T && a.N == r.N && a.Cin == r.Cin && a.stride == r.stride && a.a_silu == r.a_silu && a.w13 == r.w13 &&
                  a.act == r.act && a.accumulate == r.accumulate && !a.rms_w && a.ldc % 4 == r.ldc % 4 && (a.res != nullptr) == (r.res != nullptr) &&
                  (a.gamma != nullptr) == (r.gamma != nullptr) && (a.bias != nullptr) == (r.bias != nullptr),
                  "conv_gemm_group: members must share shape and epilogue");
        SVA_CHECK(a.lda % 4 == 0 && a.a_off % 4 == 0 && a.a_bstride % 4 == 0 && a.c_off % 4 == r.c_off % 4 && a.c_bstride % 4 == r.c_bstride % 4 &&
                  (!a.res || (a.ldr % 4 == r.ldr % 4 && a.r_off % 4 == r.r_off % 4 && a.r_bstride % 4 == r.r_bstride % 4)),
                  "conv_gemm_group: alignment classes must match");
        gg.g[i] = a;
        if (a.taps > gs[lead].taps) lead = i;
    }
    t_group = &gg;                       // the dispatch decision is taken for (and timed on) the member with the longest K
    const int rc = launch_conv_gemm_impl(gs[lead], st, n);
    t_group = nullptr;
    return rc;
}

static int launch_conv_gemm_impl(const ConvGemm& g, hipStream_t st, int group_n) {
    SVA_CHECK(g.Cin % 16 == 0 && g.Cin > 0, "conv_gemm: Cin must be a multiple of 16");
    // decode-sized linear layers of an fp16-weight AR: stream the fp16 weights (half the bytes of the fp32 copy) through the f16 pipes
    if (g.Wh && group_n == 1 && g.M <= 256 && debug_options().f16_weights && (f16w_gemm_validated_compiler() || debug_options().f16_weights == 2) &&
        f16w_gemm_supported(g)) {
        t_last_kind = 5;
        return launch_f16w_gemm(g, st);
    }
    SVA_CHECK(g.lda % 4 == 0 && (g.a_off % 4) == 0 && (g.a_bstride % 4) == 0, "conv_gemm: A must be float4-aligned");
    const bool c_vec = g.N % 4 == 0 && g.ldc % 4 == 0 && g.c_off % 4 == 0 && g.c_bstride % 4 == 0 &&
                       (!g.res || (g.ldr % 4 == 0 && g.r_off % 4 == 0 && g.r_bstride % 4 == 0));
    SVA_CHECK(g.M > 0 && g.N > 0 && g.T > 0, "conv_gemm: empty problem");
    if (g.w13) SVA_CHECK(g.N % 32 == 0, "conv_gemm: w13 needs N % 32 == 0");
    if (g.rms_w) SVA_CHECK(g.taps == 1 && !g.a_silu && conv_gemm_can_fuse_rms(g.M, g.N), "conv_gemm: fused RMSNorm needs taps == 1 on the small-M path");
    if (g.dw_wT) SVA_CHECK(g.taps == 1 && g.M <= 16 && g.Cin <= 512 && !g.a_silu && !g.rms_w && !g.w13 && group_n == 1 && g.dw_b && g.ln_w && g.ln_b,
                           "conv_gemm: the fused ConvNeXt prologue needs taps == 1, M <= 16, Cin <= 512");
    Choice ch = heuristic_choice(g, c_vec);
    const unsigned long long key_flags = (unsigned long long)(g.a_silu ? 1 : 0) | (g.rms_w ? 2 : 0) | (g.w13 ? 4 : 0) | (c_vec ? 8 : 0) | (g.accumulate ? 16 : 0) |
                                         (group_n > 1 ? 32 : 0) | (g.dw_wT ? 64 : 0);
    {
        const auto& tab = static_table();
        auto it = tab.find({g.M, g.N, g.taps * g.Cin, g.taps, (int)key_flags, g.stride});
        // a decode-sized row count between two tabulated ones (40 or 44 streams: the table holds 36 and 48) takes the choice of the next larger one within
        // 1.5 x -- every kernel handles partial row tiles, and the heuristic's pick there measured 20 % slower (AR stage 4.7 ms at 40 / 44 streams, 3.8 at 48)
        // (the scan is memoised per shape -- ADVICE r05: an untabulated shape paid up to M / 2 map look-ups on every launch)
        if (it == tab.end() && g.M >= 8 && g.M <= 512) {
            static std::mutex memo_mu;
            static std::map<std::array<int, 6>, int> memo;            // shape -> the tabulated row count it borrows (0: none)
            const std::array<int, 6> key{g.M, g.N, g.taps * g.Cin, g.taps, (int)key_flags, g.stride};
            int borrowed = -1;
            {
                std::lock_guard<std::mutex> lk(memo_mu);
                auto mi = memo.find(key);
                if (mi != memo.end()) borrowed = mi->second;
            }
            if (borrowed < 0) {
                borrowed = 0;
                for (int m = g.M + 1; m <= g.M + g.M / 2; ++m)
                    if (tab.find({m, g.N, g.taps * g.Cin, g.taps, (int)key_flags, g.stride}) != tab.end()) { borrowed = m; break; }
                std::lock_guard<std::mutex> lk(memo_mu);
                memo[key] = borrowed;
            }
            if (borrowed > 0) it = tab.find({borrowed, g.N, g.taps * g.Cin, g.taps, (int)key_flags, g.stride});
        }
        if (it != tab.end()) {
            // the table is keyed by shape only; the pipelined / split kernels also need aligned operands (a seam such as sva_op_conv can
            // present a tuned shape with other strides): keep the tuned choice only if its kernel accepts THIS problem
            const Choice& tc = it->second;
            const bool ok = (tc.kind == 2) ? (c_vec && pipe_gemm_supported(g)) : (tc.kind == 4) ? (c_vec && split_gemm_supported(g)) :
                            (tc.kind == 6) ? (group_n == 1 && stream_gemm_supported(g) && (g.M + 16 * tc.a - 1) / (16 * tc.a) * (long)((g.N + 16 * tc.c - 1) / (16 * tc.c)) < 65536) :
                            (tc.kind == 1 ? c_vec || tc.a == 0 : true);
            if (ok) ch = tc;
        }
    }
    // Deterministic by default: the kernel / configuration of a problem shape comes from the compiled-in table (tune_table.inc,
    // generated offline from a logged tuning run) or the heuristic -- never from wall-clock measurements of this process, so two
    // processes, ranks or runs sum in the same order.  SVA_DEBUG=autotune=1 re-enables the timed search (tools/make_tune_table.py uses it).
    static const bool tune = debug_options().autotune != 0;
    if (tune && !g.Ap && !g.Cp) {          // (a problem whose operands are planes has one kernel family: its variant comes from planes_table.inc)
        // Shape-keyed autotune: the first eager launch of a problem shape times the candidate kernels / configurations on
        // the real operands with the output redirected to scratch, and keeps a candidate only if it beats the heuristic
        // by > 7 %.  Launches inside a stream capture (and shapes first seen there) use the heuristic.
        const unsigned long long flags = (unsigned long long)(g.a_silu ? 1 : 0) | (g.rms_w ? 2 : 0) | (g.w13 ? 4 : 0) | (c_vec ? 8 : 0) | (g.accumulate ? 16 : 0) |
                                         (group_n > 1 ? 32 : 0) | (g.dw_wT ? 64 : 0);
        const std::array<int, 6> key = {g.M, g.N, g.taps * g.Cin, g.taps, (int)flags, g.stride};
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tune.find(key);
        if (it != g_tune.end()) ch = it->second;
        else {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
                const int ldc = g.w13 ? g.N / 2 : g.N;
                const size_t need = (size_t)g.M * ldc * (size_t)group_n;
                if (need > g_tune_elems) {
                    if (g_tune_c) (void)hipFree(g_tune_c);
                    SVA_HIP(hipMalloc((void**)&g_tune_c, need * sizeof(float)));
                    g_tune_elems = need;
                }
                ConvGemm t = g;
                t.C = g_tune_c; t.c_bstride = (long)g.T * ldc; t.c_off = 0; t.ldc = ldc;
                // a group is timed as a group, every member's output redirected to its own scratch slab
                const ConvGemmGroup* real_group = t_group;
                ConvGemmGroup tg;
                if (real_group) {
                    tg = *real_group;
                    for (int i = 0; i < tg.n; ++i) {
                        tg.g[i].C = g_tune_c + (size_t)i * g.M * ldc; tg.g[i].c_bstride = (long)g.T * ldc; tg.g[i].c_off = 0; tg.g[i].ldc = ldc;
                    }
                    t_group = &tg;
                }
                hipEvent_t e0, e1;
                SVA_HIP(hipEventCreate(&e0)); SVA_HIP(hipEventCreate(&e1));
                // measured alone on the device (other streams drained first) and as the better of two batches: the pick should
                // not depend on what happened to run beside the probe
                SVA_HIP(hipDeviceSynchronize());
                auto time_choice = [&](const Choice& c, float* ms) -> int {
                    SVA_TRY_RC(launch_choice(t, st, c));
                    float best_ms = 1e30f;
                    for (int rep = 0; rep < 2; ++rep) {
                        SVA_HIP(hipEventRecord(e0, st));
                        for (int r = 0; r < 5; ++r) SVA_TRY_RC(launch_choice(t, st, c));
                        SVA_HIP(hipEventRecord(e1, st));
                        SVA_HIP(hipEventSynchronize(e1));
                        float m = 0.f;
                        SVA_HIP(hipEventElapsedTime(&m, e0, e1));
                        if (m < best_ms) best_ms = m;
                    }
                    *ms = best_ms;
                    return 0;
                };
                std::vector<Choice> cand;
                const long tiles64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
                const bool must_skinny = g.rms_w || g.dw_wT || !c_vec;
                if (must_skinny || tiles64 < 1024) {
                    const int mt_total = (g.M + 15) / 16;
                    const long nk = (long)g.taps * g.Cin / 16;
                    for (int nt = 1; nt <= 4; nt *= 2) {
                        if (g.w13 && nt == 1) continue;
                        if (nt == 2 && (g.N % 32 != 0 || g.dw_wT)) continue;
                        if (nt == 4 && (g.N % 64 != 0 || g.dw_wT || g.M < 32)) continue;       // 64-column workgroups: a third of the operand reads per output
                        const int mts[3] = {1, 2, 4}, kws[3] = {4, 8, 16};
                        for (int a = 0; a < 3; ++a)
                            for (int b2 = 0; b2 < 3; ++b2) {
                                if (mts[a] > mt_total || (mts[a] >= 2 && kws[b2] == 16) || nk / kws[b2] < 1) continue;
                                if (nt == 4 && kws[b2] == 16) continue;
                                cand.push_back(Choice{0, mts[a], kws[b2], nt});
                                if (g.rms_w || group_n > 1) continue;
                                const long wgs = (long)((g.N + 16 * nt - 1) / (16 * nt)) * ((mt_total + mts[a] - 1) / mts[a]);
                                for (int z = 2; z <= 8; z *= 2)
                                    if (wgs * z <= 512 && nk / ((long)z * kws[b2]) >= 1) cand.push_back(Choice{0, mts[a], kws[b2], nt, z});
                            }
                    }
                }
                if (!must_skinny) {
                    if (g.N > 32) cand.push_back(Choice{1, 0, 0, 0});
                    if (g.M >= 128 && g.N >= 128) cand.push_back(Choice{1, 1, 0, 0});
                    if (g.M >= 128 && g.N >= 64) cand.push_back(Choice{1, 4, 0, 0});
                    if (g.M >= 64 && g.N >= 128) cand.push_back(Choice{1, 5, 0, 0});
                    if (g.M >= 256 && g.N >= 64) cand.push_back(Choice{1, 6, 0, 0});
                    if (g.M >= 256 && g.N >= 128) cand.push_back(Choice{1, 7, 0, 0});
                    if (g.N <= 64) cand.push_back(Choice{1, 2, 0, 0});
                    if (g.N <= 16 && !g.w13) cand.push_back(Choice{1, 3, 0, 0});
                }
                if (group_n == 1 && stream_gemm_supported(g) && g.M <= 512 && !g.dw_wT && g.N >= 16) {
                    // weight-streaming kernel: (row tiles, column tiles, K-split waves) per workgroup
                    const int mt_total = (g.M + 15) / 16;
                    const int cfgs[15][3] = {{1, 1, 4}, {1, 1, 8}, {1, 1, 16}, {2, 1, 4}, {2, 1, 8}, {2, 1, 16}, {4, 1, 4}, {4, 1, 8},
                                             {1, 2, 4}, {1, 2, 8}, {1, 2, 16}, {2, 2, 4}, {2, 2, 8}, {4, 2, 4}, {4, 2, 8}};
                    for (const auto& cf : cfgs) {
                        if (cf[0] > mt_total || (g.w13 && cf[1] != 2) || (cf[1] == 2 && g.N % 32 != 0)) continue;
                        cand.push_back(Choice{6, cf[0], cf[2], cf[1]});
                    }
                }
                if (c_vec && pipe_gemm_supported(g) && g.M >= 32 && g.N >= 32)
                    for (int v = 0; v <= 6; ++v) {
                        if (v == 4 && (g.M < 128 || g.N < 128)) continue;
                        if ((v == 2 && g.M < 128) || ((v == 3 || v == 5) && g.N < 128)) continue;
                        cand.push_back(Choice{2, v, 0, 0});
                    }
                if (c_vec && split_gemm_supported(g) && g.M >= 64 && g.N >= 64)
                    for (int v = 0; v <= 4; ++v) {
                        if ((v == 0 || v == 1 || v == 4) && g.M < 128) continue;
                        if ((v == 0 || v == 2 || v == 4) && g.N < 128) continue;
                        cand.push_back(Choice{4, v, 0, 0});
                    }
                float base = 0.f;
                SVA_TRY_RC(time_choice(ch, &base));
                float best = base * 0.93f;
                for (const Choice& c : cand) {
                    if (c.kind == ch.kind && c.a == ch.a && c.b == ch.b && c.c == ch.c && c.z == ch.z) continue;
                    float ms = 0.f;
                    SVA_TRY_RC(time_choice(c, &ms));
                    if (ms < best) { best = ms; ch = c; }
                }
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
                t_group = real_group;
                static const bool tlog = debug_options().tune_log != 0;
                if (tlog)
                    fprintf(stderr, "[sva tune] M=%d N=%d K=%d taps=%d flags=%llu: heuristic %.1f us -> kind %d (%d,%d,%d) z%d %.1f us\n", g.M, g.N,
                            g.taps * g.Cin, g.taps, flags, base * 200.f, ch.kind, ch.a, ch.b, ch.c, ch.z, (best < base * 0.93f ? best : base) * 200.f);
                g_tune[key] = ch;
            }
        }
    }
    // Weights that carry pre-split planes (gemm_planes.hip).  fp16 x 1 (a voc_dtype = 1 vocoder): one product per block instead of the
    // six or eight of any fp32-grade kernel -- every problem with enough rows to fill its tiles.  fp32-grade planes (fp16 x 2, bf16 x 3):
    // where the split kernels are the choice anyway and the batch is large enough for the 128-row tiles to fill the chip (measured:
    // with its tile variant from the measured table it beats the tuned in-loop split kernels on 38 of 40 mid-size shapes, by 5-60 %:
    // profiles/r04_old_vs_planes.txt (full chip); from 3072 rows = 24 streams -- below that the pipelined mode runs the encoder on a CU partition the old table was tuned for: 16 streams -2.6 %, 24 / 32 / 48 streams +1 / +1 / +8 %).  A
    // problem whose operands only exist as planes has no other kernel.
    {
        bool planes = c_vec;
        if (t_group) { for (int i = 0; i < t_group->n; ++i) planes = planes && planes_gemm_supported(t_group->g[i]); }
        else planes = planes && planes_gemm_supported(g);
        // (and the few 2048+-row problems of a 64-stream batch that the table gives to the f32-MFMA kernels: the C = 256 HiFiGAN level's
        // grouped convs -- 5.6 GFLOP per launch at ~60 TF/s there)
        const double gflop = 2e-9 * g.M * (double)g.N * g.taps * g.Cin * group_n;
        const bool want = g.Ap || g.Cp ||
                          (g.pmode == PLANES_H1 ? g.M >= 1024 && g.N >= 32 : g.N < 64 ? false : ((ch.kind == 4 && g.M >= 3072) || (ch.kind != 4 && g.M >= 2048 && g.N >= 128 && gflop >= 2.0)));
        if (planes && want) ch = Choice{4, 8 + planes_variant(g, group_n), 0, 0};
        else SVA_CHECK(!g.Ap && !g.Cp, "conv_gemm: operand planes handed to a problem the planes kernel does not take");
    }
    t_planes_mode = -1;
    SVA_TRY_RC(launch_choice(g, st, ch));
    t_last_kind = t_planes_mode >= 0 ? 6 + t_planes_mode : ch.kind;       // 7 / 8: planes kernel in H3 / H1, 9 / 10: its LDS-DMA form
    SVA_HIP(hipGetLastError());
    return 0;
}

int conv_gemm_last_kind() { return t_last_kind; }

// test hook: run one specific dispatch choice (kind 0: a = rows/16, b = K split, c = column tiles; kind 1: a = tile variant)
int launch_conv_gemm_choice(const ConvGemm& g, hipStream_t st, int kind, int a, int b, int c) {
    SVA_CHECK(g.Cin % 16 == 0 && g.lda % 4 == 0, "conv_gemm_choice: alignment");
    if (kind == 2) {
        SVA_CHECK(pipe_gemm_supported(g) && a >= 0 && a <= 6, "conv_gemm_choice: the pipelined kernel needs Cin % 64 == 0 and 16-byte aligned operands");
        SVA_TRY_RC(launch_choice(g, st, Choice{2, a, 0, 0}));
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (kind == 6) {                    // the planes kernel (gemm_planes.hip), a = its tile variant
        SVA_CHECK(planes_gemm_supported(g) && a >= 0 && a <= 12 && a != 8 && (a < 8 || planes_dma_gemm_supported(g)) && g.N % 4 == 0 && g.ldc % 4 == 0, "conv_gemm_choice: the planes kernel needs weight planes, Cin % 32 == 0 and 16-byte aligned C rows");
        SVA_TRY_RC(launch_choice(g, st, Choice{4, 8 + a, 0, 0}));
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (kind == 4) {
        SVA_CHECK(split_gemm_supported(g) && a >= 0 && a <= 4 && g.N % 4 == 0 && g.ldc % 4 == 0, "conv_gemm_choice: the split-bf16 kernel needs Cin % 32 == 0 and 16-byte aligned C rows");
        SVA_TRY_RC(launch_choice(g, st, Choice{4, a, 0, 0}));
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (kind == 7) {                    // the weight-streaming kernel (gemm_stream.hip): a = row tiles, b = K-split waves, c = column tiles
        SVA_CHECK(stream_gemm_supported(g) && (a == 1 || a == 2 || a == 4) && (c == 1 || c == 2) && (b == 4 || b == 8 || (b == 16 && a * c <= 2)) && !(g.w13 && c != 2),
                  "conv_gemm_choice: bad weight-streaming configuration");
        SVA_TRY_RC(launch_choice(g, st, Choice{6, a, b, c}));
        SVA_HIP(hipGetLastError());
        return 0;
    }
    SVA_CHECK(kind == 0 || kind == 1, "conv_gemm_choice: kind");
    if (kind == 0) SVA_CHECK((a == 1 || a == 2 || a == 3 || a == 4) && (b == 4 || b == 8 || (b == 16 && a == 1)) &&
                                 (c == 1 || (c == 2 && g.N % 32 == 0) || (c == 4 && g.N % 64 == 0 && a != 3 && b != 16)),
                             "conv_gemm_choice: bad small-M configuration");
    else SVA_CHECK(a >= 0 && a <= 7 && g.N % 4 == 0 && g.ldc % 4 == 0, "conv_gemm_choice: bad tile variant");
    SVA_TRY_RC(launch_choice(g, st, Choice{kind, a, b, c}));
    SVA_HIP(hipGetLastError());
    return 0;
}
int launch_conv_gemm_choice_z(const ConvGemm& g, hipStream_t st, int a, int b, int c, int z) {
    SVA_CHECK(g.Cin % 16 == 0 && g.lda % 4 == 0, "conv_gemm_choice: alignment");
    SVA_CHECK((a == 1 || a == 2 || a == 3 || a == 4) && (b == 4 || b == 8 || (b == 16 && a == 1)) &&
                  (c == 1 || (c == 2 && g.N % 32 == 0) || (c == 4 && g.N % 64 == 0 && a != 3 && b != 16)) && z >= 1 && z <= 8,
              "conv_gemm_choice: bad small-M configuration");
    Choice ch{0, a, b, c};
    ch.z = z;
    SVA_TRY_RC(launch_choice(g, st, ch));
    SVA_HIP(hipGetLastError());
    return 0;
}

bool conv_gemm_can_fuse_rms(int M, int N) {
    const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    return M <= 64 || (tiles64 < 256 && N >= 64);
}

}  // namespace sva
