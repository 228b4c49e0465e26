// fp32 conv-GEMM on the bf16 matrix pipes: every fp32 operand element is split into three bf16 parts (x = hi + mid + lo exactly:
// 3 x 8 significand bits) when its tile is staged in LDS, and each 16 x 16 x 32 block is the sum of the six part products whose
// weight is >= 2^-16 of the leading one (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16.  The three dropped products are below 2^-24 relative -- one fp32 rounding -- so the result is
// fp32-grade (measured on the encoder: max |du| 1.1e-6 against the exact fp32 formulation, BSQ indices identical;
// tools/bf16split_feasibility.py), while six 16-cycle MFMAs replace eight 32-cycle v_mfma_f32_16x16x4_f32 per 32 k: 2.7x less
// matrix-pipe time.  Same problem description (ConvGemm: taps, SiLU prologue, fused epilogues, groups) as conv_gemm_kernel.
//
// LDS: one buffer of 3 planes per operand, [rows][32 bf16] with a 96-byte row stride (16-byte fragment reads, the lanes of every
// ds_read_b128 lane group on distinct banks); global -> registers prefetch of the next K tile runs under the MFMAs; two workgroups per CU.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sva_common.h"

namespace sva {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// two fp32 -> packed (hi, mid, lo) bf16 pairs; RNE conversions (v_cvt_pk_bf16_f32), residuals exact in fp32
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r1 = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1.x - __uint_as_float(mid << 16), r1.y - __uint_as_float(mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void split_gemm_kernel(const ConvGemmGroup gg) {
    constexpr int NTH = 256, BK = 32;
    static_assert(WM * WN == 4, "4 waves");
    const ConvGemm& g = gg.g[blockIdx.z];
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int RS = 48;                        // row stride in bf16 elements (96 bytes = 24 dwords: the 16 lanes of every ds_read_b128
                                                  // lane group -- rows {0-3,12-15} with k offset 0 and rows 4-11 with k offset 8, etc. -- land on 64 distinct banks)
    constexpr int F4R = BK / 4;                   // float4 per tile row (8)
    constexpr int RPP = NTH / F4R;                // rows covered per pass (32)
    constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* Ap = reinterpret_cast<unsigned short*>(smem);                  // [3][BM][RS]
    unsigned short* Bp = Ap + 3 * BM * RS;                                         // [3][BN][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tbx = blockIdx.x, tby = blockIdx.y;
    xcd_tile(gg.xcd_swz, gridDim.x, gridDim.y, tbx, tby);
    const int bm0 = tby * BM, bn0 = tbx * BN;
    const int kq = tid % F4R;
    const int lrow = tid / F4R;

    const float* a_ptr[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int m = bm0 + lrow + i * RPP;
        if (m > g.M - 1) m = g.M - 1;
        const int b = m / g.T, t = m - b * g.T;
        a_ptr[i] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + kq * 4;
    }
    const float* b_ptr[B_LD];
    const long Kt = (long)g.taps * g.Cin;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        int n = bn0 + lrow + i * RPP;
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
    f32x4 ra[A_LD], rb[B_LD];

    auto gload = [&](int kt) {
        const int tap = kt / kc_tiles;
        const int kc = (kt - tap * kc_tiles) * BK;
        const long aoff = (long)tap * g.dil * g.lda + kc;
        const long boff = (long)tap * g.Cin + kc;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + aoff);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + boff);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            f32x4 v = ra[i];
            if (g.a_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            unsigned h0, m0, l0, h1, m1, l1;
            split2(v.x, v.y, h0, m0, l0);
            split2(v.z, v.w, h1, m1, l1);
            unsigned short* d = Ap + (lrow + i * RPP) * RS + kq * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + BM * RS) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 2 * BM * RS) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const f32x4 v = rb[i];
            unsigned h0, m0, l0, h1, m1, l1;
            split2(v.x, v.y, h0, m0, l0);
            split2(v.z, v.w, h1, m1, l1);
            unsigned short* d = Bp + (lrow + i * RPP) * RS + kq * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + BN * RS) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 2 * BN * RS) = make_uint2(l0, l1);
        }
    };

    gload(0);
    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        lstore();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);            // in flight under the MFMAs of this tile
        // lane (fr, fk) supplies k = 8 fk .. 8 fk + 7 of row fr: the same rule on both operands
        const unsigned short* Ab = Ap + (wm * TM + fr) * RS + fk * 8;
        const unsigned short* Bb = Bp + (wn * TN + fr) * RS + fk * 8;
        bf16x8 ah[MI], am[MI], al[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * RS);
            am[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * RS + BM * RS);
            al[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * RS + 2 * BM * RS);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * RS);
            const bf16x8 bm = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * RS + BN * RS);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * RS + 2 * BN * RS);
            // small products first
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
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

// Wave-specialised form (512 threads): waves 0..3 are CONSUMERS (fragment reads + MFMAs, 64 x 64 each of a 128 x 128 tile), waves
// 4..7 are PRODUCERS (global loads, the hi/mid/lo split on the VALU, LDS stores into the other buffer).  One producer and one
// consumer wave share each SIMD, so the split arithmetic and the LDS writes run under the matrix pipe without relying on
// instruction scheduling inside one wave; LDS is double-buffered, one barrier per K tile.
template <int BM, int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void split_ws_kernel(const ConvGemmGroup gg) {
    constexpr int NTH = 512, BK = 32, WM = 2, WN = 2;
    const ConvGemm& g = gg.g[blockIdx.z];
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int RS = 48;
    constexpr int F4R = BK / 4;
    constexpr int RPP = 256 / F4R;                // rows covered per pass of the 256 producer threads
    constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
    constexpr int PLANE_A = BM * RS, PLANE_B = BN * RS, BUF = 3 * (PLANE_A + PLANE_B);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* lds = reinterpret_cast<unsigned short*>(smem);                 // [2][ A: 3 planes | B: 3 planes ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int cw = wave & 3;
    const int wm = cw / WN, wn = cw % WN;
    int tbx = blockIdx.x, tby = blockIdx.y;
    xcd_tile(gg.xcd_swz, gridDim.x, gridDim.y, tbx, tby);
    const int bm0 = tby * BM, bn0 = tbx * BN;
    const int ptid = tid & 255;
    const int kq = ptid % F4R;
    const int lrow = ptid / F4R;
    const int kc_tiles = g.Cin / BK;
    const int nk = g.taps * kc_tiles;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (producer) {
        const float* a_ptr[A_LD];
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int m = bm0 + lrow + i * RPP;
            if (m > g.M - 1) m = g.M - 1;
            const int b = m / g.T, t = m - b * g.T;
            a_ptr[i] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + kq * 4;
        }
        const float* b_ptr[B_LD];
        const long Kt = (long)g.taps * g.Cin;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            int n = bn0 + lrow + i * RPP;
            if (n > g.N - 1) n = g.N - 1;
            b_ptr[i] = g.W + (long)n * Kt + kq * 4;
        }
        // two register sets: the tile stored in iteration kt was requested two iterations earlier (the A operand streams from HBM:
        // one K tile of MFMAs is shorter than its latency under load)
        f32x4 ra[2][A_LD], rb[2][B_LD];
        auto gload = [&](int kt, f32x4 (&xa)[A_LD], f32x4 (&xb)[B_LD]) {
            const int tap = kt / kc_tiles;
            const int kc = (kt - tap * kc_tiles) * BK;
            const long aoff = (long)tap * g.dil * g.lda + kc;
            const long boff = (long)tap * g.Cin + kc;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) xa[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + aoff);
#pragma unroll
            for (int i = 0; i < B_LD; ++i) xb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + boff);
        };
        auto lstore = [&](int buf, const f32x4 (&xa)[A_LD], const f32x4 (&xb)[B_LD]) {
            unsigned short* Ap = lds + buf * BUF;
            unsigned short* Bp = Ap + 3 * PLANE_A;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                f32x4 v = xa[i];
                if (g.a_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                unsigned h0, m0, l0, h1, m1, l1;
                split2(v.x, v.y, h0, m0, l0);
                split2(v.z, v.w, h1, m1, l1);
                unsigned short* d = Ap + (lrow + i * RPP) * RS + kq * 4;
                *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(d + PLANE_A) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = make_uint2(l0, l1);
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const f32x4 v = xb[i];
                unsigned h0, m0, l0, h1, m1, l1;
                split2(v.x, v.y, h0, m0, l0);
                split2(v.z, v.w, h1, m1, l1);
                unsigned short* d = Bp + (lrow + i * RPP) * RS + kq * 4;
                *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(d + PLANE_B) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(d + 2 * PLANE_B) = make_uint2(l0, l1);
            }
        };
        gload(0, ra[0], rb[0]);
        if (nk > 1) gload(1, ra[1], rb[1]);
        lstore(0, ra[0], rb[0]);
        if (nk > 2) gload(2, ra[0], rb[0]);
        __syncthreads();
        // iteration kt stores tile kt + 1 (set (kt + 1) & 1) and requests tile kt + 3 into the set it just freed
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 1 < nk) {
                lstore(1, ra[1], rb[1]);
                if (kt + 3 < nk) gload(kt + 3, ra[1], rb[1]);
            }
            __syncthreads();
            if (kt + 1 < nk) {
                if (kt + 2 < nk) {
                    lstore(0, ra[0], rb[0]);
                    if (kt + 4 < nk) gload(kt + 4, ra[0], rb[0]);
                }
                __syncthreads();
            }
        }
    } else {
        const int fr = lane & 15, fk = lane >> 4;
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned short* Ap = lds + (kt & 1) * BUF;
            const unsigned short* Bp = Ap + 3 * PLANE_A;
            const unsigned short* Ab = Ap + (wm * TM + fr) * RS + fk * 8;
            const unsigned short* Bb = Bp + (wn * TN + fr) * RS + fk * 8;
            bf16x8 ah[MI], am[MI], al[MI], bh[NI], bm[NI], bl[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * RS);
                am[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * RS + PLANE_A);
                al[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * RS + 2 * PLANE_A);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {       // every fragment of the K tile is requested before the first MFMA (256 VGPRs per wave)
                bh[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * RS);
                bm[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * RS + PLANE_B);
                bl[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * RS + 2 * PLANE_B);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue ----
    // The accumulators (C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg) are staged
    // through LDS so that bias / residual reads and the C stores are whole 16-byte, row-contiguous accesses
    // (a lane-per-element epilogue touches a 64-byte segment per row per instruction and costs up to 30 % of a
    // K = 512 GEMM).  The last loop iteration ended with a barrier, so the A/B buffers are free to reuse.
    constexpr int CS = BN + 4;
    float* Cs = smem;                              // [BM][CS]
    const int col = lane & 15, rq = (lane >> 4) * 4;
    if (!producer) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(wm * TM + i * 16 + rq + r) * CS + wn * TN + j * 16 + col] = acc[i][j][r];
    }
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

template <int BM, int BN>
int launch_split_ws(const ConvGemmGroup& gg_in, hipStream_t st) {
    ConvGemmGroup gg = gg_in;
    const ConvGemm& g = gg.g[0];
    constexpr size_t smem_ab = (size_t)2 * 3 * (BM + BN) * 48 * sizeof(unsigned short);
    constexpr size_t smem_c = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t smem = smem_ab > smem_c ? smem_ab : smem_c;
    static DeviceOnce attr_set;
    if (attr_set.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)split_ws_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set.done();
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, gg.n);
    gg.xcd_swz = xcd_swizzle_for(grid.x, grid.y);
    hipLaunchKernelGGL((split_ws_kernel<BM, BN>), grid, dim3(512), smem, st, gg);
    return 0;
}

template <int BM, int BN, int WM, int WN>
int launch_split_t(const ConvGemmGroup& gg_in, hipStream_t st) {
    ConvGemmGroup gg = gg_in;
    const ConvGemm& g = gg.g[0];
    constexpr size_t smem_ab = (size_t)3 * (BM + BN) * 48 * sizeof(unsigned short);
    constexpr size_t smem_c = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t smem = smem_ab > smem_c ? smem_ab : smem_c;
    static DeviceOnce attr_set;
    if (attr_set.needed() && smem > 48 * 1024) {
        SVA_HIP(hipFuncSetAttribute((const void*)split_gemm_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set.done();
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, gg.n);
    gg.xcd_swz = xcd_swizzle_for(grid.x, grid.y);
    hipLaunchKernelGGL((split_gemm_kernel<BM, BN, WM, WN>), grid, dim3(256), smem, st, gg);
    return 0;
}

}  // namespace

// the tiled epilogue's conditions (16-byte aligned C rows) are the caller's; here: channels in whole 32-wide K tiles
bool split_gemm_supported(const ConvGemm& g) { return g.Cin % 32 == 0 && g.stride >= 1 && !g.rms_w && !g.dw_wT; }

// variant: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64 (one wave group does everything, two workgroups per CU);
// 4 = 128x128 wave-specialised (producer / consumer waves, 512 threads, double-buffered LDS)
int launch_split_gemm(const ConvGemmGroup& gg, int variant, hipStream_t st) {
    SVA_CHECK(split_gemm_supported(gg.g[0]), "split_gemm: unsupported problem");
    switch (variant) {
        case 4: return launch_split_ws<128, 128>(gg, st);
        case 0: return launch_split_t<128, 128, 2, 2>(gg, st);
        case 1: return launch_split_t<128, 64, 2, 2>(gg, st);
        case 2: return launch_split_t<64, 128, 2, 2>(gg, st);
        case 3: return launch_split_t<64, 64, 2, 2>(gg, st);
    }
    set_error("split_gemm: bad variant");
    return -1;
}

}  // namespace sva
