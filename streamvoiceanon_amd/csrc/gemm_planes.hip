// conv-GEMM on the 16-bit matrix pipes from PRE-SPLIT operand planes.
//
// gemm_split.hip splits every fp32 element of both operands into three bf16 parts while staging each tile, once per tile that
// touches it (a weight element M / 128 times, an activation element N / 128 times): 2.6 VALU instructions per MFMA, 49 % issue
// stalls (profiles/r03_split_gemm_pmc_*).  Here the split is done ONCE per element, by whoever produces it:
//   * weights at engine finalize (make_weight_planes below), stored as NPL 16-bit planes [NPL][N][K], scaled by a power of two so
//     that the low parts of ordinary weights stay normal numbers (undone on the accumulator, exactly);
//   * activations by the epilogue of the producing GEMM (ConvGemm::Cp: the Linear -> GELU -> Linear pair of a ConvNeXt block,
//     firefly.py:421-440, and w1|w3 -> w2 of the window transformer, windowed_transformer.py:134-143, hand their hidden tensor
//     over as planes and never store it as fp32) or by to_planes_kernel after a non-GEMM producer; an A operand that only exists
//     as fp32 is split while it is staged, as before (APL = false).
// Two precisions (ConvGemm::pmode), both accumulating in fp32 on v_mfma_f32_16x16x32_f16 (round 4's third format -- S6, three bf16
// planes / six products -- was measured slower than the in-loop split of gemm_split.hip, 6 bytes per element against 4, and removed in
// round 5: profiles/r04_mm_mode_ab.txt, git history; the range-safe fp32-grade path is sva_config.mm_mode = 0 = gemm_split.hip):
//   H3: x = hi + lo in fp16 (2 x 11 bits), three products hi.hi + hi.lo + lo.hi; the dropped lo.lo is <= 2^-24 of the leading one --
//       fp32-grade with HALF the matrix work of the six-product bf16 split, for operands inside the fp16 range (the reference runs this
//       path under torch.autocast(fp16), evaluations/infer_arvc.py:493, so its own activations are).  Error of the split itself:
//       |x - hi - lo| <= 2^-23 |x| while lo is a NORMAL fp16 number, i.e. for |x| >= 2^-3; below that lo is subnormal (or the residual
//       falls under fp16's smallest subnormal 2^-24) and the error is ABSOLUTE, <= 2^-25: 3e-8 per element, i.e. 3e-5 relative at
//       |x| = 1e-3.  Weights are pre-scaled into [2^7, 2^8) so theirs are relative; activation planes are NOT scaled (LayerNorm / GELU /
//       SwiGLU outputs of O(1) magnitude): an output's error from its small activations is bounded by 2^-25 * sum_k |w_k|, far below
//       the 2^-24-relative rounding of the O(1) terms of the same sum (tests: test_planes_gemm_small_activations_absolute_floor);
//   H1: x = fp16(x), one product -- the reference's own precision (autocast), a sixth of the matrix work.
//
// LDS image (both operands, every path): a K tile is cut into 1 KiB PIECES = 16 rows x 32 k of one plane, stored chunk-major --
// lane l = (k-chunk l >> 4, row l & 15) of the piece owns bytes [16 l, 16 l + 16).  That is the order the MFMA fragment wants
// (lane (fr, fk) supplies k = 8 fk .. 8 fk + 7 of row fr), so a fragment read is ONE lane-linear ds_read_b128 per piece, free of
// bank conflicts in all four lane groups of the instruction (MI355X_MICROARCH.md, LDS table), and staging a piece is one 16-byte
// load + one ds_write_b128 per lane.  Double-buffered, one barrier per K tile; tile k + 2 is requested from memory while tile k
// multiplies and tile k + 1 is being written.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "planes_split.h"
#include "sva_common.h"

namespace sva {
bool planes_gemm_supported(const ConvGemm& g);
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int MODE> struct PM;
template <> struct PM<PLANES_H3> { static constexpr int NPL = 2; };
template <> struct PM<PLANES_H1> { static constexpr int NPL = 1; };

// two fp32 -> one packed pair per plane
template <int MODE>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&o)[PM<MODE>::NPL]) {
    const f32x2 v = {a, b};
    f16x2 h = __builtin_convertvector(v, f16x2);
    o[0] = __builtin_bit_cast(unsigned, h);
    if constexpr (MODE == PLANES_H3) {
        // (an opaque copy of hi: the residual must be taken from the ROUNDED value, whatever the optimiser makes of the casts -- gemm_f16w.hip: split8)
        asm("" : "+v"(o[0]));
        const f16x2 hq = __builtin_bit_cast(f16x2, o[0]);
        const f32x2 r = {a - (float)hq.x, b - (float)hq.y};
        o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
}
// eight consecutive k of one row -> one 16-byte chunk per plane
template <int MODE>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, u32x4 (&o)[PM<MODE>::NPL]) {
    constexpr int NPL = PM<MODE>::NPL;
    unsigned p0[NPL], p1[NPL], p2[NPL], p3[NPL];
    split_pair<MODE>(v0.x, v0.y, p0);
    split_pair<MODE>(v0.z, v0.w, p1);
    split_pair<MODE>(v1.x, v1.y, p2);
    split_pair<MODE>(v1.z, v1.w, p3);
#pragma unroll
    for (int p = 0; p < NPL; ++p) o[p] = (u32x4){p0[p], p1[p], p2[p], p3[p]};
}

template <int MODE>
__device__ __forceinline__ f32x4 mma1(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// BM x BN tile, WM x WN waves of 64 x 64 (or smaller) wave tiles, BK = 32 KB k per tile.  APL: the A operand comes as planes (g.Ap),
// otherwise as fp32 (g.A) and is split while it is staged
template <int MODE, int BM, int BN, int KB, bool APL, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 4 && BM > 128) ? 1 : 2) void planes_gemm_kernel(const ConvGemmGroup gg) {
    constexpr int NW = WM * WN, NTH = 64 * NW, BK = 32 * KB, NPL = PM<MODE>::NPL;
    constexpr int NS = KB == 1 ? 3 : 2;                         // register stages of global loads in flight
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int RBA = BM / 16, RBB = BN / 16;                 // row blocks (pieces per plane and k block) of the A / B tile
    constexpr int A_BYTES = KB * NPL * RBA * 1024, B_BYTES = KB * NPL * RBB * 1024, STAGE = A_BYTES + B_BYTES;
    constexpr int IA = RBA / NW, IB = RBB / NW;                 // row blocks per wave
    static_assert(RBA % NW == 0 && RBB % NW == 0 && IA >= 1 && IB >= 1, "tile rows: whole row blocks per wave");
    const ConvGemm& g = gg.g[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tbx = blockIdx.x, tby = blockIdx.y;
    xcd_tile(gg.xcd_swz, gridDim.x, gridDim.y, tbx, tby);
    const int bm0 = tby * BM, bn0 = tbx * BN;
    const int prow = lane & 15, pchunk = lane >> 4;            // this lane's (row, k-chunk) inside a piece

    // ---- operand pointers of the pieces this thread stages: row blocks wave, wave + 4, ... ----
    const float* a_f32[IA];
    const unsigned short* a_pl[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        int m = bm0 + (wave + NW * i) * 16 + prow;
        if (m > g.M - 1) m = g.M - 1;
        const int b = m / g.T, t = m - b * g.T;
        const long off = (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + pchunk * 8;
        a_f32[i] = g.A + off;
        // planes are K-blocked over dense rows (planes_split.h): this lane's 16 bytes of k block 0
        a_pl[i] = g.Ap + ((long)b * (g.a_bstride / g.lda) + g.a_off / g.lda + t) * 32 + pchunk * 8;
    }
    const unsigned short* b_pl[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        int n = bn0 + (wave + NW * i) * 16 + prow;
        if (n > g.N - 1) n = g.N - 1;
        b_pl[i] = g.Wp + (long)n * 32 + pchunk * 8;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kc_tiles = g.Cin / BK;
    const int nk = g.taps * kc_tiles;
    const int dbg = gg.xcd_swz >> 8;            // TIMING EXPERIMENT (results are garbage): 1 no global loads in the K loop, 2 no LDS stores, 4 no MFMAs, 8 no epilogue
    // (kernel-argument fields the K loop needs, as values)
    const long g_tapstep = (long)g.dil * g.lda, g_ap_ps = g.ap_pstride, g_wp_ps = g.wp_pstride;
    const long g_ablk = g.ap_rows * 32, g_wblk = (long)g.N * 32;          // elements between consecutive 32-k blocks of a plane
    const int g_cin = g.Cin, g_silu = g.a_silu;

    // NS register stages: the loads of tiles k + 2 .. k + NS are in flight while tile k multiplies and tile k + 1 is written to LDS
    // (one stage ahead -- the first version -- left a tile's loads ~0.3 us, one tile's worth of MFMAs, to cover a 1-2 us trip to
    // L2 / HBM: every variant and precision ran at the same ~5.5 TB/s of operand traffic, 12-26 % of its matrix pipe)
    constexpr int AV = APL ? NPL : 2;                           // 16-byte loads per staged A piece
    constexpr int LPS = KB * (IA * AV + IB * NPL);              // loads per thread and K tile
    struct Regs {
        u32x4 a[KB][IA][AV];
        u32x4 b[KB][IB][NPL];
    };
    Regs rg[NS];
    auto gload = [&](int kt, Regs& r) {
        const int tap = kt / kc_tiles;
        const int kc = (kt - tap * kc_tiles) * BK;
        const long aoff = (long)tap * g_tapstep + kc;
        const long bblk = ((long)tap * g_cin + kc) >> 5;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                if constexpr (APL) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) r.a[kb][i][p] = *reinterpret_cast<const u32x4*>(a_pl[i] + (long)p * g_ap_ps + ((kc >> 5) + kb) * g_ablk);
                } else {
                    r.a[kb][i][0] = *reinterpret_cast<const u32x4*>(a_f32[i] + aoff + kb * 32);
                    r.a[kb][i][1] = *reinterpret_cast<const u32x4*>(a_f32[i] + aoff + kb * 32 + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int p = 0; p < NPL; ++p) r.b[kb][i][p] = *reinterpret_cast<const u32x4*>(b_pl[i] + (long)p * g_wp_ps + (bblk + kb) * g_wblk);
        }
    };
    auto lstore = [&](int buf, const Regs& r) {
        char* const st = lds + buf * STAGE + lane * 16;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                u32x4 o[NPL];
                if constexpr (APL) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) o[p] = r.a[kb][i][p];
                } else {
                    f32x4 v0 = __builtin_bit_cast(f32x4, r.a[kb][i][0]), v1 = __builtin_bit_cast(f32x4, r.a[kb][i][1]);
                    if (g_silu) {
                        v0.x = silu_f(v0.x); v0.y = silu_f(v0.y); v0.z = silu_f(v0.z); v0.w = silu_f(v0.w);
                        v1.x = silu_f(v1.x); v1.y = silu_f(v1.y); v1.z = silu_f(v1.z); v1.w = silu_f(v1.w);
                    }
                    split8<MODE>(v0, v1, o);
                }
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(st + ((kb * NPL + p) * RBA + wave + NW * i) * 1024) = o[p];
            }
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(st + A_BYTES + ((kb * NPL + p) * RBB + wave + NW * i) * 1024) = r.b[kb][i][p];
        }
    };
    // fragments of one 32-wide k block of the staged tile, and its products for the column sub-tiles [j0, j1)
    struct Frags { u32x4 a[MI][NPL], b[NI][NPL]; };
    auto load_frags = [&](int buf, int kb, Frags& f) {
        const char* const sa = lds + buf * STAGE + lane * 16 + (wm * MI) * 1024;
        const char* const sb = lds + buf * STAGE + A_BYTES + lane * 16 + (wn * NI) * 1024;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) f.a[i][p] = *reinterpret_cast<const u32x4*>(sa + ((kb * NPL + p) * RBA + i) * 1024);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) f.b[j][p] = *reinterpret_cast<const u32x4*>(sb + ((kb * NPL + p) * RBB + j) * 1024);
    };
    auto mma = [&](const Frags& f, int j0, int j1) {
        // small products first
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            if constexpr (MODE != PLANES_H1) {
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mma1<MODE>(f.a[i][1], f.b[j][0], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mma1<MODE>(f.a[i][0], f.b[j][1], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = mma1<MODE>(f.a[i][0], f.b[j][0], acc[i][j]);
        }
    };

    // tile 0 -> LDS stage 0; tiles 1 .. NS in flight (tile s in register stage s % NS).  Every iteration stores and requests
    // unconditionally (beyond the last tile: a clamped re-read of it, stored into the stage nobody reads any more), so the queue
    // always holds NS stages in issue order when a stage is waited for
    gload(0, rg[0]);
    lstore(0, rg[0]);
#pragma unroll
    for (int s_ = 1; s_ <= NS; ++s_) gload(s_ < nk ? s_ : nk - 1, rg[s_ % NS]);
    for (int kt = 0; kt < nk; kt += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int k = kt + u;
            if (k >= nk) goto k_done;
            __syncthreads();                    // tile k is in LDS stage k & 1; nobody still reads the other stage
            // this tile's fragment reads go to the LDS queue FIRST and half of its products are issued before the wave stops at
            // the wait for tile k + 1's global loads and stores that tile: the stores drain behind the reads while the matrix pipe
            // works (stores first -- the first version -- put them, and the wait in front of them, on every tile's critical path)
            Frags f;
            load_frags(k & 1, 0, f);
            if (!(dbg & 4)) mma(f, 0, NI / 2);
            if (!(dbg & 2)) lstore((k + 1) & 1, rg[(u + 1) % NS]);
            if (!(dbg & 1)) gload(k + 1 + NS < nk ? k + 1 + NS : nk - 1, rg[(u + 1) % NS]);
            if (!(dbg & 4)) mma(f, NI / 2, NI);
#pragma unroll
            for (int kb = 1; kb < KB; ++kb) {
                load_frags(k & 1, kb, f);
                mma(f, 0, NI);
            }
        }
    }
k_done:
    __syncthreads();

    if (dbg & 8) { if (acc[0][0][0] == 123.456f) g.C[0] = 1.f; return; }
    // ---- epilogue (as conv_gemm_kernel / split_gemm_kernel: accumulators staged through LDS for whole 16-byte row accesses) ----
    constexpr int CS = BN + 4;
    float* Cs = smem;                              // [BM][CS]
    const int col = lane & 15, rq = (lane >> 4) * 4;
    const float winv = g.wp_inv;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wm * TM + i * 16 + rq + r) * CS + wn * TN + j * 16 + col] = acc[i][j][r] * winv;
    __syncthreads();
    const long c_rows_b = g.ldc ? g.c_bstride / g.ldc : 0, c_row0 = g.ldc ? g.c_off / g.ldc : 0;
    auto store4 = [&](const float4& v, long idx, int b_, int t_, int n_) {
        {
            // fp16 parts have fp16's range: an operand beyond +-65504 turns into inf and the output into inf / nan.  The reference
            // (torch.autocast(fp16)) has the same limit; here it is REPORTED (sva_sync fails, naming mm_mode = 0) instead of propagating
            if (g.ovf && !(fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w) < INFINITY)) *reinterpret_cast<volatile int*>(g.ovf) = 1;
        }
        if (g.C) *reinterpret_cast<float4*>(g.C + idx) = v;
        if (g.Cp) {
            unsigned p0[NPL], p1[NPL];
            split_pair<MODE>(v.x, v.y, p0);
            split_pair<MODE>(v.z, v.w, p1);
            const long po = plane_off_blocked((long)b_ * c_rows_b + c_row0 + t_, n_, g.cp_rows);
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x2*>(g.Cp + (long)p * g.cp_pstride + po) = (u32x2){p0[p], p1[p]};
        }
    };
    if (g.w13) {
        // SwiGLU: tile columns alternate 16 x w1 | 16 x w3; output column (n0 >> 1) + c
        constexpr int OC4 = BN / 8;                // float4 chunks of output per row
        const int eb0w = bm0 / g.T, et0w = bm0 - eb0w * g.T;
        for (int idx = tid; idx < BM * OC4; idx += NTH) {
            const int row = idx / OC4, q = idx - row * OC4;
            const int m = bm0 + row;
            const int grp = q >> 2, c4 = (q & 3) * 4;       // 16-wide group, offset inside it
            const int n = bn0 + grp * 32 + c4;              // w1 column
            if (m >= g.M || n >= g.N) continue;
            int b = eb0w, t = et0w + row;
            if (t >= g.T) { t -= g.T; ++b; if (t >= g.T) { b += t / g.T; t %= g.T; } }
            if (t >= g.skip_lo && t < g.skip_hi) continue;
            const float4 a = *reinterpret_cast<const float4*>(&Cs[row * CS + grp * 32 + c4]);
            const float4 w = *reinterpret_cast<const float4*>(&Cs[row * CS + grp * 32 + 16 + c4]);
            float4 o;
            o.x = silu_f(a.x) * w.x; o.y = silu_f(a.y) * w.y; o.z = silu_f(a.z) * w.z; o.w = silu_f(a.w) * w.w;
            store4(o, (long)b * g.c_bstride + g.c_off + (long)t * g.ldc + ((bn0 + grp * 32) >> 1) + c4, b, t, ((bn0 + grp * 32) >> 1) + c4);
        }
        return;
    }
    constexpr int C4 = BN / 4;
    const int eb0 = bm0 / g.T, et0 = bm0 - eb0 * g.T;          // (one division per workgroup; a tile rarely spans more than two batch items)
    for (int idx = tid; idx < BM * C4; idx += NTH) {
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        const int m = bm0 + row, n = bn0 + c4;
        if (m >= g.M || n >= g.N) continue;
        int b = eb0, t = et0 + row;
        if (t >= g.T) { t -= g.T; ++b; if (t >= g.T) { b += t / g.T; t %= g.T; } }
        if (t >= g.skip_lo && t < g.skip_hi) continue;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * CS + c4]);
        if (g.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (g.act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        else if (g.act == ACT_LOGCLAMP) { v.x = __logf(fmaxf(v.x, 1e-5f)); v.y = __logf(fmaxf(v.y, 1e-5f)); v.z = __logf(fmaxf(v.z, 1e-5f)); v.w = __logf(fmaxf(v.w, 1e-5f)); }
        if (g.gamma) {
            const float4 gm = *reinterpret_cast<const float4*>(g.gamma + n);
            v.x *= gm.x; v.y *= gm.y; v.z *= gm.z; v.w *= gm.w;
        }
        if (g.res) {
            const float4 rr = *reinterpret_cast<const float4*>(g.res + (long)b * g.r_bstride + g.r_off + (long)t * g.ldr + n);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        v.x *= g.scale; v.y *= g.scale; v.z *= g.scale; v.w *= g.scale;
        const long ci = (long)b * g.c_bstride + g.c_off + (long)t * g.ldc + n;
        if (g.accumulate) {
            const float4 cc = *reinterpret_cast<const float4*>(g.C + ci);
            v.x += cc.x; v.y += cc.y; v.z += cc.z; v.w += cc.w;
        }
        store4(v, ci, b, t, n);
    }
}

template <int MODE, int BM, int BN, int KB, bool APL, int WM = 2, int WN = 2>
int launch_planes_t(const ConvGemmGroup& gg_in, hipStream_t st) {
    ConvGemmGroup gg = gg_in;
    const ConvGemm& g = gg.g[0];
    constexpr size_t smem_ab = (size_t)2 * KB * PM<MODE>::NPL * (BM + BN) / 16 * 1024;
    constexpr size_t smem_c = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t smem = smem_ab > smem_c ? smem_ab : smem_c;
    static_assert(smem <= 160 * 1024, "LDS of one CU");
    static DeviceOnce attr_set;
    if (attr_set.needed() && smem > 48 * 1024) {
        SVA_HIP(hipFuncSetAttribute((const void*)planes_gemm_kernel<MODE, BM, BN, KB, APL, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set.done();
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, gg.n);
    gg.xcd_swz = xcd_swizzle_for(grid.x, grid.y) | (debug_options().planes_dbg << 8);
    hipLaunchKernelGGL((planes_gemm_kernel<MODE, BM, BN, KB, APL, WM, WN>), grid, dim3(64 * WM * WN), smem, st, gg);
    return 0;
}

template <int MODE, bool APL>
int launch_planes_m(const ConvGemmGroup& gg, int variant, hipStream_t st) {
    const bool k64 = gg.g[0].Cin % 64 == 0;
    switch (variant) {
        case 0: return launch_planes_t<MODE, 128, 128, 1, APL>(gg, st);
        case 1: return launch_planes_t<MODE, 128, 64, 1, APL>(gg, st);
        case 2: return launch_planes_t<MODE, 64, 128, 1, APL>(gg, st);
        case 3: return launch_planes_t<MODE, 64, 64, 1, APL>(gg, st);
        // 64-deep K tiles (half the barriers; the tile's stage is twice as large)
        case 4: if (k64) return launch_planes_t<MODE, 128, 128, 2, APL>(gg, st); return launch_planes_t<MODE, 128, 128, 1, APL>(gg, st);
        case 5: if (k64) return launch_planes_t<MODE, 64, 64, 2, APL>(gg, st); return launch_planes_t<MODE, 64, 64, 1, APL>(gg, st);
        // 8 waves: 256 x 128 (each weight tile feeds twice the rows: two thirds of the operand traffic per flop of 128 x 128)
        case 6: return launch_planes_t<MODE, 256, 128, 1, APL, 4, 2>(gg, st);
        // 128 x 128 on 8 waves of 32 x 64: four waves per SIMD with two workgroups per CU (more independent instruction streams to
        // overlap a tile's load / LDS-store / fragment-read / MFMA phases, for 1.5 x the fragment reads)
        case 7: return launch_planes_t<MODE, 128, 128, 1, APL, 4, 2>(gg, st);
    }
    set_error("planes_gemm: bad variant");
    return -1;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// planes_dma_kernel (round 5): the same GEMM as ONE CONTINUOUS, LDS-DMA-FED STREAM OF K STEPS per workgroup.
//
// Round 4's leave-one-out probe (profiles/r04_planes_probe.txt) showed planes_gemm_kernel to be the SUM of its phases: global loads
// into registers, ds_write_b128 of the same bytes, fragment reads, MFMAs and an epilogue through LDS each took their turn -- hipcc
// drains the load queue (vmcnt(0)) at the loop head whatever ring of register stages the source keeps, and with K = 384-2048 a tile's
// fill / drain / epilogue are ~40 % of it.  This kernel changes the structure instead of the schedule:
//   * both operands are planes (the producers split: weights at finalize, activations in the epilogue / output pass of whoever wrote
//     them), so global -> LDS is a pure copy: `global_load_lds_dwordx4` (16 bytes per lane, one 1 KiB piece = 16 rows x 32 k of one
//     plane per wave-instruction; lane quads take a row's four chunks so the wave reads its KiB front to back, the LDS image is row-major
//     with the chunk XOR-swizzled per row group -- see the lane-order comment in the kernel; the rows are picked by the per-lane SOURCE
//     offsets).  No staging registers, no ds_write, no VALU in the K loop;
//   * a ring of NST LDS stages, requests NST - 1 K steps ahead, COUNTED s_waitcnt vmcnt(N) (never 0 in steady state) + one raw
//     s_barrier per K step: the DMA stays in flight across barriers (cdna_hip_programming.md, "Pipelining across barriers").  The
//     loads are inline asm, invisible to hipcc's wait-count pass, and the K loop holds no other VMEM instruction;
//   * PERSISTENT: a workgroup walks its tiles as one stream of (tile, k) steps -- the first K steps of the next tile are already in
//     flight while the current tile's epilogue runs;
//   * the epilogue never touches LDS (the ring keeps flowing): the products are taken TRANSPOSED -- weights as the MFMA's row
//     operand -- so a lane holds FOUR CONSECUTIVE OUTPUT COLUMNS of one row (D[n][m]: n = 4 (lane >> 4) + r, m = lane & 15): bias /
//     GELU / gamma / residual / SwiGLU (its w1 | w3 column groups are two accumulators of the same lane) and the split into output
//     planes happen in registers, stores are 16 bytes (fp32) or 8 bytes per plane per lane.
// Tile BM x 128 on 8 multiplying waves (two per SIMD): BM = 256 -> 4 x 2 waves of 64 x 64, three 48 KiB stages; BM = 128 -> 2 x 4 waves of
// 64 x 32, four (or two: then two workgroups per CU) 32 KiB stages.  LW = 2 adds two LOADER waves that issue every request (see LW in the
// kernel); CONV adds conv taps over the A planes, groups of three problems and narrow tiles (the HiFiGAN ResBlock convs).  The grid is
// min(tiles, CUs of the stream x workgroups per CU); XCD x owns a contiguous band of the (n-fastest) tile sequence, so the row panel a band
// shares is fetched into one L2.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char* uni_ptr(const char* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned long long r = ((unsigned long long)uni((unsigned)(u >> 32)) << 32) | uni((unsigned)u);
    return reinterpret_cast<const char*>(r);
}
// one 1 KiB piece: lane l copies 16 bytes from base + voff(l) to LDS byte address lds_dst + 16 l (M0 = the wave-uniform destination;
// saved and restored around the instruction: the register is the compiler's)
__device__ __forceinline__ void glds16(const char* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// resident workgroups per CU of a tile configuration: eight-wave workgroups two when the LDS allows (one's epilogue runs under the other's K
// steps); the narrow-tile configurations of the HiFiGAN levels (four / two waves) as many as LDS and 16 waves per CU allow
template <int MODE, int BM, int BN, int NST, int NW> struct DmaCfg {
    static constexpr size_t smem = (size_t)NST * PM<MODE>::NPL * (BM + BN) / 16 * 1024;
    static constexpr int by_lds = (int)((160 * 1024) / smem), by_waves = 16 / NW;
    static constexpr int WG_PER_CU = NW == 8 ? (by_lds >= 2 ? 2 : 1) : (by_lds < by_waves ? by_lds : by_waves);
    static constexpr int WAVES_PER_SIMD = NW == 8 ? 2 : (WG_PER_CU * NW + 3) / 4;
};

template <int MODE, int BM, int NST, bool PROBE, int BN = 128, int NW = 8, bool CONV = false, int LW = 0>
__global__ __launch_bounds__(64 * (NW + LW), (LW > 0 ? 3 : DmaCfg<MODE, BM, BN, NST, NW>::WAVES_PER_SIMD)) void planes_dma_kernel(const ConvGemmGroup gg, const int n_tiles_n, const int n_tiles, const int dbg_arg) {
    // CONV: conv taps and / or a group of problems (the generic tile deal and per-tile problem lookup); !CONV: one taps == 1 problem, every
    // per-problem quantity a launch constant (the encoder's GEMMs: the K loop carries no trace of the generality)
    // gg.n problems of ONE shape (M, N, Cin, T) that may differ in taps / dilation / pointers (the three ResBlock branches of a HiFiGAN level, k = 3 / 7 / 11):
    // tile ids [p * n_tiles, (p + 1) * n_tiles) belong to problem p; a workgroup's stream of K steps simply runs on across problems
    const int dbg = PROBE ? dbg_arg : 0;          // (the product instantiation carries none of the switches below)
    // dbg: TIMING EXPERIMENTS (SVA_DEBUG planes_dbg; results are garbage): 1 no LDS-DMA requests, 4 no MFMAs, 8 no epilogue, 32 no fragment reads
    constexpr int NPL = PM<MODE>::NPL;
    constexpr int PD = NST - 1;                                           // NST LDS stages; K steps requested ahead
    constexpr int WM = BM / 64, WN = NW / WM, TN = BN / WN, MI = 4, NI = TN / 16;
    // LW > 0: LOADER WAVES.  Waves NW .. NW + LW - 1 issue every LDS-DMA request of the workgroup, waves 0 .. NW - 1 only read fragments, multiply and
    // run the epilogue.  With every wave doing both, the 8 waves' 32 requests of a step queue at the CU's address unit for ~680 cycles right behind the
    // barrier and no wave reaches its MFMAs before its own are accepted; then the matrix pipe runs ~770 cycles during which nobody requests anything:
    // request phase and multiply phase alternate (tools/probes/fill_probe.hip: the request pattern alone 0.41 us per step, with the reads and MFMAs
    // of the step 0.78, with two loader waves 0.55)
    constexpr int NL = LW > 0 ? LW : NW;                                  // loading waves
    constexpr int RBA = BM / 16, RBB = BN / 16, PA = RBA / NL, PB = RBB / NL;
    constexpr int A_BYTES = NPL * RBA * 1024, STAGE = A_BYTES + NPL * RBB * 1024;
    constexpr int LPW = (PA + PB) * NPL;                                  // LDS-DMA instructions per loading wave and K step
    static_assert(PD * LPW <= 48 && RBA % NL == 0 && RBB % NL == 0, "vmcnt is a 6-bit counter; whole pieces per loading wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds0 = uni((unsigned)(size_t)lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)uni((unsigned)(tid >> 6));
    const int wm = wave / WN, wn = wave % WN;
    const int lwave = LW > 0 ? wave - NW : wave;          // index among the loading waves (negative: a consumer wave of the LW > 0 form)
    // Lane order inside a 1 KiB piece (16 rows x 64 bytes, contiguous in the K-blocked planes): lane QUADS take a row's four 16-byte chunks, so the
    // wave reads the KiB front to back -- with (row = lane & 15, chunk = lane >> 4), the MFMA operand order, every quad touches four rows and the
    // fill runs at 76 instead of 105 GB/s per CU (tools/probes/fill_probe.hip).  The LDS image is therefore ROW-MAJOR; the chunk a quad lane takes is
    // XOR-swizzled per row group (f = 0, 2, 3, 1 for rows 0-3, 4-7, 8-11, 12-15) so that the fragment reads (lane -> row lane & 15, chunk lane >> 4) of
    // every ds_read_b128 lane group fall on sixteen different 16-byte slots (SQ_LDS_BANK_CONFLICT = 0)
    const int prow = lane >> 2, pchunk = (lane & 3) ^ ((0x78 >> (2 * (lane >> 4))) & 3);
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((0x78 >> (2 * ((lane & 15) >> 2))) & 3)) * 16);

    // ---- this workgroup's tiles (XCD x = workgroup id & 7, a speed assumption only).  One problem: XCD x owns the x-th contiguous eighth of the
    // tile sequence (a band of M tiles with all their N tiles: the A panel is fetched into one L2).  A group (members sorted by the host, longest K
    // first): XCD x owns the M tiles with tm % 8 == x of every member, and its workgroups deal them out boustrophedon (slot s: s, 2 S - 1 - s,
    // 2 S + s, ...), so that the workgroup that drew a long-K tile first draws a short one next -- the members' K differ 11 : 7 : 3 ----
    const int G = gridDim.x, wg = blockIdx.x;
    const int all_tiles = n_tiles * gg.n;
    const int xcd = wg & 7, slot = wg >> 3, nslots = (G - xcd + 7) >> 3;
    const bool grouped = CONV && gg.n > 1;
    const int n_tiles_m = n_tiles / n_tiles_n;
    const int q = all_tiles >> 3, r8 = all_tiles & 7;
    const int band_lo = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int per_member = ((n_tiles_m - xcd + 7) >> 3) * n_tiles_n;          // grouped: this XCD's tiles of one member
    const int band_n = grouped ? per_member * gg.n : q + (xcd < r8 ? 1 : 0);
    auto seq_of = [&](int it) { return (grouped && (it & 1)) ? it * nslots + (nslots - 1 - slot) : it * nslots + slot; };
    int my_tiles = slot < band_n ? (band_n - slot + nslots - 1) / nslots : 0;
    if (grouped) { my_tiles = 0; while (seq_of(my_tiles) < band_n) ++my_tiles; }
    auto tau_of = [&](int it) -> int {       // tile id = member * n_tiles + tm * n_tiles_n + tn
        const int j = seq_of(it);
        if (!grouped) return band_lo + j;
        const int pi = j / per_member, r = j - pi * per_member;
        const int tmx = r / n_tiles_n;
        return pi * n_tiles + (xcd + 8 * tmx) * n_tiles_n + (r - tmx * n_tiles_n);
    };
    const int kcb = gg.g[0].Cin / 32;                                     // 32-wide K blocks per tap
    // (!CONV: the row geometry of the one problem, divided once)
    const long a_rows_b0 = gg.g[0].a_bstride / gg.g[0].lda, a_row00 = gg.g[0].a_off / gg.g[0].lda;
    const long c_rows_b0 = gg.g[0].ldc ? gg.g[0].c_bstride / gg.g[0].ldc : 0, c_row00 = gg.g[0].ldc ? gg.g[0].c_off / gg.g[0].ldc : 0;
    auto nk_of = [&](int it) { return CONV ? gg.g[tau_of(it) / n_tiles].taps * kcb : kcb; };
    int total = my_tiles * kcb;
    if (CONV) { total = 0; for (int it = 0; it < my_tiles; ++it) total += nk_of(it); }
    if (total == 0) return;

    // ---- load side: (tile, k) of the next step to request, per-lane source offsets relative to the tile's wave-uniform bases ----
    int ld_it = 0, ld_k = 0, ld_kb = 0, ld_tap = 0, ld_nk = 0, ld_stage = 0;
    unsigned offA[PA], offW[PB];
    const char *baseA = nullptr, *baseW = nullptr;
    long a_blk2 = 0, a_ps2 = 0, w_ps2 = 0, tap2 = 0;          // bytes between 32-k blocks of an A plane / between planes / per tap (dil rows of 64 bytes)
    const long w_blk2 = (long)gg.g[0].N * 64;
    auto ld_tile = [&]() {
        const int tau = tau_of(ld_it);
        const int pi = CONV ? tau / n_tiles : 0, tile = tau - pi * n_tiles;
        const ConvGemm& gl = gg.g[pi];
        const int tm = tile / n_tiles_n, tn = tile - tm * n_tiles_n;
        const int bm0 = tm * BM, bn0 = tn * BN;
        // K-blocked planes over dense rows (planes_split.h): dense row of (b, t) = b * (a_bstride / lda) + a_off / lda + t, + tap * dil for a conv tap
        const long a_rows_b = CONV ? gl.a_bstride / gl.lda : a_rows_b0, a_row0 = CONV ? gl.a_off / gl.lda : a_row00;
        auto row_of = [&](int m) -> long {
            if (m > gl.M - 1) m = gl.M - 1;
            const int b = m / gl.T, t = m - b * gl.T;
            return (long)b * a_rows_b + a_row0 + t;
        };
        const long r0 = row_of(bm0);
        baseA = uni_ptr(reinterpret_cast<const char*>(gl.Ap + r0 * 32));
        baseW = uni_ptr(reinterpret_cast<const char*>(gl.Wp + (long)bn0 * 32));
#pragma unroll
        for (int i = 0; i < PA; ++i) offA[i] = (unsigned)(((row_of(bm0 + (lwave + NL * i) * 16 + prow) - r0) * 32 + pchunk * 8) * 2);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int n = bn0 + (lwave + NL * i) * 16 + prow;
            if (n > gl.N - 1) n = gl.N - 1;
            offW[i] = (unsigned)(((long)(n - bn0) * 32 + pchunk * 8) * 2);
        }
        a_blk2 = gl.ap_rows * 64; a_ps2 = gl.ap_pstride * 2; w_ps2 = gl.wp_pstride * 2; tap2 = (long)gl.dil * 64;
        ld_nk = CONV ? gl.taps * kcb : kcb;
    };
    auto issue = [&]() {
        if (ld_k == 0) ld_tile();
        const char* const ba = CONV ? uni_ptr(baseA + (long)ld_kb * a_blk2 + (long)ld_tap * tap2) : baseA + (long)ld_k * a_blk2;
        const char* const bw = CONV ? uni_ptr(baseW + (long)ld_k * w_blk2) : baseW + (long)ld_k * w_blk2;             // (W is K-blocked over taps * Cin: block = tap * kcb + kb = the step)
        const unsigned dst = lds0 + (unsigned)ld_stage * STAGE;
        if (!(dbg & 1)) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int i = 0; i < PA; ++i) glds16(ba + p * a_ps2, offA[i], dst + (unsigned)((p * RBA + lwave + NL * i) * 1024));
#pragma unroll
                for (int i = 0; i < PB; ++i) glds16(bw + p * w_ps2, offW[i], dst + (unsigned)(A_BYTES + (p * RBB + lwave + NL * i) * 1024));
            }
        }
        if (CONV) { if (++ld_kb == kcb) { ld_kb = 0; ++ld_tap; } }
        if (++ld_k == ld_nk) { ld_k = 0; ld_tap = 0; ++ld_it; }
        if (++ld_stage == NST) ld_stage = 0;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- epilogue of the tile the compute side has finished: registers -> global, no LDS ----
    auto epilogue = [&](int tau) {
        const int pi = CONV ? tau / n_tiles : 0, tile = tau - pi * n_tiles;
        const ConvGemm& g = gg.g[pi];
        const float winv = g.wp_inv;
        const long c_rows_b = CONV ? (g.ldc ? g.c_bstride / g.ldc : 0) : c_rows_b0, c_row0 = CONV ? (g.ldc ? g.c_off / g.ldc : 0) : c_row00;
        auto store4 = [&](const f32x4& v, long idx, int b_, int t_, int n_) {
            // fp16 parts have fp16's range: an operand beyond +-65504 is REPORTED (host-mapped flag; sva_sync / the next step fails naming mm_mode = 0)
            if (g.ovf && !(fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w) < INFINITY)) *reinterpret_cast<volatile int*>(g.ovf) = 1;
            if (g.C) *reinterpret_cast<f32x4*>(g.C + idx) = v;
            if (g.Cp) {
                f32x4 w = v;
                if (CONV && g.cp_silu) { w.x = silu_f(v.x); w.y = silu_f(v.y); w.z = silu_f(v.z); w.w = silu_f(v.w); }      // the consumer conv reads silu(.) (HiFiGAN)
                unsigned p0[NPL], p1[NPL];
                split_pair<MODE>(w.x, w.y, p0);
                split_pair<MODE>(w.z, w.w, p1);
                const long po = plane_off_blocked((long)b_ * c_rows_b + c_row0 + t_, n_, g.cp_rows);
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x2*>(g.Cp + (long)p * g.cp_pstride + po) = (u32x2){p0[p], p1[p]};
            }
        };
        const int tm = tile / n_tiles_n, tn = tile - tm * n_tiles_n;
        const int bm0 = tm * BM, bn0 = tn * BN;
        const int nq = 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = bm0 + wm * 64 + i * 16 + (lane & 15);
            const int b = m / g.T, t = m - b * g.T;
            const bool row_ok = m < g.M && !(t >= g.skip_lo && t < g.skip_hi);
            if (g.w13) {
                // SwiGLU: the tile's columns alternate 16 x w1 | 16 x w3 -- accumulators j (w1) and j + 1 (w3) of this lane; output column (n0 >> 1) + c
#pragma unroll
                for (int j = 0; j + 1 < NI; j += 2) {
                    const int n = bn0 + wn * TN + j * 16;              // w1 column block
                    if (!row_ok || n >= g.N) continue;
                    const f32x4 a = acc[i][j] * winv, w = acc[i][j + 1] * winv;
                    f32x4 o;
                    o.x = silu_f(a.x) * w.x; o.y = silu_f(a.y) * w.y; o.z = silu_f(a.z) * w.z; o.w = silu_f(a.w) * w.w;
                    store4(o, (long)b * g.c_bstride + g.c_off + (long)t * g.ldc + (n >> 1) + nq, b, t, (n >> 1) + nq);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = bn0 + wn * TN + j * 16 + nq;
                if (!row_ok || n >= g.N) continue;
                f32x4 v = acc[i][j] * winv;
                if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (g.act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
                else if (g.act == ACT_LOGCLAMP) { v.x = __logf(fmaxf(v.x, 1e-5f)); v.y = __logf(fmaxf(v.y, 1e-5f)); v.z = __logf(fmaxf(v.z, 1e-5f)); v.w = __logf(fmaxf(v.w, 1e-5f)); }
                if (g.gamma) v *= *reinterpret_cast<const f32x4*>(g.gamma + n);
                if (g.res) v += *reinterpret_cast<const f32x4*>(g.res + (long)b * g.r_bstride + g.r_off + (long)t * g.ldr + n);
                v *= g.scale;
                const long ci = (long)b * g.c_bstride + g.c_off + (long)t * g.ldc + n;
                if (g.accumulate) v += *reinterpret_cast<const f32x4*>(g.C + ci);
                store4(v, ci, b, t, n);
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // ---- the stream: steps 0 .. PD - 1 requested up front; iteration s waits for step s (its own pieces), meets the other waves (all
    // pieces of step s landed, everybody is done reading the stage of step s - 1), requests step s + PD into that stage, multiplies ----
    auto wait_step = [&](int allow) {         // at most `allow` requested steps may still be in flight (LPW instructions each)
        if (allow <= 0) wait_vm<0>();
        else if (allow == 1) wait_vm<LPW>();
        else if (PD < 3 || allow == 2) wait_vm<2 * LPW>();
        else wait_vm<3 * LPW>();
    };
    int cs = 0, ck = 0, cit = 0, cnk = nk_of(0);
    if constexpr (LW > 0) {
        // one barrier per K step for everybody.  Loader at barrier s: its pieces of step s have landed; behind it it requests step s + PD into the stage
        // of step s - 1.  Consumer at barrier s: its MFMAs of step s - 1 are issued (their fragments long read); behind it it reads step s.
        if (wave >= NW) {
#pragma unroll
            for (int s_ = 0; s_ < PD; ++s_)
                if (s_ < total) issue();
            for (int s_ = 0; s_ < total; ++s_) {
                wait_step(min(PD - 1, total - 1 - s_));
                __builtin_amdgcn_s_barrier();
                if (s_ + PD < total) issue();
            }
            return;
        }
        for (int s_ = 0; s_ < total; ++s_) {
            __builtin_amdgcn_s_barrier();
            const char* const sa = lds + cs * STAGE + frag_off + (wm * MI) * 1024;
            const char* const sb = lds + cs * STAGE + A_BYTES + frag_off + (wn * NI) * 1024;
            u32x4 fa[MI][NPL], fb[NI][NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j][p] = *reinterpret_cast<const u32x4*>(sb + (p * RBB + j) * 1024);
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][p] = *reinterpret_cast<const u32x4*>(sa + (p * RBA + i) * 1024);
            }
            if constexpr (MODE == PLANES_H3) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][0], fa[i][1], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][1], fa[i][0], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][0], fa[i][0], acc[i][j]);
            if (++cs == NST) cs = 0;
            if (++ck == cnk) {
                ck = 0;
                epilogue(tau_of(cit));
                ++cit;
                if (cit < my_tiles) cnk = nk_of(cit);
            }
        }
        return;
    }
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_)
        if (s_ < total) issue();
    bool prewaited = false;
    for (int s_ = 0; s_ < total; ++s_) {
        if (!prewaited) wait_step(min(PD - 1, total - 1 - s_));
        prewaited = false;
        __builtin_amdgcn_s_barrier();
        if (s_ + PD < total) issue();
        {
            const char* const sa = lds + cs * STAGE + frag_off + (wm * MI) * 1024;
            const char* const sb = lds + cs * STAGE + A_BYTES + frag_off + (wn * NI) * 1024;
            u32x4 fa[MI][NPL], fb[NI][NPL];
            if (dbg & 32) {
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) fb[j][p] = (u32x4){(unsigned)s_, 1u, 2u, 3u};
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[i][p] = (u32x4){(unsigned)s_, 5u, 6u, 7u};
                }
            } else {
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) fb[j][p] = *reinterpret_cast<const u32x4*>(sb + (p * RBB + j) * 1024);
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[i][p] = *reinterpret_cast<const u32x4*>(sa + (p * RBA + i) * 1024);
                }
            }
            if (dbg & 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) { acc[i][j][0] += __uint_as_float(fa[i][0][0] ^ fb[j][NPL - 1][1]); acc[i][j][1] += __uint_as_float(fa[i][NPL - 1][2] ^ fb[j][0][3]); }
            } else {
            // D[n][m] = sum_k W[n][k] X[m][k]: weights first.  Small products first; 16 (8) independent accumulators between dependent ones
            if constexpr (MODE == PLANES_H3) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][0], fa[i][1], acc[i][j]);        // w_hi . x_lo
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][1], fa[i][0], acc[i][j]);        // w_lo . x_hi
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][0], fa[i][0], acc[i][j]);
            }
        }
        if (++cs == NST) cs = 0;
        if (++ck == cnk) {
            ck = 0;
            // the next step's pieces are waited for BEFORE the epilogue's stores join the queue (vmcnt counts them too: waiting behind them
            // would park the first K step of every tile for a store round trip); the barrier of the next iteration makes it workgroup-wide
            if (s_ + 1 < total) { wait_step(min(PD - 1, total - 2 - s_)); prewaited = true; }
            if (dbg & 8) { if (acc[0][0][0] == 123.456f) epilogue(0); }
            else
            epilogue(tau_of(cit));
            ++cit;
            if (cit < my_tiles) cnk = nk_of(cit);
        }
    }
}

static int g_dma_cu_limit = 0;          // CUs a planes-DMA launch may count on (0: the device's); the engine lowers it for CU-masked streams

template <int MODE, int BM, int NST, bool CONV, int BN = 128, int NW = 8, int LW = 0>
int launch_planes_dma_t(const ConvGemmGroup& gg, hipStream_t st) {
    const ConvGemm& g = gg.g[0];
    constexpr size_t smem = DmaCfg<MODE, BM, BN, NST, NW>::smem;
    constexpr int WG_PER_CU = LW > 0 ? 1 : DmaCfg<MODE, BM, BN, NST, NW>::WG_PER_CU;
    static_assert(smem <= 160 * 1024 && WG_PER_CU >= 1, "LDS of one CU");
    static DeviceOnce attr_set;
    if (attr_set.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)planes_dma_kernel<MODE, BM, NST, false, BN, NW, CONV, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (MODE == PLANES_H3) SVA_HIP(hipFuncSetAttribute((const void*)planes_dma_kernel<PLANES_H3, BM, NST, true, BN, NW, CONV, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set.done();
    }
    int cus = g.cu_limit > 0 ? g.cu_limit : g_dma_cu_limit;
    if (cus <= 0) {
        static int dev_cus = 0;
        if (dev_cus == 0) {
            int dev = 0;
            SVA_HIP(hipGetDevice(&dev));
            SVA_HIP(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev));
        }
        cus = dev_cus;
    }
    const int tn = g.N / BN, tm = (g.M + BM - 1) / BM, tiles = tn * tm;
    const int grid = std::min(tiles * gg.n, cus * WG_PER_CU);
    ConvGemmGroup sorted = gg;                      // (longest K first: the kernel's tile deal relies on it)
    std::stable_sort(sorted.g, sorted.g + sorted.n, [](const ConvGemm& a, const ConvGemm& b_) { return a.taps > b_.taps; });
    const int dbg = debug_options().planes_dbg;
    if (dbg && MODE == PLANES_H3) hipLaunchKernelGGL((planes_dma_kernel<PLANES_H3, BM, NST, true, BN, NW, CONV, LW>), dim3(grid), dim3(64 * (NW + LW)), smem, st, sorted, tn, tiles, dbg);
    else hipLaunchKernelGGL((planes_dma_kernel<MODE, BM, NST, false, BN, NW, CONV, LW>), dim3(grid), dim3(64 * (NW + LW)), smem, st, sorted, tn, tiles, 0);
    return 0;
}

// variants 9 .. 14 of launch_planes_gemm (9 / 10: one / two workgroups per CU; 11 / 12: loader waves, 128 x 128 / 256 x 128; 13 / 14: the conv form's narrow tiles)
bool planes_dma_supported(const ConvGemm& g, bool conv = false) {
    return planes_gemm_supported(g) && g.Ap && !g.a_silu && g.N % (conv ? 64 : 128) == 0 && (!conv || !g.w13) && g.ksplit <= 1 && (!g.w13 || g.N % 32 == 0) &&
           (!g.C || (g.ldc % 4 == 0 && g.c_off % 4 == 0 && g.c_bstride % 4 == 0)) && (!g.res || (g.ldr % 4 == 0 && g.r_off % 4 == 0 && g.r_bstride % 4 == 0));
}
int launch_planes_dma(const ConvGemmGroup& gg, int variant, hipStream_t st) {
    const ConvGemm& g = gg.g[0];
    bool conv = gg.n > 1;
    for (int i = 0; i < gg.n; ++i) conv = conv || gg.g[i].taps > 1 || gg.g[i].cp_silu;
    const int bn = variant == 13 || variant == 14 ? 64 : 128;
    SVA_CHECK(variant == 9 || variant == 10 || variant == 11 || (!conv && variant == 12) || (conv && (variant == 13 || variant == 14) && g.N % bn == 0), "planes_dma: variant");
    for (int i = 0; i < gg.n; ++i)
        SVA_CHECK(planes_dma_supported(gg.g[i], conv) && g.N % bn == 0 && gg.g[i].M == g.M && gg.g[i].N == g.N && gg.g[i].Cin == g.Cin && gg.g[i].T == g.T && gg.g[i].pmode == g.pmode,
                  "planes_dma: unsupported problem (A as planes, N % 128 == 0, group members of one shape)");
    // 9: 128 x 128, four stages, one workgroup per CU; 10: 128 x 128, two stages, TWO workgroups per CU (one's epilogue under the other's K steps).
    // Measured and NOT instantiated (profiles/r05_planes_dma_bench_all_variants.txt; the template still takes them): 256 x 128 on 8 waves /
    // three stages (was 8), 256 x 256 on 16 waves / two stages (11: half the operand bytes per flop, 56 % MFMA use inside its K loop, but
    // one workgroup per CU leaves its epilogue and the 1.5-round tile counts of these shapes exposed), 128 x 256 on 8 waves (12) -- none
    // wins on any encoder shape
    if (conv) {
        // conv taps / a group / SiLU'd output planes: the HiFiGAN levels' ResBlock convs.  Narrow outputs take narrow tiles on four waves -- 13: 128 x 64,
        // two stages (C = 64); 14: 64 x 64, four stages (C >= 128 with too few rows for 128 x 128 tiles to fill the chip: a workgroup there holds one
        // tile, so only the ring depth hides the fill latency -- two stages 77 us, four 54).  (128 x 32 on two waves, C = 32: measured equal to
        // voc_conv_kernel, which also takes C = 16 -- not instantiated.)
        if (g.pmode == PLANES_H3) {
            switch (variant) {
                case 11: return launch_planes_dma_t<PLANES_H3, 128, 4, true, 128, 8, 2>(gg, st);
                case 9: return launch_planes_dma_t<PLANES_H3, 128, 4, true>(gg, st);
                case 10: return launch_planes_dma_t<PLANES_H3, 128, 2, true>(gg, st);
                case 13: return launch_planes_dma_t<PLANES_H3, 128, 2, true, 64, 4>(gg, st);
                default: return launch_planes_dma_t<PLANES_H3, 64, 4, true, 64, 4>(gg, st);
            }
        }
        switch (variant) {
            case 11: return launch_planes_dma_t<PLANES_H1, 128, 4, true, 128, 8, 2>(gg, st);
            case 9: return launch_planes_dma_t<PLANES_H1, 128, 4, true>(gg, st);
            case 10: return launch_planes_dma_t<PLANES_H1, 128, 2, true>(gg, st);
            case 13: return launch_planes_dma_t<PLANES_H1, 128, 2, true, 64, 4>(gg, st);
            default: return launch_planes_dma_t<PLANES_H1, 64, 4, true, 64, 4>(gg, st);
        }
    }
    // 11: variant 9's tile and ring with TWO LOADER WAVES beside the eight multiplying ones; 12: variant 10's (two stages, two workgroups per CU) with them
    if (variant == 12)
        return g.pmode == PLANES_H3 ? launch_planes_dma_t<PLANES_H3, 256, 3, false, 128, 8, 2>(gg, st) : launch_planes_dma_t<PLANES_H1, 256, 3, false, 128, 8, 2>(gg, st);
    if (variant == 11)
        return g.pmode == PLANES_H3 ? launch_planes_dma_t<PLANES_H3, 128, 4, false, 128, 8, 2>(gg, st) : launch_planes_dma_t<PLANES_H1, 128, 4, false, 128, 8, 2>(gg, st);
    if (g.pmode == PLANES_H3) return variant == 9 ? launch_planes_dma_t<PLANES_H3, 128, 4, false>(gg, st) : launch_planes_dma_t<PLANES_H3, 128, 2, false>(gg, st);
    return variant == 9 ? launch_planes_dma_t<PLANES_H1, 128, 4, false>(gg, st) : launch_planes_dma_t<PLANES_H1, 128, 2, false>(gg, st);
}

// fp32 [rows][K] (row stride ld) -> K-blocked planes (planes_split.h): one thread per 8 consecutive k of a row
template <int MODE>
__global__ void to_planes_kernel(const float* __restrict__ src, long rows, int K, long ld, unsigned short* __restrict__ dst, long pstride, float scale, int silu) {
    constexpr int NPL = PM<MODE>::NPL;
    const int k8 = K / 8;
    const long n8 = rows * k8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long row = i / k8;
        const int k = (int)(i - row * k8) * 8;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(src + row * ld + k), v1 = *reinterpret_cast<const f32x4*>(src + row * ld + k + 4);
        v0 *= scale; v1 *= scale;
        if (silu) {
            v0.x = silu_f(v0.x); v0.y = silu_f(v0.y); v0.z = silu_f(v0.z); v0.w = silu_f(v0.w);
            v1.x = silu_f(v1.x); v1.y = silu_f(v1.y); v1.z = silu_f(v1.z); v1.w = silu_f(v1.w);
        }
        u32x4 o[NPL];
        split8<MODE>(v0, v1, o);
        const long po = plane_off_blocked(row, k, rows);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dst + (long)p * pstride + po) = o[p];
    }
}

// rows [row0, row0 + T) of every stream of an activation tensor [nb][rows_b][K] -> the same dense rows of its planes mirror (nb * rows_b rows)
template <int MODE>
__global__ void to_planes_act_kernel(const float* __restrict__ src, int nb, long rows_b, long row0, int T, int K, unsigned short* __restrict__ dst, long pstride, int silu, int blocked) {
    constexpr int NPL = PM<MODE>::NPL;
    const int k8 = K / 8;
    const long n8 = (long)nb * T * k8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / k8;
        const int k = (int)(i - m * k8) * 8;
        const long b = m / T, row = b * rows_b + row0 + (m - b * T);
        f32x4 v0 = *reinterpret_cast<const f32x4*>(src + row * K + k), v1 = *reinterpret_cast<const f32x4*>(src + row * K + k + 4);
        if (silu) {
            v0.x = silu_f(v0.x); v0.y = silu_f(v0.y); v0.z = silu_f(v0.z); v0.w = silu_f(v0.w);
            v1.x = silu_f(v1.x); v1.y = silu_f(v1.y); v1.z = silu_f(v1.z); v1.w = silu_f(v1.w);
        }
        u32x4 o[NPL];
        split8<MODE>(v0, v1, o);
        const long po = blocked ? plane_off_blocked(row, k, (long)nb * rows_b) : row * K + k;
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dst + (long)p * pstride + po) = o[p];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// voc_conv_kernel: one conv stage (c1 or c2) of the three ResBlock branches of a NARROW HiFiGAN level (C = 16 / 32 channels; sva_common.h:
// VocConv).  The LDS-DMA GEMM above re-fetches the input rows once per tap (K = taps * C in 32-wide steps: at C = 32 every step is a whole
// new A piece for 32 output columns) and is bound by that fill; here a workgroup (four waves) keeps
//   * its branch's WHOLE weight as MFMA fragments in LDS (C = 32, k = 11: 44 KiB; loaded once, the workgroup walks tiles of one branch),
//   * per tile of BM output rows the input rows WITH their halo, (BM + (taps - 1) dil) x C fp16 parts per plane, as the row-major image
//     the global tensor has (one contiguous run per plane -> 1 KiB LDS-DMA pieces),
// and every K step's A operand is a SHIFTED ds_read_b128 of that image: rows + tap * dil (C = 32: one tap per 32-k step; C = 16: two taps
// per step, lanes' k chunks 0-1 / 2-3 read tap 2s / 2s + 1, the odd last tap meets zero weights).  Products transposed as above (weights as
// the row operand): a lane ends with four consecutive channels of one row -- bias, residual, fp32 store and the parts of silu(.) for the
// next conv from registers.  Workgroups are split over the branches in proportion to their taps (11 : 7 : 3).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int MODE, int C, int BM>
__global__ __launch_bounds__(256, C == 16 ? 4 : 2) void voc_conv_kernel(const VocConvGroup gg) {
    constexpr int NPL = PM<MODE>::NPL, NI = C / 16, MI = BM / 64;
    constexpr int KS_MAX = C == 32 ? 11 : 6;
    constexpr int ROWB = C * 2, RPP = 1024 / ROWB;          // bytes per row of a plane; rows per 1 KiB piece
    constexpr int R_MAX = BM + 64, A_PLANE = R_MAX * ROWB;  // image rows (tile + halo <= 50, whole pieces)
    constexpr int W_BYTES = NPL * KS_MAX * NI * 1024;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds0 = uni((unsigned)(size_t)lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)uni((unsigned)(tid >> 6));

    int pi = 0;
    while (pi + 1 < gg.n && (int)blockIdx.x >= gg.wg0[pi + 1]) ++pi;
    const VocConv& g = gg.g[pi];
    const int nw = gg.wg0[pi + 1] - gg.wg0[pi], wi = (int)blockIdx.x - gg.wg0[pi];
    const int tiles_b = gg.T / BM, n_tiles = gg.B * tiles_b;
    const int taps = g.taps, dil = g.dil;
    const int KS = C == 32 ? taps : (taps + 1) / 2;
    if (wi >= n_tiles) return;

    // ---- the branch's weight fragments: K-blocked planes [plane][ks][C rows][32] -> fragment (plane, ks, 16-row block j) lane-linear ----
    for (int f = wave; f < NPL * KS * NI; f += 4) {
        const int pl = f / (KS * NI), r = f - pl * (KS * NI);             // r = ks * NI + j: 16 rows x 32 k = the r-th KiB of the plane
        const char* src = reinterpret_cast<const char*>(g.Wp + (long)pl * g.wp_pstride) + (long)r * 1024;
        glds16(uni_ptr(src), (unsigned)(((lane & 15) * 32 + (lane >> 4) * 8) * 2), uni(lds0 + (unsigned)f * 1024));
    }
    const float winv = g.wp_inv;
    const int padL = (taps - 1) * dil;
    const int npieces = (BM + padL + RPP - 1) / RPP;
    const int rl = wave * (BM / 4) + (lane & 15), c = lane >> 4;

    for (int tile = wi; tile < n_tiles; tile += nw) {
        const int b = tile / tiles_b, t0 = (tile - b * tiles_b) * BM;
        const long row_in0 = (long)b * g.a_rows_b + g.a_row0 + t0;
        __builtin_amdgcn_s_barrier();              // (every wave has read the previous tile's image)
        for (int q = wave; q < npieces * NPL; q += 4) {
            const int pl = q / npieces, pc = q - pl * npieces;
            long r = row_in0 + (long)pc * RPP + (lane * 16) / ROWB;
            if (r > g.a_rows_total - 1) r = g.a_rows_total - 1;        // (the whole pieces' tail rows past the tile's halo: never used)
            const char* base = reinterpret_cast<const char*>(g.Ap + (long)pl * g.ap_pstride + row_in0 * C);
            glds16(uni_ptr(base), (unsigned)((r - row_in0) * ROWB + (lane * 16) % ROWB), uni(lds0 + (unsigned)(W_BYTES + pl * A_PLANE + pc * 1024)));
        }
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();

        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* const aimg = lds + W_BYTES;
        for (int ks = 0; ks < KS; ++ks) {
            int tap, coff;
            if (C == 32) { tap = ks; coff = c * 16; }
            else { tap = 2 * ks + (c >> 1); if (tap > taps - 1) tap = taps - 1; coff = (c & 1) * 16; }
            const int roff = (rl + tap * dil) * ROWB + coff;
            u32x4 fa[MI][NPL], fb[NI][NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j][p] = *reinterpret_cast<const u32x4*>(lds + ((p * KS + ks) * NI + j) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][p] = *reinterpret_cast<const u32x4*>(aimg + p * A_PLANE + roff + i * 16 * ROWB);
            }
            if constexpr (MODE == PLANES_H3) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][0], fa[i][1], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][1], fa[i][0], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mma1<MODE>(fb[j][0], fa[i][0], acc[i][j]);
        }
        // ---- epilogue: registers -> global ----
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int t = t0 + rl + i * 16;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = j * 16 + 4 * c;
                f32x4 v = acc[i][j] * winv;
                if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (g.res) v += *reinterpret_cast<const f32x4*>(g.res + (long)b * g.r_bstride + g.r_off + (long)t * C + n);
                if (gg.ovf && !(fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w) < INFINITY)) *reinterpret_cast<volatile int*>(gg.ovf) = 1;
                if (g.Cf) *reinterpret_cast<f32x4*>(g.Cf + (long)b * g.c_bstride + g.c_off + (long)t * C + n) = v;
                if (g.Cp) {
                    unsigned p0[NPL], p1[NPL];
                    split_pair<MODE>(silu_f(v.x), silu_f(v.y), p0);
                    split_pair<MODE>(silu_f(v.z), silu_f(v.w), p1);
                    const long po = ((long)b * g.c_rows_b + g.c_row0 + t) * C + n;
#pragma unroll
                    for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x2*>(g.Cp + (long)p * g.cp_pstride + po) = (u32x2){p0[p], p1[p]};
                }
            }
        }
    }
}

template <int MODE, int C, int BM>
int launch_voc_conv_t(VocConvGroup& gg, int cus, hipStream_t st) {
    constexpr int NPL = PM<MODE>::NPL;
    constexpr size_t smem = (size_t)NPL * (C == 32 ? 11 : 6) * (C / 16) * 1024 + (size_t)NPL * (BM + 64) * C * 2;
    constexpr int WG_PER_CU = C == 16 ? 4 : 2;
    static_assert(smem * WG_PER_CU <= 160 * 1024, "LDS of one CU");
    static DeviceOnce attr_set;
    if (attr_set.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)voc_conv_kernel<MODE, C, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set.done();
    }
    const int n_tiles = gg.B * (gg.T / BM);
    // longest K first; workgroups in proportion to the members' taps (each member's tiles are equally many), at most one per tile
    std::stable_sort(gg.g, gg.g + gg.n, [](const VocConv& a, const VocConv& b_) { return a.taps > b_.taps; });
    int sum_taps = 0;
    for (int i = 0; i < gg.n; ++i) sum_taps += gg.g[i].taps;
    const int G = std::min(n_tiles * gg.n, cus * WG_PER_CU);
    gg.wg0[0] = 0;
    for (int i = 0; i < gg.n; ++i) {
        int w = std::max(1, (int)((long)G * gg.g[i].taps / sum_taps));
        w = std::min(w, n_tiles);
        gg.wg0[i + 1] = gg.wg0[i] + w;
    }
    hipLaunchKernelGGL((voc_conv_kernel<MODE, C, BM>), dim3(gg.wg0[gg.n]), dim3(256), smem, st, gg);
    return 0;
}

}  // namespace

void planes_dma_set_cu_limit(int cus) { g_dma_cu_limit = cus; }
bool planes_dma_gemm_supported(const ConvGemm& g) { return planes_dma_supported(g); }
bool planes_dma_conv_supported(const ConvGemm& g) { return planes_dma_supported(g, true); }

int planes_count(int mode) { return mode == PLANES_H3 ? 2 : 1; }

// channels in whole 32-wide K tiles, planes of the weights present; the tiled epilogue's conditions (16-byte aligned C rows) are the caller's
bool planes_gemm_supported(const ConvGemm& g) {
    return g.Wp && (g.pmode == PLANES_H3 || g.pmode == PLANES_H1) && g.Cin % 32 == 0 && g.stride >= 1 && !g.rms_w && !g.dw_wT && (g.C || g.Cp) && !(g.accumulate && !g.C) &&
           (g.A || g.Ap) &&
           (!g.Ap || (g.taps >= 1 && g.stride == 1 && g.lda > 0 && g.a_off % g.lda == 0 && g.a_bstride % g.lda == 0 && g.ap_pstride % 8 == 0 && g.ap_rows > 0)) &&
           (!g.Cp || (g.ldc > 0 && g.ldc % 4 == 0 && g.c_off % g.ldc == 0 && g.c_bstride % g.ldc == 0 && g.cp_pstride % 4 == 0 && g.cp_rows > 0 && (g.w13 ? g.N / 2 : g.N) % 32 == 0));
}

int launch_planes_gemm(const ConvGemmGroup& gg, int variant, hipStream_t st) {
    const ConvGemm& g = gg.g[0];
    SVA_CHECK(planes_gemm_supported(g), "planes_gemm: unsupported problem");
    for (int i = 1; i < gg.n; ++i)
        SVA_CHECK(planes_gemm_supported(gg.g[i]) && gg.g[i].pmode == g.pmode && !gg.g[i].Ap == !g.Ap, "planes_gemm: group members differ");
    if (g.Ap) {
        SVA_CHECK(!g.a_silu, "planes_gemm: SiLU belongs to the producer of the planes");
        if (variant >= 8) return launch_planes_dma(gg, variant, st);
        for (int i = 0; i < gg.n; ++i) SVA_CHECK(gg.g[i].taps == 1 && !gg.g[i].cp_silu, "planes_gemm: conv taps over A planes / SiLU'd output planes need the LDS-DMA form");
        return g.pmode == PLANES_H3 ? launch_planes_m<PLANES_H3, true>(gg, variant, st) : launch_planes_m<PLANES_H1, true>(gg, variant, st);
    }
    SVA_CHECK(variant < 8, "planes_gemm: the LDS-DMA variants take the A operand as planes");
    for (int i = 0; i < gg.n; ++i) SVA_CHECK(!gg.g[i].cp_silu, "planes_gemm: SiLU'd output planes need the LDS-DMA form");
    return g.pmode == PLANES_H3 ? launch_planes_m<PLANES_H3, false>(gg, variant, st) : launch_planes_m<PLANES_H1, false>(gg, variant, st);
}

int launch_to_planes(const float* src, long rows, int K, long ld, unsigned short* dst, long pstride, int mode, float scale, int silu, hipStream_t st) {
    SVA_CHECK(K % 32 == 0 && ld % 4 == 0 && (mode == PLANES_H3 || mode == PLANES_H1), "to_planes: whole 32-k blocks, an fp16 planes format");
    const long n8 = rows * (K / 8);
    const int blocks = (int)std::min<long>((n8 + 255) / 256, 2048);
    if (mode == PLANES_H3) hipLaunchKernelGGL(to_planes_kernel<PLANES_H3>, dim3(blocks), dim3(256), 0, st, src, rows, K, ld, dst, pstride, scale, silu);
    else hipLaunchKernelGGL(to_planes_kernel<PLANES_H1>, dim3(blocks), dim3(256), 0, st, src, rows, K, ld, dst, pstride, scale, silu);
    SVA_HIP(hipGetLastError());
    return 0;
}

bool voc_conv_supported(int C, int T, int mode) {
    return (mode == PLANES_H3 || mode == PLANES_H1) && ((C == 32 && T % 128 == 0) || (C == 16 && T % 256 == 0));
}
int launch_voc_conv(VocConvGroup gg, int C, int mode, int cu_limit, hipStream_t st) {
    SVA_CHECK(voc_conv_supported(C, gg.T, mode) && gg.n >= 1 && gg.n <= 3 && gg.B >= 1, "voc_conv: C = 16 / 32, whole tiles per stream");
    for (int i = 0; i < gg.n; ++i)
        SVA_CHECK(gg.g[i].Ap && gg.g[i].Wp && (gg.g[i].Cf || gg.g[i].Cp) && gg.g[i].taps >= 1 && gg.g[i].taps <= 11 && (gg.g[i].taps - 1) * gg.g[i].dil <= 50,
                  "voc_conv: operands as planes, at most 11 taps and 50 halo rows");
    int cus = cu_limit > 0 ? cu_limit : g_dma_cu_limit;
    if (cus <= 0) {
        int dev = 0;
        SVA_HIP(hipGetDevice(&dev));
        SVA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    int rc;
    if (mode == PLANES_H3) rc = C == 32 ? launch_voc_conv_t<PLANES_H3, 32, 128>(gg, cus, st) : launch_voc_conv_t<PLANES_H3, 16, 256>(gg, cus, st);
    else rc = C == 32 ? launch_voc_conv_t<PLANES_H1, 32, 128>(gg, cus, st) : launch_voc_conv_t<PLANES_H1, 16, 256>(gg, cus, st);
    SVA_HIP(hipGetLastError());
    return rc;
}

int launch_to_planes_act(const float* src, int nb, long rows_b, long row0, int T, int K, unsigned short* dst, long pstride, int mode, int silu, hipStream_t st, int blocked) {
    SVA_CHECK((blocked ? K % 32 == 0 : K % 8 == 0) && (mode == PLANES_H3 || mode == PLANES_H1), "to_planes_act: whole 32-k blocks, an fp16 planes format");
    const long n8 = (long)nb * T * (K / 8);
    const int blocks = (int)std::min<long>((n8 + 255) / 256, 4096);
    if (mode == PLANES_H3) hipLaunchKernelGGL(to_planes_act_kernel<PLANES_H3>, dim3(blocks), dim3(256), 0, st, src, nb, rows_b, row0, T, K, dst, pstride, silu, blocked);
    else hipLaunchKernelGGL(to_planes_act_kernel<PLANES_H1>, dim3(blocks), dim3(256), 0, st, src, nb, rows_b, row0, T, K, dst, pstride, silu, blocked);
    SVA_HIP(hipGetLastError());
    return 0;
}

// Planes of a weight matrix W [N][K] already on the device: dst = [NPL] K-blocked planes of N rows, element = part of W * 2^e with e chosen so
// that max |W| 2^e lies in [2^7, 2^8) in the fp16 modes (the low part of a weight 2^-9 of the largest is still a normal fp16; far
// from the fp16 overflow threshold) and e = 0 for bf16 (fp32's exponent range).  *inv = 2^-e for the accumulator.
int make_weight_planes(const float* dW, int N, int K, float max_abs, int mode, unsigned short* dst, float* inv, hipStream_t st) {
    int e = 0;
    if (max_abs > 0.f && std::isfinite(max_abs)) {
        int ex;
        (void)frexpf(max_abs, &ex);            // max_abs = f * 2^ex, f in [0.5, 1)
        e = 8 - ex;
    }
    *inv = ldexpf(1.f, -e);
    return launch_to_planes(dW, N, K, K, dst, (long)N * K, mode, ldexpf(1.f, e), 0, st);
}

}  // namespace sva
