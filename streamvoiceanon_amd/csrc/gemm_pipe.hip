// Deep-prefetch f32-MFMA conv-GEMM for gfx950: 64-float-deep K tiles, RS register stages of global loads in flight per
// thread (global -> registers -> LDS), XOR-swizzled LDS double buffer, one barrier per K tile.
//
// Why: at one stream every GEMM of the encoder / vocoder is a few hundred MFLOP on a 128..160-row activation.  The kernel of
// gemm.hip has ONE K tile in flight per workgroup, so each K step costs a full L2 / HBM round trip; the barrier-free small-M
// kernel avoids that with 16-column tiles but re-reads the A panel N / 16 times.  Here every thread keeps RS tiles' worth of
// 16-byte loads outstanding (64..96 KiB per workgroup), so the K loop runs at the rate the CU can pull data, not at latency.
// (A first version fed the ring with LDS-DMA, global_load_lds_dwordx4: correct, but one CU lands only ~13-25 GB/s through that
// path -- tools/gemm_pipe_sweep.py: time proportional to bytes per workgroup -- so the loads go through registers.)
//
// LDS layout of a stage: rows of 64 floats = 16 chunks of 16 B; chunk c of tile row r sits at chunk position c ^ (r & 15).
// Writes: 16 consecutive lanes store the 16 chunks of one row (a permutation of one 256-byte row: conflict-free).  The MFMA
// fragment reads -- lane (fr, fk) takes chunk 4 kb + fk of row r0 + fr, one ds_read_b128 per 16 k, same k-permutation trick as
// the other kernels -- touch 16 distinct chunk positions per 16 lanes: conflict-free.
#include "sva_common.h"
#include "device_util.h"

namespace sva {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float silu_p(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_p(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int BM, int BN, int RS, int PRO>
__global__ __launch_bounds__(256) void pipe_gemm_kernel(const ConvGemmGroup gg) {
    constexpr int BK = 64, WM = 2, WN = 2;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int STAGE_FLOATS = (BM + BN) * BK;
    constexpr int A_LD = BM * 16 / 256, B_LD = BN * 16 / 256;        // 16-byte chunks per thread and stage
    static_assert(RS >= 2 && RS <= 4, "register stages");
    const ConvGemm& g = gg.g[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tbx = blockIdx.x, tby = blockIdx.y;
    xcd_tile(gg.xcd_swz, gridDim.x, gridDim.y, tbx, tby);
    const int bm0 = tby * BM, bn0 = tbx * BN;
    const long Kt = (long)g.taps * g.Cin;

    // chunk q = tid + 256 j of a tile: row q >> 4, chunk q & 15 (16 consecutive lanes = one 256-byte row: coalesced)
    const int crow = tid >> 4, cc = tid & 15;
    const float* aptr[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        int m = bm0 + crow + 16 * j;
        if (m > g.M - 1) m = g.M - 1;
        const int b = m / g.T, t = m - b * g.T;
        aptr[j] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + cc * 4;
    }
    const float* bptr[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        int n = bn0 + crow + 16 * j;
        if (n > g.N - 1) n = g.N - 1;
        bptr[j] = g.W + (long)n * Kt + cc * 4;
    }
    const int kc_tiles = g.Cin / BK;
    const int nk = g.taps * kc_tiles;
    float* nws = smem + 2 * STAGE_FLOATS;          // [Cin <= 2048] norm weight of the fused RMSNorm
    if constexpr (PRO == 2)
        for (int i = tid; i < g.Cin; i += 256) nws[i] = g.rms_w[i];

    // register stages as separately named arrays touched only through macros (a [RS][..] array handed to lambdas by reference
    // ends up in scratch memory, which serialises every load)
    // (native vector type: an assignment of the float4 STRUCT lowers to a memcpy into a private array, which is never promoted)
    f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD], ra2[A_LD], rb2[B_LD], ra3[A_LD], rb3[B_LD];
#define PIPE_LOAD(RA, RB, KT)                                                                         \
    do {                                                                                              \
        const int kt_ = (KT);                                                                         \
        const int tap_ = kt_ / kc_tiles, kc_ = (kt_ - tap_ * kc_tiles) * BK;                          \
        const long aoff_ = (long)tap_ * g.dil * g.lda + kc_, boff_ = (long)tap_ * g.Cin + kc_;        \
        _Pragma("unroll") for (int j = 0; j < A_LD; ++j) RA[j] = *reinterpret_cast<const f32x4*>(aptr[j] + aoff_); \
        _Pragma("unroll") for (int j = 0; j < B_LD; ++j) RB[j] = *reinterpret_cast<const f32x4*>(bptr[j] + boff_); \
    } while (0)
    const int wpos = (cc ^ (crow & 15)) << 2;      // swizzled chunk position of this thread's chunks (row & 15 is the same for all j)
#define PIPE_STORE(RA, RB, BUF)                                                                       \
    do {                                                                                              \
        float* As_ = smem + (BUF) * STAGE_FLOATS;                                                     \
        float* Bs_ = As_ + BM * BK;                                                                   \
        _Pragma("unroll") for (int j = 0; j < A_LD; ++j) *reinterpret_cast<f32x4*>(As_ + (crow + 16 * j) * BK + wpos) = RA[j]; \
        _Pragma("unroll") for (int j = 0; j < B_LD; ++j) *reinterpret_cast<f32x4*>(Bs_ + (crow + 16 * j) * BK + wpos) = RB[j]; \
    } while (0)

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ssq[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) ssq[i] = 0.f;

    const int fr = lane & 15, fk = lane >> 4;
    auto compute = [&](int kt) {
        const float* As = smem + (kt & 1) * STAGE_FLOATS;
        const float* Bs = As + BM * BK;
        const float* Ab = As + (wm * TM + fr) * BK;
        const float* Bb = Bs + (wn * TN + fr) * BK;
        const int kc = (kt % kc_tiles) * BK;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int pos = ((kb * 4 + fk) ^ fr) << 2;
            float4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 16 * BK + pos);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 16 * BK + pos);
            if constexpr (PRO == 1) {
#pragma unroll
                for (int i = 0; i < MI; ++i) { af[i].x = silu_p(af[i].x); af[i].y = silu_p(af[i].y); af[i].z = silu_p(af[i].z); af[i].w = silu_p(af[i].w); }
            }
            if constexpr (PRO == 2) {         // fused RMSNorm (taps == 1): row statistics of the raw rows, weight folded into the operand
                const float4 nw = *reinterpret_cast<const float4*>(nws + kc + kb * 16 + 4 * fk);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    // explicit sequential fmas: under -ffp-contract the pairwise form was fused differently per unrolled row tile (gemm_stream.hip), so a row's statistic depended on its position
                    ssq[i] = __builtin_fmaf(af[i].x, af[i].x, ssq[i]);
                    ssq[i] = __builtin_fmaf(af[i].y, af[i].y, ssq[i]);
                    ssq[i] = __builtin_fmaf(af[i].z, af[i].z, ssq[i]);
                    ssq[i] = __builtin_fmaf(af[i].w, af[i].w, ssq[i]);
                    af[i].x *= nw.x; af[i].y *= nw.y; af[i].z *= nw.z; af[i].w *= nw.w;
                }
            }
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
    };
    // RS tiles in flight: tile kt is written to LDS from register stage kt % RS, whose registers are then refilled with tile
    // kt + RS.  The steady-state loop body is branch-free straight-line code (tile indices beyond the end are clamped: a harmless
    // re-load), because hipcc's vmcnt bookkeeping falls back to "wait for everything" at control-flow joins -- which would leave
    // one tile in flight instead of RS; the nk % RS leftover tiles run after the loop.
#define PIPE_STEP(RA, RB, KT)                                                                         \
    {                                                                                                 \
        PIPE_STORE(RA, RB, (KT) & 1);                                                                 \
        PIPE_LOAD(RA, RB, ((KT) + RS < nk ? (KT) + RS : nk - 1));                                     \
        __syncthreads(); /* tile KT is in LDS; every wave is done reading tile KT - 1 (the buffer the next store reuses) */ \
        compute(KT);                                                                                  \
    }
    PIPE_LOAD(ra0, rb0, 0);
    PIPE_LOAD(ra1, rb1, (1 < nk ? 1 : nk - 1));
    if constexpr (RS > 2) PIPE_LOAD(ra2, rb2, (2 < nk ? 2 : nk - 1));
    if constexpr (RS > 3) PIPE_LOAD(ra3, rb3, (3 < nk ? 3 : nk - 1));
    int kt0 = 0;
    for (; kt0 + RS <= nk; kt0 += RS) {
        PIPE_STEP(ra0, rb0, kt0)
        PIPE_STEP(ra1, rb1, kt0 + 1)
        if constexpr (RS > 2) PIPE_STEP(ra2, rb2, kt0 + 2)
        if constexpr (RS > 3) PIPE_STEP(ra3, rb3, kt0 + 3)
    }
    if (kt0 < nk) {
        PIPE_STEP(ra0, rb0, kt0)
        if (kt0 + 1 < nk) {
            PIPE_STEP(ra1, rb1, kt0 + 1)
            if constexpr (RS > 3) {
                if (kt0 + 2 < nk) PIPE_STEP(ra2, rb2, kt0 + 2)
            }
        }
    }
#undef PIPE_STEP
#undef PIPE_STORE
#undef PIPE_LOAD
    __syncthreads();                 // all fragment reads done: the ring is reused as the epilogue staging tile

    // ---- epilogue (same as conv_gemm_kernel: accumulators staged through LDS, 16-byte row-contiguous stores) ----
    constexpr int CS = BN + 4;
    float* Cs = smem;                              // [BM][CS]
    float* rs = smem + BM * CS;                    // [BM] row sums of squares (fused RMSNorm)
    const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wm * TM + i * 16 + rq + r) * CS + wn * TN + j * 16 + col] = acc[i][j][r];
    if (PRO == 2 && wn == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fk == 0) rs[wm * TM + i * 16 + fr] = v;
        }
    }
    __syncthreads();
    if (g.w13) {
        constexpr int OC4 = BN / 8;
        for (int idx = tid; idx < BM * OC4; idx += 256) {
            const int row = idx / OC4, q = idx - row * OC4;
            const int m = bm0 + row;
            const int grp = q >> 2, c4 = (q & 3) * 4;
            const int n = bn0 + grp * 32 + c4;
            if (m >= g.M || n >= g.N) continue;
            const int b = m / g.T, t = m - b * g.T;
            if (t >= g.skip_lo && t < g.skip_hi) continue;
            float4 a = *reinterpret_cast<const float4*>(&Cs[row * CS + grp * 32 + c4]);
            float4 w = *reinterpret_cast<const float4*>(&Cs[row * CS + grp * 32 + 16 + c4]);
            if constexpr (PRO == 2) {
                const float inv = 1.f / sqrtf(rs[row] / (float)Kt + g.rms_eps);
                a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv; w.x *= inv; w.y *= inv; w.z *= inv; w.w *= inv;
            }
            float4 o;
            o.x = silu_p(a.x) * w.x; o.y = silu_p(a.y) * w.y; o.z = silu_p(a.z) * w.z; o.w = silu_p(a.w) * w.w;
            float* crow = g.C + (long)b * g.c_bstride + g.c_off + (long)t * g.ldc;
            *reinterpret_cast<float4*>(crow + ((bn0 + grp * 32) >> 1) + c4) = o;
        }
        return;
    }
    constexpr int C4 = BN / 4;
    for (int idx = tid; idx < BM * C4; idx += 256) {
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        const int m = bm0 + row, n = bn0 + c4;
        if (m >= g.M || n >= g.N) continue;
        const int b = m / g.T, t = m - b * g.T;
        if (t >= g.skip_lo && t < g.skip_hi) continue;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * CS + c4]);
        if constexpr (PRO == 2) {
            const float inv = 1.f / sqrtf(rs[row] / (float)Kt + g.rms_eps);
            v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        }
        if (g.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (g.act == ACT_GELU) { v.x = gelu_p(v.x); v.y = gelu_p(v.y); v.z = gelu_p(v.z); v.w = gelu_p(v.w); }
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
        float* cp = g.C + (long)b * g.c_bstride + g.c_off + (long)t * g.ldc + n;
        if (g.accumulate) {
            const float4 cc = *reinterpret_cast<const float4*>(cp);
            v.x += cc.x; v.y += cc.y; v.z += cc.z; v.w += cc.w;
        }
        *reinterpret_cast<float4*>(cp) = v;
    }
}

template <int BM, int BN, int RS, int PRO>
int launch_pipe_p(const ConvGemmGroup& gg_in, hipStream_t st) {
    ConvGemmGroup gg = gg_in;
    constexpr size_t ring = (size_t)2 * (BM + BN) * 64 * sizeof(float) + 2048 * sizeof(float);
    constexpr size_t epi = ((size_t)BM * (BN + 4) + BM) * sizeof(float);
    constexpr size_t smem = ring > epi ? ring : epi;
    static_assert(smem <= 160 * 1024, "LDS");
    static DeviceOnce attr;
    if (attr.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)pipe_gemm_kernel<BM, BN, RS, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr.done();
    }
    const ConvGemm& g = gg.g[0];
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, gg.n);
    gg.xcd_swz = xcd_swizzle_for(grid.x, grid.y);
    hipLaunchKernelGGL((pipe_gemm_kernel<BM, BN, RS, PRO>), grid, dim3(256), smem, st, gg);
    return 0;
}

template <int BM, int BN, int RS>
int launch_pipe(const ConvGemmGroup& gg, hipStream_t st) {
    if (gg.g[0].rms_w) return launch_pipe_p<BM, BN, RS, 2>(gg, st);
    if (gg.g[0].a_silu) return launch_pipe_p<BM, BN, RS, 1>(gg, st);
    return launch_pipe_p<BM, BN, RS, 0>(gg, st);
}

}  // namespace

bool pipe_gemm_supported(const ConvGemm& g) {
    return g.Cin % 64 == 0 && !g.dw_wT && g.N % 4 == 0 && g.ldc % 4 == 0 && g.c_off % 4 == 0 && g.c_bstride % 4 == 0 && g.lda % 4 == 0 &&
           g.a_off % 4 == 0 && g.a_bstride % 4 == 0 && (!g.res || (g.ldr % 4 == 0 && g.r_off % 4 == 0 && g.r_bstride % 4 == 0)) &&
           (!g.rms_w || (g.taps == 1 && g.Cin <= 2048)) && (!g.w13 || g.N % 32 == 0);
}

// variant (tile, register stages): 0 = 32x64 (4), 1 = 64x64 (3), 2 = 128x64 (3), 3 = 64x128 (3), 4 = 128x128 (2), 5 = 32x128 (3), 6 = 32x32 (4)
int launch_pipe_gemm(const ConvGemmGroup& gg, int variant, hipStream_t st) {
    switch (variant) {
        case 0: return launch_pipe<32, 64, 4>(gg, st);
        case 1: return launch_pipe<64, 64, 3>(gg, st);
        case 2: return launch_pipe<128, 64, 3>(gg, st);
        case 3: return launch_pipe<64, 128, 3>(gg, st);
        case 4: return launch_pipe<128, 128, 2>(gg, st);
        case 5: return launch_pipe<32, 128, 3>(gg, st);
        case 6: return launch_pipe<32, 32, 4>(gg, st);
        default: set_error("pipe_gemm: bad variant"); return -1;
    }
}

}  // namespace sva
