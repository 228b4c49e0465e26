// Weight-streaming f32-MFMA GEMM for few rows (M <= ~256): the AR chain's linear layers at batch scale and the encoder / vocoder GEMMs of a
// single stream.  These problems are a few MB of weights against a few hundred KB of activations: what decides their time is how early every
// byte is requested, not the matrix pipe.  So, unlike the K loops of gemm.hip / gemm_pipe.hip:
//   * a workgroup owns a (16 MT) x (16 NT) output tile for the WHOLE K axis, its KW waves take the 16-k blocks round-robin, and every wave
//     requests ALL its operand fragments before it multiplies the first one (KB blocks = 4 (MT + NT) KB registers in flight per lane;
//     problems with more blocks per wave re-request a block as soon as one is consumed: the queue stays KB deep).  Straight-line code, no
//     load under a branch (hipcc drains the queue at control-flow joins);
//   * the weight fragments are requested first (they come from HBM / MALL, the activation rows from L2);
//   * optionally the weights are read from a fragment-major packing made at finalize (one wave-instruction = one contiguous KiB instead of
//     sixteen 64-byte row segments: 10-30 % per launch, profiles/r06_stream_sweep_1.txt; the non-temporal hint measured 3-6 % slower and is gone);
//   * the epilogue's operands (bias, gamma, residual) are requested before anything else, so the tail is arithmetic and stores only;
//   * tiles are numbered so that XCD x owns a contiguous band of column tiles with all their row tiles: a weight panel is fetched from the
//     fabric by one L2 (sva_common.h: workgroup b runs on XCD b % 8);
//   * the KW partial tiles meet in LDS once and EVERY thread sums and stores its share of the tile (the small-M kernel of gemm.hip leaves the
//     tail to MT waves).
// Operand mapping as in gemm.hip: lane l supplies, for MFMA step j of block kb, W[n0 + (l & 15)][16 kb + 4 (l >> 4) + j] and the same k of
// A[m0 + (l & 15)] -- a permutation of k inside the block, identical on both operands.
// Same ConvGemm semantics as the other kernels (taps over shifted rows, SiLU / RMSNorm prologues, bias / act / gamma / residual / scale /
// accumulate / SwiGLU epilogues, skipped history rows); every output element is the fixed-order sum wave 0 .. KW - 1 of fixed-order block sums.
#include "sva_common.h"
#include "device_util.h"

namespace sva {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float silu_s(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_s(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

struct StreamArgs {
    const float* A; long a_bstride, a_off; int lda, T, M, stride, dil, taps, Cin;
    const float* W; int N; int nkb;              // nkb = taps * Cin / 16
    const float* bias; const float* gamma; const float* res; long r_bstride, r_off; int ldr;
    float* C; long c_bstride, c_off; int ldc;
    float scale; int skip_lo, skip_hi, act, accumulate, w13;
    const float* rms_w; float rms_eps;
    int m_tiles, n_tiles;                        // workgroup tiles
};

// AOP: 0 nothing, 1 SiLU on A, 2 RMSNorm of the A rows (weight folded into the operand, row statistics applied in the epilogue)
// WP: weights in the fragment-major packing [N / 16][nkb][64 lanes][4]
// KB: blocks a wave holds in registers = the smallest instantiated count that covers its share of K (a block beyond the share is a clamped
//     re-load whose MFMAs are skipped by a wave-uniform branch -- a branch around arithmetic only, the loads stay unconditional)
// PROBE (timing diagnostics, results are garbage): 1 exit at once, 2 weight loads only, 3 weight + activation loads, no MFMA
// LOOP: some wave holds more than KB blocks (a consumed block's registers are re-requested at once, clamped indices instead of a branch)
template <int MT, int NT, int KW, int KB, int AOP, bool WP, int PROBE, bool LOOP>
__global__ __launch_bounds__(64 * KW) void stream_gemm_kernel(const StreamArgs g) {
    // no implicit contraction in this kernel: every row tile's arithmetic must round the same way (a row's result must not depend on its position)
#pragma clang fp contract(off)
    if constexpr (PROBE == 1) return;
    constexpr bool SILU = AOP == 1, RMS = AOP == 2;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [KW][MT*NT][64] f32x4 (+ [KW][MT][16] row sums of squares)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD x owns the x-th contiguous eighth of the tile sequence (row tiles fastest)
    int V;
    {
        const int Tn = g.m_tiles * g.n_tiles, L = blockIdx.x;
        const int xcd = L & 7, idx = L >> 3, q = Tn >> 3, r = Tn & 7;
        V = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tn = V / g.m_tiles, tm = V - tn * g.m_tiles;
    const int n0 = tn * (16 * NT), m_base = tm * (16 * MT);
    const int fr = lane & 15, fg = lane >> 4;
    const long Kt = (long)g.nkb * 16;
    const int kc_tiles = g.Cin >> 4;
    const float* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if constexpr (WP) {
            int t16 = (n0 >> 4) + j;
            const int last = (g.N - 1) >> 4;
            if (t16 > last) t16 = last;
            wp[j] = g.W + ((long)t16 * g.nkb * 64 + lane) * 4;
        } else {
            int n = n0 + j * 16 + fr;
            if (n > g.N - 1) n = g.N - 1;
            wp[j] = g.W + (long)n * Kt + 4 * fg;
        }
    }
    const float* ap[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m_base + i * 16 + fr;
        if (m > g.M - 1) m = g.M - 1;
        const int b = m / g.T, t = m - b * g.T;
        ap[i] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.stride * g.lda + 4 * fg;
    }
    const int my_n = g.nkb > wave ? (g.nkb - wave + KW - 1) / KW : 0;           // blocks of this wave: wave, wave + KW, ...
    const int last_kb = my_n > 0 ? wave + (my_n - 1) * KW : 0;

    f32x4 wv[KB][NT], av[KB][MT], nv[KB];
    auto issue_w = [&](f32x4 (&w)[NT], int kb) {
        kb = kb < g.nkb ? kb : last_kb;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x4* p;
            if constexpr (WP) p = reinterpret_cast<const f32x4*>(wp[j] + (long)kb * 256);
            else {
                const int tap = kb / kc_tiles;           // (W's K axis is the taps back to back: block kb sits at 16 kb)
                (void)tap;
                p = reinterpret_cast<const f32x4*>(wp[j] + (long)kb * 16);
            }
            w[j] = *p;
        }
    };
    auto issue_a = [&](f32x4 (&a)[MT], f32x4& nw, int kb) {
        kb = kb < g.nkb ? kb : last_kb;
        const int tap = kb / kc_tiles;
        const int kc = (kb - tap * kc_tiles) * 16;
        const long aoff = (long)tap * g.dil * g.lda + kc;
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap[i] + aoff);
        if constexpr (RMS) nw = *reinterpret_cast<const f32x4*>(g.rms_w + kc + 4 * fg);
    };
    // epilogue operands of this thread's output units, requested first (unit u = (row tile i, column tile j, lane slot l): see the tail)
    constexpr int UNITS = (MT * NT + KW - 1) / KW;
    const bool w13 = g.w13 != 0;
    float e_bias[UNITS], e_gamma[UNITS], e_res[UNITS][4];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
        e_bias[k] = 0.f; e_gamma[k] = 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) e_res[k][r] = 0.f;
    }
    if (!w13) {
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
            int u = tid + k * 64 * KW;
            if (u > MT * NT * 64 - 1) u = MT * NT * 64 - 1;
            const int l = u & 63, ij = u >> 6, i = ij / NT, jj = ij - i * NT;
            int n = n0 + jj * 16 + (l & 15);
            if (n > g.N - 1) n = g.N - 1;
            if (g.bias) e_bias[k] = g.bias[n];
            if (g.gamma) e_gamma[k] = g.gamma[n];
            if (g.res) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = m_base + i * 16 + (l >> 4) * 4 + r;
                    if (m > g.M - 1) m = g.M - 1;
                    const int b = m / g.T, tt = m - b * g.T;
                    e_res[k][r] = g.res[(long)b * g.r_bstride + g.r_off + (long)tt * g.ldr + n];
                }
            }
        }
    }
    // everything this wave needs (or its first KB blocks) is requested here, weights first
#pragma unroll
    for (int d = 0; d < KB; ++d) issue_w(wv[d], wave + d * KW);
#pragma unroll
    for (int d = 0; d < KB; ++d) issue_a(av[d], nv[d], wave + d * KW);
    __builtin_amdgcn_sched_barrier(0);          // hipcc's scheduler otherwise sinks the requests next to their uses (a queue two blocks deep)

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ssq[i] = 0.f;

    for (int it = 0; it < (LOOP ? my_n : 1); it += KB) {
#pragma unroll
        for (int d = 0; d < KB; ++d) {
            f32x4 w[NT], a[MT], nw;
#pragma unroll
            for (int j = 0; j < NT; ++j) w[j] = wv[d][j];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = av[d][i];
            nw = nv[d];
            if constexpr (LOOP) {
                issue_w(wv[d], wave + (it + d + KB) * KW);
                issue_a(av[d], nv[d], wave + (it + d + KB) * KW);
            }
            if (it + d < my_n) {                     // wave-uniform; no load inside
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if constexpr (SILU) { a[i].x = silu_s(a[i].x); a[i].y = silu_s(a[i].y); a[i].z = silu_s(a[i].z); a[i].w = silu_s(a[i].w); }
                    if constexpr (RMS) {
                        // explicit, sequential fmas: left to -ffp-contract the sum of squares was fused differently per unrolled row tile (fma(x, x, y * y)
                        // here, two products and an add there), which made a row's statistic -- and every logit after it -- depend on the row's position
                        ssq[i] = __builtin_fmaf(a[i].x, a[i].x, ssq[i]);
                        ssq[i] = __builtin_fmaf(a[i].y, a[i].y, ssq[i]);
                        ssq[i] = __builtin_fmaf(a[i].z, a[i].z, ssq[i]);
                        ssq[i] = __builtin_fmaf(a[i].w, a[i].w, ssq[i]);
                        a[i] *= nw;
                    }
                }
                if constexpr (PROBE == 2 || PROBE == 3) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[0][j] += w[j];
                    if constexpr (PROBE == 3) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][0] += a[i];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], w[j][e], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    // the KW partial tiles meet in LDS
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + j) * 64 + lane) * 4]) = acc[i][j];
    float* redss = red + KW * MT * NT * 256;                          // [KW][MT][16]
    if constexpr (RMS) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fg == 0) redss[(wave * MT + i) * 16 + fr] = v;
        }
    }
    __syncthreads();
    // every thread sums and stores its share: unit u = (row tile i, column tile j [pair for SwiGLU], lane slot l)
    const int jt = w13 ? NT / 2 : NT;
    const int units = MT * jt * 64;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
        const int u = tid + k * 64 * KW;
        if (u >= units) break;
        const int l = u & 63, ij = u >> 6;
        const int i = ij / jt, jj = ij - i * jt;
        const int col = l & 15, rq = (l >> 4) * 4;
        auto tile_sum = [&](int j) {
            f32x4 s = *reinterpret_cast<const f32x4*>(&red[((i * NT + j) * 64 + l) * 4]);
#pragma unroll 4
            for (int w = 1; w < KW; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * (MT * NT) + i * NT + j) * 64 + l) * 4]);
            return s;
        };
        f32x4 inv4 = (f32x4){1.f, 1.f, 1.f, 1.f};
        if constexpr (RMS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tot = 0.f;
#pragma unroll 4
                for (int w = 0; w < KW; ++w) tot += redss[(w * MT + i) * 16 + rq + r];
                inv4[r] = 1.f / sqrtf(tot / (float)Kt + g.rms_eps);
            }
        }
        if (w13) {
            const f32x4 ga = tile_sum(2 * jj) * inv4, up = tile_sum(2 * jj + 1) * inv4;
            const int n = n0 + jj * 32 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + rq + r;
                if (m >= g.M || n >= g.N) continue;
                const int b = m / g.T, tt = m - b * g.T;
                if (tt >= g.skip_lo && tt < g.skip_hi) continue;
                float* crow = g.C + (long)b * g.c_bstride + g.c_off + (long)tt * g.ldc;
                crow[(n0 >> 1) + jj * 16 + col] = silu_s(ga[r]) * up[r];
            }
            continue;
        }
        const f32x4 s = tile_sum(jj) * inv4;
        const int n = n0 + jj * 16 + col;
        if (n >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + i * 16 + rq + r;
            if (m >= g.M) continue;
            const int b = m / g.T, tt = m - b * g.T;
            if (tt >= g.skip_lo && tt < g.skip_hi) continue;
            float* cp = g.C + (long)b * g.c_bstride + g.c_off + (long)tt * g.ldc + n;
            float v = s[r] + e_bias[k];
            if (g.act == ACT_GELU) v = gelu_s(v);
            else if (g.act == ACT_LOGCLAMP) v = __logf(fmaxf(v, 1e-5f));
            v *= e_gamma[k];
            v += e_res[k][r];
            v *= g.scale;
            if (g.accumulate) v += *cp;
            *cp = v;
        }
    }
}

template <int MT, int NT, int KW, int KB, int AOP, bool WP, int PROBE, bool LOOP>
int launch_stream_kl(const ConvGemm& g, const float* Wsrc, hipStream_t st) {
    StreamArgs a;
    a.A = g.A; a.a_bstride = g.a_bstride; a.a_off = g.a_off; a.lda = g.lda; a.T = g.T; a.M = g.M; a.stride = g.stride; a.dil = g.dil;
    a.taps = g.taps; a.Cin = g.Cin; a.W = Wsrc; a.N = g.N; a.nkb = g.taps * g.Cin / 16;
    a.bias = g.bias; a.gamma = g.gamma; a.res = g.res; a.r_bstride = g.r_bstride; a.r_off = g.r_off; a.ldr = g.ldr;
    a.C = g.C; a.c_bstride = g.c_bstride; a.c_off = g.c_off; a.ldc = g.ldc; a.scale = g.scale; a.skip_lo = g.skip_lo; a.skip_hi = g.skip_hi;
    a.act = g.act; a.accumulate = g.accumulate; a.w13 = g.w13; a.rms_w = g.rms_w; a.rms_eps = g.rms_eps;
    a.m_tiles = (g.M + 16 * MT - 1) / (16 * MT); a.n_tiles = (g.N + 16 * NT - 1) / (16 * NT);
    const size_t smem = ((size_t)KW * MT * NT * 256 + (AOP == 2 ? KW * MT * 16 : 0)) * sizeof(float);
    static DeviceOnce attr;
    if (attr.needed() && smem > 48 * 1024) {
        SVA_HIP(hipFuncSetAttribute((const void*)stream_gemm_kernel<MT, NT, KW, KB, AOP, WP, PROBE, LOOP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr.done();
    }
    hipLaunchKernelGGL((stream_gemm_kernel<MT, NT, KW, KB, AOP, WP, PROBE, LOOP>), dim3(a.m_tiles * a.n_tiles), dim3(64 * KW), smem, st, a);
    return 0;
}

// registers in flight: 4 (MT + NT [+ 1 with the fused RMSNorm]) KB per lane -- at most ~200 load registers per wave (a 16-wave workgroup's
// waves own 128 registers; the looping form keeps a second copy of the block it is consuming)
template <int MT, int NT, int KW, int AOP, bool LOOP>
constexpr int kb_for() {
    constexpr int per = 4 * (MT + NT + (AOP == 2 ? 1 : 0));
    constexpr int budget = KW == 16 ? (LOOP ? 44 : 64) : (LOOP ? 136 : 200);
    return budget / per > 24 ? 24 : (budget / per < 2 ? 2 : budget / per);
}

template <int MT, int NT, int KW, int AOP, bool WP, int PROBE>
int launch_stream_k(const ConvGemm& g, const float* Wsrc, hipStream_t st) {
    const int nkb = g.taps * g.Cin / 16, need = (nkb + KW - 1) / KW;
    constexpr int KB0 = kb_for<MT, NT, KW, AOP, false>(), KB1 = kb_for<MT, NT, KW, AOP, true>();
#define SVA_KB(KBv) \
    if constexpr (KBv <= KB0) { if (need <= KBv) return launch_stream_kl<MT, NT, KW, KBv, AOP, WP, PROBE, false>(g, Wsrc, st); }
    SVA_KB(1) SVA_KB(2) SVA_KB(3) SVA_KB(4) SVA_KB(6) SVA_KB(8) SVA_KB(12) SVA_KB(16) SVA_KB(24)
#undef SVA_KB
    if constexpr (PROBE == 0) return launch_stream_kl<MT, NT, KW, KB1, AOP, WP, 0, true>(g, Wsrc, st);
    set_error("stream_gemm: probes are single-round only");
    return -1;
}

template <int MT, int NT, int KW, bool WP>
int launch_stream_aop(const ConvGemm& g, const float* Wsrc, int probe, hipStream_t st) {
    if constexpr (WP) {
        if (probe == 1) return launch_stream_k<MT, NT, KW, 0, WP, 1>(g, Wsrc, st);
        if (probe == 2) return launch_stream_k<MT, NT, KW, 0, WP, 2>(g, Wsrc, st);
        if (probe == 3) return launch_stream_k<MT, NT, KW, 0, WP, 3>(g, Wsrc, st);
    }
    if (probe) { set_error("stream_gemm: probes run on the packed weights"); return -1; }
    if (g.rms_w) return launch_stream_k<MT, NT, KW, 2, WP, 0>(g, Wsrc, st);
    if (g.a_silu) return launch_stream_k<MT, NT, KW, 1, WP, 0>(g, Wsrc, st);
    return launch_stream_k<MT, NT, KW, 0, WP, 0>(g, Wsrc, st);
}

template <int MT, int NT, int KW>
int launch_stream_w(const ConvGemm& g, const float* Wsrc, int wmode, int probe, hipStream_t st) {
    return (wmode & 2) ? launch_stream_aop<MT, NT, KW, true>(g, Wsrc, probe, st) : launch_stream_aop<MT, NT, KW, false>(g, Wsrc, probe, st);
}

}  // namespace

bool stream_gemm_supported(const ConvGemm& g) {
    return g.Cin % 16 == 0 && g.lda % 4 == 0 && g.a_off % 4 == 0 && g.a_bstride % 4 == 0 && !g.dw_wT && !g.Ap && !g.Cp && (!g.w13 || g.N % 32 == 0) &&
           (!g.rms_w || (g.taps == 1 && !g.a_silu));
}

// mt in {1, 2, 4}, nt in {1, 2}, kw in {4, 8, 16}; wmode bit 1 = Wsrc is the fragment-major packing
int launch_stream_gemm(const ConvGemm& g, const float* Wsrc, int mt, int nt, int kw, int wmode, int probe, hipStream_t st) {
    SVA_CHECK(stream_gemm_supported(g), "stream_gemm: unsupported problem");
    SVA_CHECK(!(g.w13 && nt != 2), "stream_gemm: SwiGLU needs column-tile pairs (nt = 2)");
#define SVA_SG(MTv, NTv, KWv) \
    if (mt == MTv && nt == NTv && kw == KWv) return launch_stream_w<MTv, NTv, KWv>(g, Wsrc, wmode, probe, st);
    SVA_SG(1, 1, 4) SVA_SG(1, 1, 8) SVA_SG(1, 1, 16)
    SVA_SG(2, 1, 4) SVA_SG(2, 1, 8) SVA_SG(2, 1, 16)
    SVA_SG(4, 1, 4) SVA_SG(4, 1, 8)
    SVA_SG(1, 2, 4) SVA_SG(1, 2, 8) SVA_SG(1, 2, 16)
    SVA_SG(2, 2, 4) SVA_SG(2, 2, 8)
    SVA_SG(4, 2, 4) SVA_SG(4, 2, 8)
#undef SVA_SG
    set_error("stream_gemm: bad configuration");
    return -1;
}

}  // namespace sva
