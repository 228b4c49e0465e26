// Linear layers of the batched AR decode with fp16 WEIGHTS on the f16 matrix pipes (ar_dtype = 1, more than 6 streams):
//   C[m, n] = epi( sum_k A[m, k] * Wh[n, k] ),  A fp32 activations, Wh fp16 [N][K] (the reference decodes under torch.autocast(fp16) with
//   fp16 weights: evaluations/infer_arvc.py:55-59, 493; modules/dual_ar_stream.py:1168-1219), fp32 accumulation in
//   v_mfma_f32_16x16x32_f16.
// The round-2 engine ran these GEMMs as f32 MFMAs on an fp32 COPY of the rounded weights: twice the bytes of a weight-streaming
// problem (M = 2 x streams rows against 0.6-3.5 M weight elements per matrix).  Here the fp16 weights are streamed once, straight
// from global memory into MFMA operands (no LDS in the K loop: the K axis is split over the KW waves of a workgroup, lanes load
// 16-byte fragments = 8 halves of a weight row, partial tiles are reduced through LDS once -- the structure of skinny_gemm_kernel).
// Activations keep fp32 precision: each fp32 value is split exactly into hi + lo halves and both part products are accumulated
// (fp16 x fp16 products are exact in fp32), so the result equals the persistent kernel's fp32 x fp16 FMAs to fp32 rounding and the
// two decode paths of an ar_dtype = 1 engine agree (the MFMA count doubles; the kernel is bound by the weight stream).
// Prologue: RMSNorm of the A rows folded in (weight into the operand, 1 / rms applied to the accumulators).  Epilogues: residual
// add, SwiGLU over (gate, up) column-tile pairs, bias.
#include "sva_common.h"

namespace sva {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

// eight fp32 -> (hi, lo) fp16 fragments with hi + lo == x to 2^-22 relative (RNE conversions, exact fp32 residuals)
__device__ __forceinline__ void split8(const float4& a, const float4& b, f16x8& hi, f16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // hi is made opaque before lo is derived from it: with the RMSNorm product feeding v, hipcc otherwise contracts the two uses
        // differently -- the hi operand of the MFMA from the fp32-rounded product (v_cvt_pk_f16_f32), lo against v_fma_mixlo_f16 of
        // the EXACT product -- and the two roundings of hi disagree by one fp16 ulp about once per 2^13 elements (measured: single
        // elements 2^-12 off).  With one hi, lo = v - hi is consistent whichever way v is evaluated.
        // the value itself is pinned first: where it is a product (RMSNorm weight x activation) some unrolled copies otherwise round it straight to fp16
        // (v_fma_mixlo_f16: one rounding) and others through fp32 (v_mul + v_cvt: two), and a row's hi / lo pair depended on the row tile it sat in
        float xv = v[i];
        asm("" : "+v"(xv));
        _Float16 h = (_Float16)xv;
        asm("" : "+v"(h));
        hi[i] = h;
        lo[i] = (_Float16)(xv - (float)h);
    }
}

template <int MT, int NT, int KW, bool RMS>
__global__ __launch_bounds__(64 * KW) void f16w_gemm_kernel(const ConvGemm g, const _Float16* __restrict__ Wh) {
    constexpr int D = 2;                                   // K blocks in flight per wave
    constexpr int LPS = NT + 2 * MT + (RMS ? 2 : 0);      // 16-byte loads per slot (issue())
    extern __shared__ __attribute__((aligned(16))) float red[];      // [KW][MT*NT][64][4] (+ [KW][MT][16] row sums of squares)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * (16 * NT);
    const int m_base = blockIdx.y * (16 * MT);
    const int fr = lane & 15, fk = lane >> 4;
    const int K = g.Cin;
    const _Float16* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + j * 16 + fr;
        if (n > g.N - 1) n = g.N - 1;
        wp[j] = Wh + (long)n * K + 8 * fk;
    }
    const float* ap[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m_base + i * 16 + fr;
        if (m > g.M - 1) m = g.M - 1;
        const int b = m / g.T, t = m - b * g.T;
        ap[i] = g.A + (long)b * g.a_bstride + g.a_off + (long)t * g.lda + 8 * fk;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = K / 32;
    const int my_n = nk > wave ? (nk - wave + KW - 1) / KW : 0;          // K blocks owned by this wave (kb = wave + q KW)
    const int last_kb = my_n > 0 ? wave + (my_n - 1) * KW : 0;
    uint4 wv[D][NT];
    float4 av[D][MT][2], nv[D][2];
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ssq[i] = 0.f;
    // no branch around any load (a conditional load makes hipcc drain vmcnt(0) at the join): out-of-range blocks re-load the wave's
    // last valid block and are masked
    auto issue = [&](int d, int kb) {
        kb = kb < nk ? kb : last_kb;
        const long off = (long)kb * 32;
#pragma unroll
        for (int j = 0; j < NT; ++j) wv[d][j] = *reinterpret_cast<const uint4*>(wp[j] + off);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            av[d][i][0] = *reinterpret_cast<const float4*>(ap[i] + off);
            av[d][i][1] = *reinterpret_cast<const float4*>(ap[i] + off + 4);
        }
        if constexpr (RMS) {
            nv[d][0] = *reinterpret_cast<const float4*>(g.rms_w + off + 8 * fk);
            nv[d][1] = *reinterpret_cast<const float4*>(g.rms_w + off + 8 * fk + 4);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, wave + d * KW);
    for (int it = 0; it < my_n; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // Explicit, counted wait for slot d's loads (the other slot's LPS loads, issued later, may stay in flight; vector memory
            // returns in order); the memory clobber pins the loads on their side of the wait.  History: the <2, 2, 4, RMS> instantiation
            // computed row sums of squares 1-4 % low, differently per launch (found by the M = 24 / 32 SwiGLU unit tests); with these
            // waits that build was clean, but an -amdgpu-waitcnt-forcezero build then failed the same way in <4, 2, 4, RMS>.  What
            // The defect follows the packed-fp32 code hipcc makes of the sum-of-squares tree (see below), not the waits; the
            // opaque copy below additionally gives both uses of a fragment one set of registers.  tools/f16w_stress.py must pass
            // under both builds (tools/waitcnt_audit.sh).
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (D - 1)) : "memory");
            const int kb = wave + (it + d) * KW;
            const float keep = kb < nk ? 1.f : 0.f;
            f16x8 w[NT], ah[MT], al[MT];
#pragma unroll
            for (int j = 0; j < NT; ++j) w[j] = __builtin_bit_cast(f16x8, wv[d][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float4 a0 = av[d][i][0], a1 = av[d][i][1];
                // one physical copy of the landed values feeds both the row norms and the split (hipcc otherwise re-loads parts of
                // the fragment for one of the two uses)
                asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z), "+v"(a0.w), "+v"(a1.x), "+v"(a1.y), "+v"(a1.z), "+v"(a1.w));
                a0.x *= keep; a0.y *= keep; a0.z *= keep; a0.w *= keep;
                a1.x *= keep; a1.y *= keep; a1.z *= keep; a1.w *= keep;
                if constexpr (RMS) {
                    // a plain FMA chain: the pairwise tree this replaced is what hipcc turned into the v_pk_mul / v_pk_fma / op_sel
                    // sequence that mis-summed (same source, forcezero build: tree 8 of 12 cases wrong, chain 0 of 12)
                    float t_ = ssq[i];
                    t_ = __builtin_fmaf(a0.x, a0.x, t_); t_ = __builtin_fmaf(a0.y, a0.y, t_); t_ = __builtin_fmaf(a0.z, a0.z, t_); t_ = __builtin_fmaf(a0.w, a0.w, t_);
                    t_ = __builtin_fmaf(a1.x, a1.x, t_); t_ = __builtin_fmaf(a1.y, a1.y, t_); t_ = __builtin_fmaf(a1.z, a1.z, t_); t_ = __builtin_fmaf(a1.w, a1.w, t_);
                    ssq[i] = t_;
                    a0.x *= nv[d][0].x; a0.y *= nv[d][0].y; a0.z *= nv[d][0].z; a0.w *= nv[d][0].w;
                    a1.x *= nv[d][1].x; a1.y *= nv[d][1].y; a1.z *= nv[d][1].z; a1.w *= nv[d][1].w;
                }
                split8(a0, a1, ah[i], al[i]);
            }
            issue(d, wave + (it + d + D) * KW);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], w[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], w[j], acc[i][j], 0, 0, 0);
        }
    }
    // cross-wave reduction of the K slices
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
            if (fk == 0) redss[(wave * MT + i) * 16 + fr] = v;
        }
    }
    __syncthreads();
    const int col = lane & 15, rq = (lane >> 4) * 4;       // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    for (int i = wave; i < MT; i += KW) {
        f32x4 t[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 s = *reinterpret_cast<const f32x4*>(&red[((i * NT + j) * 64 + lane) * 4]);
#pragma unroll
            for (int w = 1; w < KW; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * (MT * NT) + i * NT + j) * 64 + lane) * 4]);
            t[j] = s;
        }
        if constexpr (RMS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < KW; ++w) tot += redss[(w * MT + i) * 16 + rq + r];
                const float inv = 1.f / sqrtf(tot / (float)K + g.rms_eps);
#pragma unroll
                for (int j = 0; j < NT; ++j) t[j][r] *= inv;

            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + i * 16 + rq + r;
            if (m >= g.M) continue;
            const int b = m / g.T, tt = m - b * g.T;
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
                    if (rrow) v += rrow[n];
                    crow[n] = v;
                }
            }
        }
    }
}

template <int MT, int NT, int KW>
int launch_cfg(const ConvGemm& g, const _Float16* Wh, hipStream_t st) {
    const dim3 grid((g.N + 16 * NT - 1) / (16 * NT), (g.M + 16 * MT - 1) / (16 * MT));
    const size_t smem = ((size_t)KW * MT * NT * 256 + KW * MT * 16) * sizeof(float);
    static_assert((size_t)KW * MT * NT * 256 * 4 + KW * MT * 16 * 4 <= 64 * 1024, "partial tiles fit the default LDS limit");
    if (g.rms_w) hipLaunchKernelGGL((f16w_gemm_kernel<MT, NT, KW, true>), grid, dim3(64 * KW), smem, st, g, Wh);
    else hipLaunchKernelGGL((f16w_gemm_kernel<MT, NT, KW, false>), grid, dim3(64 * KW), smem, st, g, Wh);
    SVA_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// This kernel carries compiler-specific workarounds (opaque copies of the A fragments, a plain FMA chain for the row norms: see the header
// and DESIGN.md section 4) that were validated -- tools/f16w_stress.py under the normal and the -amdgpu-waitcnt-forcezero build, inside the
// GPU tests -- with the compiler of ROCm 7.2 (AMD clang 22.0, HIP 7.2).  Built with anything else the dispatcher does not pick it by
// default: the fp32 copy of the rounded weights runs instead until the stress sweep has been repeated and this check updated.
bool f16w_gemm_validated_compiler() {
#if defined(__clang_major__) && __clang_major__ == 22 && defined(HIP_VERSION_MAJOR) && HIP_VERSION_MAJOR == 7 && HIP_VERSION_MINOR == 2
    return true;
#else
    return false;
#endif
}

// plain linear layers: one tap, unit stride, K a multiple of 32 with 16-byte aligned rows, the epilogues of the AR chain only
bool f16w_gemm_supported(const ConvGemm& g) {
    return g.Wh && g.taps == 1 && g.stride == 1 && g.Cin % 32 == 0 && g.lda % 4 == 0 && g.a_off % 4 == 0 && g.a_bstride % 4 == 0 && !g.a_silu && !g.dw_wT &&
           !g.gamma && g.act == ACT_NONE && !g.accumulate && g.scale == 1.f && g.skip_hi <= g.skip_lo && (!g.w13 || g.N % 32 == 0) &&
           g.M >= 1 && g.M <= 1024;
}

int launch_f16w_gemm(const ConvGemm& g, hipStream_t st) {
    SVA_CHECK(f16w_gemm_supported(g), "f16w_gemm: unsupported problem");
    const _Float16* Wh = reinterpret_cast<const _Float16*>(g.Wh);
    const long K = g.Cin;
    // rows: up to 64 per workgroup (the weight stream is read once per 64 rows); columns: (gate, up) pairs for SwiGLU, else one or
    // two 16-column tiles by how many workgroups that leaves; K over 8 waves when it is long
    const int mt = g.M <= 16 ? 1 : g.M <= 32 ? 2 : 4;
    const bool nt2 = g.w13 || (g.N % 32 == 0 && (long)(g.N / 16) * ((g.M + 16 * mt - 1) / (16 * mt)) >= 512);
    const bool kw8 = K >= 1024;
#define F16W_GO(MT_, NT_, KW_) return launch_cfg<MT_, NT_, KW_>(g, Wh, st)
    if (mt == 1) { if (nt2) { if (kw8) F16W_GO(1, 2, 8); F16W_GO(1, 2, 4); } if (kw8) F16W_GO(1, 1, 8); F16W_GO(1, 1, 4); }
    if (mt == 2) { if (nt2) { if (kw8) F16W_GO(2, 2, 8); F16W_GO(2, 2, 4); } if (kw8) F16W_GO(2, 1, 8); F16W_GO(2, 1, 4); }
    if (nt2) F16W_GO(4, 2, 4);          // 8 waves of 4 x 2 partial tiles would not fit 64 KB of LDS
    if (kw8) F16W_GO(4, 1, 8);
    F16W_GO(4, 1, 4);
#undef F16W_GO
}

}  // namespace sva
