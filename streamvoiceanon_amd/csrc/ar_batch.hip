// Persistent batched decode kernel of the dual AR: ONE launch per frame for all B streams of a batch.
//
// decode_one_token_ar (modules/dual_ar_stream.py:1168-1219) for B streams is 12 slow layers on 2 B rows, the semantic head and
// 8 x (4 fast layers on B rows + codebook head + nucleus sample): ~240 dependent matrix products that each move a few MB of
// weights.  As separate launches (engine.hip: ar_layers_pass / ar_frame_tail, ~265 kernels of 5-25 us) the frame costs 3.4-3.8 ms
// at 12-64 streams although the weight stream (0.26 GB fp16 / 0.52 GB fp32) and the matrix work are worth a fraction of that.
// Here G workgroups stay resident for the whole frame and run a STATIC schedule of phases; phase p of a layer is cut into
// independent units that workgroup w takes round-robin (unit w, w + G, ...):
//   * linear phases (wqkv, wo, w1|w3, w2, heads): unit = (row tile of 16 MT rows, 16 output columns).  The K axis is split over
//     the 4 waves of the workgroup; lanes stream 16-byte weight fragments and 32-byte activation fragments straight from global
//     memory into MFMA operands (no LDS in the K loop), partial tiles are reduced through LDS once -- the structure of
//     gemm_f16w.hip.  fp16 weights: v_mfma_f32_16x16x32_f16 with the fp32 activations split exactly into hi + lo halves;
//     fp32 weights: v_mfma_f32_16x16x4_f32.  RMSNorm is folded in (weight into the operand, 1 / rms onto the accumulators);
//     RoPE + KV write, residual add and SwiGLU are epilogues.
//   * slow attention: unit = (stream, head), both new rows against the cached keys (4 waves split the keys, merged in LDS);
//     fast attention (<= 8 codebook positions): unit = stream; samplers and bookkeeping: unit = stream.
//   * hand-off between phases (cdna_hip_programming.md Guideline 16, form R1 with sc1 loads in place of the acquire): a unit
//     stores its output tile write-through (sc1), every wave drains vmcnt, one lane stores the tile's flag = epoch of the phase;
//     a consuming wave polls exactly the flags of the tiles its K range reads (relaxed agent-scope loads), then reads the
//     tile with sc1 loads.  Nothing depends on workgroup placement or dispatch order; all waits are bounded (timeout -> *fail,
//     the kernel runs to its end with garbage instead of hanging).  Every consumer unit's four waves together wait for ALL
//     column tiles of its row tile, so a buffer is rewritten only after every reader of the previous contents has finished
//     (the next writer's inputs depend on those readers' outputs).
// Weight rows of a unit are requested BEFORE its wave waits for the input flags, so the weight stream's latency overlaps the
// hand-off -- the one thing a launch boundary cannot do.
#include "ar_batch.h"
#include "device_util.h"
#include "sva_common.h"

#include "ar_device.h"

namespace sva {
namespace {

using namespace ardev;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int SPIN_LIMIT = 1 << 15;                // polls before a wait gives up (tens of ms); a healthy hand-off takes a handful
constexpr int QT = 3 * D / 16, XT = D / 16, GT = I / 16;       // column tiles of the qkv (144), x / att (48) and SwiGLU (144) buffers
constexpr int LOGT = 64, SEMT = 512;                            // flag words per row tile of the codebook / semantic logits

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wave: wait until the n (<= 64) consecutive flag words at f have reached epoch `want`
__device__ __forceinline__ void wait_flags(const unsigned* f, int n, unsigned want, int* fail, int code) {
    const int lane = threadIdx.x & 63;
    bool ok = lane >= n;
    int spins = 0;
    while (true) {
        if (!ok) ok = (int)(__hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0;
        if (__all(ok)) break;
        ++spins;
        if ((spins & 127) == 0 && *reinterpret_cast<volatile int*>(fail)) break;          // an earlier wait timed out somewhere: run through
        if (spins > SPIN_LIMIT) { if (lane == 0) *fail = code; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// end of a unit: every wave's write-through stores have completed, then ONE lane publishes the tile's flag
__device__ __forceinline__ void publish(unsigned* flag, unsigned ep) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A operand of a linear phase: rows of an fp32 buffer written earlier in this launch (read with sc1 loads)
struct ASrc {
    __amdgpu_buffer_rsrc_t rs;
    int row_stride, off;           // bytes
};
__device__ __forceinline__ ASrc make_asrc(const float* base, long floats, int row_stride_floats, int off_floats) {
    ASrc a;
    a.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(floats * 4), 0x00020000);
    a.row_stride = row_stride_floats * 4;
    a.off = off_floats * 4;
    return a;
}
__device__ __forceinline__ float4 ld_a16(const ASrc& a, int voff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a.rs, voff, 0, 16));      // aux 16 = sc1
}

// eight weights of one row per lane
template <typename WT> struct WReg;
template <> struct WReg<__half> {
    uint4 v;
    __device__ __forceinline__ void load(const __half* p) { v = *reinterpret_cast<const uint4*>(p); }
};
template <> struct WReg<float> {
    float4 v0, v1;
    __device__ __forceinline__ void load(const float* p) {
        v0 = *reinterpret_cast<const float4*>(p);
        v1 = *reinterpret_cast<const float4*>(p + 4);
    }
};

// eight fp32 -> (hi, lo) fp16 fragments with hi + lo == x to 2^-22 relative (gemm_f16w.hip: split8, incl. the opaque hi)
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 h = (_Float16)v[i];
        asm("" : "+v"(h));
        hi[i] = h;
        lo[i] = (_Float16)(v[i] - (float)h);
    }
}

// operands of one 32-wide K block of one wave: the landed A fragments (x RMSNorm weight) as MFMA operands -- fp16 weights: exact
// hi + lo halves; fp32 weights: the eight values -- and the row sums of squares of the raw A values
template <typename WT, int MT> struct AOps;
template <int MT> struct AOps<__half, MT> { f16x8 hi[MT], lo[MT]; };
template <int MT> struct AOps<float, MT> { float x[MT][8]; };

template <typename WT, int MT, bool RMS>
__device__ __forceinline__ void prep_block(const float4 (&av)[MT][2], const float4 (&nv)[2], AOps<WT, MT>& o, float (&ssq)[MT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float4 a0 = av[i][0], a1 = av[i][1];
        // one physical copy of the landed values feeds both the row norms and the operands (gemm_f16w.hip)
        asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z), "+v"(a0.w), "+v"(a1.x), "+v"(a1.y), "+v"(a1.z), "+v"(a1.w));
        if constexpr (RMS) {
            float t_ = ssq[i];          // a plain FMA chain (gemm_f16w.hip: the pairwise tree was mis-compiled into packed math)
            t_ = __builtin_fmaf(a0.x, a0.x, t_); t_ = __builtin_fmaf(a0.y, a0.y, t_); t_ = __builtin_fmaf(a0.z, a0.z, t_); t_ = __builtin_fmaf(a0.w, a0.w, t_);
            t_ = __builtin_fmaf(a1.x, a1.x, t_); t_ = __builtin_fmaf(a1.y, a1.y, t_); t_ = __builtin_fmaf(a1.z, a1.z, t_); t_ = __builtin_fmaf(a1.w, a1.w, t_);
            ssq[i] = t_;
            a0.x *= nv[0].x; a0.y *= nv[0].y; a0.z *= nv[0].z; a0.w *= nv[0].w;
            a1.x *= nv[1].x; a1.y *= nv[1].y; a1.z *= nv[1].z; a1.w *= nv[1].w;
        }
        const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        if constexpr (std::is_same<WT, __half>::value) split8(x, o.hi[i], o.lo[i]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o.x[i][e] = x[e];
        }
    }
}

template <typename WT, int MT, int NT>
__device__ __forceinline__ void mma_block(const AOps<WT, MT>& o, const WReg<WT> (&w)[NT], f32x4 (&acc)[MT][NT]) {
    if constexpr (std::is_same<WT, __half>::value) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.lo[i], __builtin_bit_cast(f16x8, w[j].v), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.hi[i], __builtin_bit_cast(f16x8, w[j].v), acc[i][j], 0, 0, 0);
    } else {
        // v_mfma_f32_16x16x4_f32 sums k = lane >> 4; element e of the lane's eight stands for k index (lane >> 4, e) on BOTH
        // operands, so eight MFMAs cover the 32-wide block
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float wf[8] = {w[j].v0.x, w[j].v0.y, w[j].v0.z, w[j].v0.w, w[j].v1.x, w[j].v1.y, w[j].v1.z, w[j].v1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.x[i][e], wf[e], acc[i][j], 0, 0, 0);
        }
    }
}

// Partial tiles of C[m0 .. m0 + 16 MT) x [n0 .. n0 + 16 NT) = A[rows] (x RMSNorm weight) . W[cols]^T, K split over the 4 waves; on return
// the per-wave partial tiles sit in `red` ([4][MT * NT][64][4] floats, then [4][MT][16] row sums of squares) behind a barrier.
// waitf() is called by every wave after its first weight fragments are requested and before its first activation load.
template <typename WT, int MT, int NT, int K, bool RMS, typename WaitF>
__device__ __forceinline__ void linear_tile(const ASrc& A, int m0, int M, const WT* __restrict__ W, int n0, int N, const float* __restrict__ rms_w,
                                            WaitF&& waitf, float* red) {
    constexpr int NKW = K / 128;                   // 32-wide K blocks per wave
    constexpr int DEP = MT >= 4 ? 2 : 3;           // K blocks in flight per wave
    static_assert(NKW % DEP == 0 && NKW >= 2 * DEP, "K blocks per wave must be a multiple of the pipeline depth");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
    const int kbase = wave * (K / 4) + 8 * fk;
    const WT* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + 16 * j + fr;
        if (n > N - 1) n = N - 1;
        wp[j] = W + (long)n * K + kbase;
    }
    int aoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m0 + 16 * i + fr;
        if (m > M - 1) m = M - 1;
        aoff[i] = A.off + m * A.row_stride + kbase * 4;
    }
    const float* np = rms_w + kbase;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ssq[i] = 0.f;
    WReg<WT> wv[DEP][NT];
    float4 av[DEP][MT][2], nv[DEP][2];
    auto issue_w = [&](int d, int it) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wv[d][j].load(wp[j] + it * 32);
        if constexpr (RMS) {
            nv[d][0] = *reinterpret_cast<const float4*>(np + it * 32);
            nv[d][1] = *reinterpret_cast<const float4*>(np + it * 32 + 4);
        }
    };
    auto issue_a = [&](int d, int it) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            av[d][i][0] = ld_a16(A, aoff[i] + it * 128);
            av[d][i][1] = ld_a16(A, aoff[i] + it * 128 + 16);
        }
    };
#pragma unroll
    for (int d = 0; d < DEP; ++d) issue_w(d, d);
    waitf();
#pragma unroll
    for (int d = 0; d < DEP; ++d) issue_a(d, d);
    // steady state: turn slot d into operands, refill it DEP blocks ahead (no branch around any load), multiply; the last DEP
    // blocks only consume
#pragma unroll 1
    for (int it0 = 0; it0 < NKW - DEP; it0 += DEP) {
#pragma unroll
        for (int d = 0; d < DEP; ++d) {
            AOps<WT, MT> o;
            prep_block<WT, MT, RMS>(av[d], nv[d], o, ssq);
            WReg<WT> wc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wc[j] = wv[d][j];
            if constexpr (std::is_same<WT, __half>::value) {
                issue_w(d, it0 + d + DEP);
                issue_a(d, it0 + d + DEP);
                mma_block<WT, MT, NT>(o, wc, acc);
            } else {            // (fp32 operands stay live through the MFMAs: refill afterwards)
                mma_block<WT, MT, NT>(o, wc, acc);
                issue_w(d, it0 + d + DEP);
                issue_a(d, it0 + d + DEP);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DEP; ++d) {
        AOps<WT, MT> o;
        prep_block<WT, MT, RMS>(av[d], nv[d], o, ssq);
        mma_block<WT, MT, NT>(o, wv[d], acc);
    }
    // cross-wave reduction of the K slices
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + j) * 64 + lane) * 4]) = acc[i][j];
    if constexpr (RMS) {
        float* redss = red + 4 * MT * NT * 256;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fk == 0) redss[(wave * MT + i) * 16 + fr] = v;
        }
    }
    __syncthreads();
}

// sum of the four waves' partials of row sub-tile i: t[j][r] = element (row 4 (lane >> 4) + r, column lane & 15) of column tile j
template <int MT, int NT>
__device__ __forceinline__ void tile_sum(const float* red, int i, int lane, f32x4 (&t)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x4 s = *reinterpret_cast<const f32x4*>(&red[((i * NT + j) * 64 + lane) * 4]);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * (MT * NT) + i * NT + j) * 64 + lane) * 4]);
        t[j] = s;
    }
}
template <int MT, int NT>
__device__ __forceinline__ float row_inv(const float* red, int i, int row16, int K, float eps) {
    const float* redss = red + 4 * MT * NT * 256;
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) tot += redss[(w * MT + i) * 16 + row16];
    return 1.f / sqrtf(tot / (float)K + eps);
}

constexpr size_t red_floats(int MT, int NT) { return (size_t)4 * MT * NT * 256 + 4 * MT * 16; }

// wo / w2 epilogue: x[m][n] = res[m][n] + acc   (res rows through an ASrc-like (stride, offset) view of a float buffer)
template <int MT>
__device__ __forceinline__ void epi_residual(const float* red, int m0, int M, int n0, const float* res, int res_stride, int res_off, float* out,
                                             float* tap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, rq = (lane >> 4) * 4;
    for (int i = wave; i < MT; i += 4) {
        f32x4 t[1];
        tile_sum<MT, 1>(red, i, lane, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + rq + r;
            if (m >= M) continue;
            const int n = n0 + col;
            const float rv = ld_sc1(res + (long)m * res_stride + res_off + n);
            if (tap) tap[(long)m * D + n] = rv;
            st_sc1(out + (long)m * D + n, rv + t[0][r]);
        }
    }
}

#define AB_MARK() do { if (a.dbg && wg == 0 && tid == 0 && nmark < 1000) a.dbg[nmark] = wall_clock64(); ++nmark; } while (0)

template <typename WT, typename KVT, int MTS, int MTF>
__global__ __launch_bounds__(256, 2) void ar_batch_kernel(const ArBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* red = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, G = a.G;
    const int col = lane & 15, rq = (lane >> 4) * 4;
    const int B = a.B, M2 = 2 * B;
    const int TMS = (M2 + 16 * MTS - 1) / (16 * MTS), TMF = (B + 16 * MTF - 1) / (16 * MTF);
    const int V = a.codebook_size, VT = (V + 15) / 16, ST = (a.vocab + 15) / 16;
    __builtin_amdgcn_s_setprio(3);
    unsigned ep = *a.epoch;
    int nmark = 0;
    const int use_forced = *a.use_forced;
    const long SH = (long)a.S * 64;
    const ASrc AXS = make_asrc(a.xs, (long)M2 * D, D, 0), AATT = make_asrc(a.att, (long)M2 * D, D, 0), AG = make_asrc(a.g, (long)M2 * I, I, 0);
    const ASrc AHID = make_asrc(a.xs, (long)M2 * D, 2 * D, D);            // content-token rows of the slow residual stream
    const ASrc AXF = make_asrc(a.xf, (long)B * D, D, 0), AATTF = make_asrc(a.attf, (long)B * D, D, 0), AGF = make_asrc(a.gf, (long)B * I, I, 0);
    AB_MARK();

    // ======================================= slow AR: 12 layers on 2 B rows =======================================
    unsigned e_x = 0;           // epoch at which xs was last published (0: by the previous kernel)
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const ArLayerW& L = a.slow[l];
        KVT* const kvl = reinterpret_cast<KVT*>(a.kv_slow) + (long)l * a.kv_layer_stride;
        // ---- QKV: RMSNorm + wqkv + RoPE + KV write ----
        ++ep;
        for (int u = wg; u < TMS * QT; u += G) {
            const int mi = u / QT, nj = u - mi * QT, m0 = mi * 16 * MTS, n0 = nj * 16;
            linear_tile<WT, MTS, 1, D, true>(AXS, m0, M2, reinterpret_cast<const WT*>(L.wqkv), n0, 3 * D, L.attn_norm,
                                             [&] { if (l > 0) wait_flags(a.f_x + mi * XT + 12 * wave, 12, e_x, a.fail, 1); }, red);
            const int region = n0 / D, nn = n0 + col - region * D, h = nn >> 6, d = nn & 63;
            for (int i = wave; i < MTS; i += 4) {
                f32x4 t[1];
                tile_sum<MTS, 1>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r, mc = m < M2 ? m : M2 - 1;
                    const int s = mc >> 1, pos = a.last_pos[s] + 1 + (mc & 1);
                    float v = t[0][r] * row_inv<MTS, 1>(red, i, rq + r, D, 1e-5f);
                    const float pv = lane_xor_f<1>(v);
                    if (region < 2) {
                        const float c = a.rope_slow[((long)pos * 32 + (d >> 1)) * 2], sn = a.rope_slow[((long)pos * 32 + (d >> 1)) * 2 + 1];
                        v = (d & 1) ? v * c + pv * sn : v * c - pv * sn;
                    }
                    if (m < M2) {
                        st_sc1(a.qkv + (long)m * 3 * D + n0 + col, v);
                        if (region >= 1) st_kv<KVT>(kvl + (long)s * a.kv_slot_stride + ((long)(region - 1) * H + h) * SH + (long)pos * 64 + d, v);
                    }
                }
            }
            publish(a.f_qkv + mi * QT + nj, ep);
        }
        AB_MARK();
        // ---- ATT: (stream, head): both new rows against keys 0 .. p0 (+ 1) ----
        ++ep;
        for (int u = wg; u < B * H; u += G) {
            const int s = u / H, h = u - s * H;
            const int p0 = a.last_pos[s] + 1;                    // positions of the two new tokens: p0, p0 + 1
            const int grp = lane >> 4, li = lane & 15;
            const KVT* kc = kvl + (long)s * a.kv_slot_stride + (long)h * SH + li * 4;
            const KVT* vc = kc + (long)H * SH;
            const int lo = (int)((long)wave * p0 / 4), hi = (int)((long)(wave + 1) * p0 / 4);       // this wave's cached keys
            // cached K / V rows were written by earlier launches: request the first 8 keys of every 16-lane group before waiting for q
            float4 pkk[8], pvv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int t = lo + grp + 4 * i;
                if (t > hi - 1) t = hi - 1;
                if (t < 0) t = 0;
                pkk[i] = ld_kv4<KVT>(kc + (long)t * 64);
                pvv[i] = ld_kv4<KVT>(vc + (long)t * 64);
            }
            asm volatile("" ::: "memory");
            {   // q / k / v column tiles of head h of this stream's row tile
                const int mi = (2 * s) / (16 * MTS);
                const int reg = lane >> 2, q = lane & 3;
                const unsigned* f = a.f_qkv + mi * QT + (lane < 12 ? reg * XT + 4 * h + q : 0);
                bool ok = lane >= 12;
                int spins = 0;
                while (true) {
                    if (!ok) ok = (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ep + 1) >= 0;
                    if (__all(ok)) break;
                    ++spins;
                    if ((spins & 127) == 0 && *reinterpret_cast<volatile int*>(a.fail)) break;
                    if (spins > SPIN_LIMIT) { if (lane == 0) *a.fail = 2; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            }
            const float* qrow = a.qkv + (long)(2 * s) * 3 * D + h * 64 + li * 4;
            float4 q0, q1;
            q0.x = ld_sc1(qrow) * 0.125f; q0.y = ld_sc1(qrow + 1) * 0.125f; q0.z = ld_sc1(qrow + 2) * 0.125f; q0.w = ld_sc1(qrow + 3) * 0.125f;
            q1.x = ld_sc1(qrow + 3 * D) * 0.125f; q1.y = ld_sc1(qrow + 3 * D + 1) * 0.125f; q1.z = ld_sc1(qrow + 3 * D + 2) * 0.125f; q1.w = ld_sc1(qrow + 3 * D + 3) * 0.125f;
            float mr0 = -INFINITY, ls0 = 0.f, mr1 = -INFINITY, ls1 = 0.f;
            float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
            auto upd = [](float sc, const float4& vv, float& mr, float& ls, float4& o) {
                const float mn = fmaxf(mr, sc);
                const float corr = expf(mr - mn), p = expf(sc - mn);
                ls = ls * corr + p;
                o.x = o.x * corr + p * vv.x; o.y = o.y * corr + p * vv.y; o.z = o.z * corr + p * vv.z; o.w = o.w * corr + p * vv.w;
                mr = mn;
            };
            auto step2 = [&](const float4& kk, const float4& vv) {
                const float s0 = row16_sum(q0.x * kk.x + q0.y * kk.y + q0.z * kk.z + q0.w * kk.w);
                const float s1 = row16_sum(q1.x * kk.x + q1.y * kk.y + q1.z * kk.z + q1.w * kk.w);
                upd(s0, vv, mr0, ls0, o0);
                upd(s1, vv, mr1, ls1, o1);
            };
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (lo + grp + 4 * i < hi) step2(pkk[i], pvv[i]);
            int t = lo + grp + 32;
#pragma unroll 4
            for (; t < hi; t += 4) {
                const float4 kk = ld_kv4<KVT>(kc + (long)t * 64), vv = ld_kv4<KVT>(vc + (long)t * 64);
                step2(kk, vv);
            }
            if (wave == 3 && grp < 2) {
                // the two keys of this frame, from the freshly published rows: group 0 = key p0 (both rows), group 1 = key p0 + 1 (row 1 only)
                const float* kr = a.qkv + (long)(2 * s + grp) * 3 * D + D + h * 64 + li * 4;
                float4 kk, vv;
                kk.x = ld_sc1(kr); kk.y = ld_sc1(kr + 1); kk.z = ld_sc1(kr + 2); kk.w = ld_sc1(kr + 3);
                vv.x = ld_sc1(kr + D); vv.y = ld_sc1(kr + D + 1); vv.z = ld_sc1(kr + D + 2); vv.w = ld_sc1(kr + D + 3);
                const float s1 = row16_sum(q1.x * kk.x + q1.y * kk.y + q1.z * kk.z + q1.w * kk.w);
                upd(s1, vv, mr1, ls1, o1);
                if (grp == 0) {
                    const float s0 = row16_sum(q0.x * kk.x + q0.y * kk.y + q0.z * kk.z + q0.w * kk.w);
                    upd(s0, vv, mr0, ls0, o0);
                }
            }
            float* part = red;                              // [2][16][68]
            {
                float* p0_ = part + (wave * 4 + grp) * 68;
                float* p1_ = part + (16 + wave * 4 + grp) * 68;
                *reinterpret_cast<float4*>(p0_ + li * 4) = o0;
                *reinterpret_cast<float4*>(p1_ + li * 4) = o1;
                if (li == 0) { p0_[64] = mr0; p0_[65] = ls0; p1_[64] = mr1; p1_[65] = ls1; }
            }
            __syncthreads();
            if (tid < 128) {
                const int row = tid >> 6, dd = tid & 63;
                const float* pr = part + row * 16 * 68;
                float Mx = -INFINITY;
#pragma unroll
                for (int g2 = 0; g2 < 16; ++g2)
                    if (pr[g2 * 68 + 65] > 0.f) Mx = fmaxf(Mx, pr[g2 * 68 + 64]);
                float val = 0.f, den = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < 16; ++g2) {
                    const float lg2 = pr[g2 * 68 + 65];
                    const float wgt = lg2 > 0.f ? expf(pr[g2 * 68 + 64] - Mx) : 0.f;
                    den = fmaf(wgt, lg2, den);
                    val = fmaf(wgt, pr[g2 * 68 + dd], val);
                }
                st_sc1(a.att + (long)(2 * s + row) * D + h * 64 + dd, val / den);
            }
            publish(a.f_att + h * B + s, ep);
        }
        AB_MARK();
        // ---- WO + residual ----
        ++ep;
        for (int u = wg; u < TMS * XT; u += G) {
            const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTS, n0 = nj * 16;
            const int s0 = m0 >> 1, ns = min(8 * MTS, B - s0);
            linear_tile<WT, MTS, 1, D, false>(AATT, m0, M2, reinterpret_cast<const WT*>(L.wo), n0, D, nullptr,
                                              [&] {
#pragma unroll
                                                  for (int hh = 0; hh < 3; ++hh) wait_flags(a.f_att + (3 * wave + hh) * B + s0, ns, ep - 1, a.fail, 3);
                                              }, red);
            epi_residual<MTS>(red, m0, M2, n0, a.xs, D, 0, a.xs, nullptr);
            publish(a.f_x + mi * XT + nj, ep);
        }
        AB_MARK();
        // ---- W13: RMSNorm + w1 | w3 + SwiGLU ----
        ++ep;
        for (int u = wg; u < TMS * GT; u += G) {
            const int mi = u / GT, nj = u - mi * GT, m0 = mi * 16 * MTS, n0 = nj * 32;
            linear_tile<WT, MTS, 2, D, true>(AXS, m0, M2, reinterpret_cast<const WT*>(L.w13), n0, 2 * I, L.ffn_norm,
                                             [&] { wait_flags(a.f_x + mi * XT + 12 * wave, 12, ep - 1, a.fail, 4); }, red);
            for (int i = wave; i < MTS; i += 4) {
                f32x4 t[2];
                tile_sum<MTS, 2>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r;
                    if (m >= M2) continue;
                    const float inv = row_inv<MTS, 2>(red, i, rq + r, D, 1e-5f);
                    st_sc1(a.g + (long)m * I + nj * 16 + col, silu_f(t[0][r] * inv) * (t[1][r] * inv));
                }
            }
            publish(a.f_g + mi * GT + nj, ep);
        }
        AB_MARK();
        // ---- W2 + residual ----
        ++ep;
        for (int u = wg; u < TMS * XT; u += G) {
            const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTS, n0 = nj * 16;
            linear_tile<WT, MTS, 1, I, false>(AG, m0, M2, reinterpret_cast<const WT*>(L.w2), n0, D, nullptr,
                                              [&] { wait_flags(a.f_g + mi * GT + 36 * wave, 36, ep - 1, a.fail, 5); }, red);
            epi_residual<MTS>(red, m0, M2, n0, a.xs, D, 0, a.xs, nullptr);
            publish(a.f_x + mi * XT + nj, ep);
        }
        e_x = ep;
        AB_MARK();
    }
    // wait for the content-token rows (2 s + 1) of the fast row tile mi: the slow row tiles that hold them
    auto wait_hidden = [&](int mi) {
        const int r_lo = 2 * (mi * 16 * MTF), r_hi = min(M2, 2 * (mi * 16 * MTF + 16 * MTF)) - 1;
        for (int ts = r_lo / (16 * MTS); ts <= r_hi / (16 * MTS); ++ts) wait_flags(a.f_x + ts * XT + 12 * wave, 12, e_x, a.fail, 6);
    };
    // ---- semantic-token logits (dual_ar_stream.py:1181-1186; the sample is discarded by every caller, :833) ----
    unsigned e_sem = 0;
    if (!a.skip_semantic) {
        ++ep;
        e_sem = ep;
        for (int u = wg; u < TMF * ST; u += G) {
            const int mi = u / ST, nj = u - mi * ST, m0 = mi * 16 * MTF, n0 = nj * 16;
            linear_tile<WT, MTF, 1, D, true>(AHID, m0, B, reinterpret_cast<const WT*>(a.out_w), n0, a.vocab, a.out_norm, [&] { wait_hidden(mi); }, red);
            for (int i = wave; i < MTF; i += 4) {
                f32x4 t[1];
                tile_sum<MTF, 1>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r, n = n0 + col;
                    if (m < B && n < a.vocab) st_sc1(a.slow_logits + (long)m * a.vocab + n, t[0][r] * row_inv<MTF, 1>(red, i, rq + r, D, 1e-5f));
                }
            }
            publish(a.f_sem + mi * SEMT + nj, ep);
        }
        AB_MARK();
    }

    // ======================================= fast AR: 8 codebooks x 4 layers on B rows =======================================
    unsigned e_row = 0;         // epoch of the previous codebook's sampler (which wrote the xf rows)
    for (int cb = 0; cb < NCB; ++cb) {
        for (int l = 0; l < AR_FAST_LAYERS; ++l) {
            const ArLayerW& L = a.fast[l];
            const bool first = l == 0, from_slow = first && cb == 0;
            // ---- FQKV: RMSNorm + wqkv + RoPE (position = codebook index) + K / V of this position ----
            ++ep;
            for (int u = wg; u < TMF * QT; u += G) {
                const int mi = u / QT, nj = u - mi * QT, m0 = mi * 16 * MTF, n0 = nj * 16;
                auto waitf = [&] {
                    if (from_slow) wait_hidden(mi);
                    else if (first) wait_flags(a.f_row + m0, min(16 * MTF, B - m0), e_row, a.fail, 7);
                    else wait_flags(a.f_xf + mi * XT + 12 * wave, 12, ep - 1, a.fail, 8);
                };
                linear_tile<WT, MTF, 1, D, true>(from_slow ? AHID : AXF, m0, B, reinterpret_cast<const WT*>(L.wqkv), n0, 3 * D, L.attn_norm, waitf, red);
                const int region = n0 / D, nn = n0 + col - region * D, d = nn & 63;
                const float c = a.rope_fast[(cb * 32 + (d >> 1)) * 2], sn = a.rope_fast[(cb * 32 + (d >> 1)) * 2 + 1];
                for (int i = wave; i < MTF; i += 4) {
                    f32x4 t[1];
                    tile_sum<MTF, 1>(red, i, lane, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + 16 * i + rq + r;
                        float v = t[0][r] * row_inv<MTF, 1>(red, i, rq + r, D, 1e-5f);
                        const float pv = lane_xor_f<1>(v);
                        if (region < 2) v = (d & 1) ? v * c + pv * sn : v * c - pv * sn;
                        if (m < B) {
                            st_sc1(a.qkvf + (long)m * 3 * D + n0 + col, v);
                            if (region >= 1) st_sc1(a.kvf + (((long)l * B + m) * NCB + cb) * 2 * D + (n0 + col - D), v);
                        }
                    }
                }
                publish(a.f_qkvf + mi * QT + nj, ep);
            }
            AB_MARK();
            // ---- FATT: attention over the <= 8 codebook positions, unit = stream (wave w: heads 3w .. 3w + 2) ----
            ++ep;
            for (int u = wg; u < B; u += G) {
                const int s = u, mi = s / (16 * MTF);
                wait_flags(a.f_qkvf + mi * QT + 36 * wave, 36, ep - 1, a.fail, 9);
                __syncthreads();
                float* big = red;                  // [2304] this stream's q | k | v row
                float* av = red + 3 * D;           // [768]
                {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.qkvf + (long)s * 3 * D, 0, 3 * D * 4, 0x00020000);
                    for (int i = tid; i < 3 * D / 4; i += 256)
                        *reinterpret_cast<float4*>(big + 4 * i) = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16));
                }
                const float* kvg = a.kvf + (((long)l * B + s) * NCB) * 2 * D;         // [8][k 768 | v 768]
                const int kg = lane >> 4, kli = lane & 15;
                unsigned long long pk[3][2][2];
                float pvv[3][7];
#pragma unroll
                for (int hh = 0; hh < 3; ++hh) {
                    const int hb = (wave * 3 + hh) * 64;
#pragma unroll
                    for (int rnd = 0; rnd < 2; ++rnd) {
                        const int t = kg + 4 * rnd;
                        if (t < cb) {
                            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(kvg + (long)t * 2 * D + hb + 4 * kli);
                            pk[hh][rnd][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            pk[hh][rnd][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 7; ++t)
                        if (t < cb) pvv[hh][t] = ld_sc1(kvg + (long)t * 2 * D + D + hb + lane);
                }
                __syncthreads();
#pragma unroll
                for (int hh = 0; hh < 3; ++hh) {
                    const int hb = (wave * 3 + hh) * 64;
                    const float4 q4 = *reinterpret_cast<const float4*>(big + hb + 4 * kli);
                    float sc2[2];
#pragma unroll
                    for (int rnd = 0; rnd < 2; ++rnd) {
                        const int t = kg + 4 * rnd;
                        float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (t < cb) {
                            k4 = make_float4(__uint_as_float((unsigned)pk[hh][rnd][0]), __uint_as_float((unsigned)(pk[hh][rnd][0] >> 32)),
                                             __uint_as_float((unsigned)pk[hh][rnd][1]), __uint_as_float((unsigned)(pk[hh][rnd][1] >> 32)));
                        } else if (t == cb) {
                            k4 = *reinterpret_cast<const float4*>(big + D + hb + 4 * kli);
                        }
                        const float dot = row16_sum(q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w) * 0.125f;
                        sc2[rnd] = t <= cb ? dot : -INFINITY;
                    }
                    const float mx = wave_max(fmaxf(sc2[0], sc2[1]));
                    const float e0 = sc2[0] > -INFINITY ? expf(sc2[0] - mx) : 0.f, e1 = sc2[1] > -INFINITY ? expf(sc2[1] - mx) : 0.f;
                    const float inv = 16.f / wave_sum(e0 + e1);                  // every row holds its value 16 times
                    float acc = 0.f;
#pragma unroll
                    for (int t = 0; t < NCB; ++t) {
                        const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (t >> 2) ? e1 : e0), (t & 3) * 16));
                        const float vd = t < cb ? (t < 7 ? pvv[hh][t < 7 ? t : 0] : 0.f) : big[2 * D + hb + lane];
                        if (t <= cb) acc = fmaf(e, vd, acc);
                    }
                    av[hb + lane] = acc * inv;
                }
                __syncthreads();
                for (int i = tid; i < D; i += 256) st_sc1(a.attf + (long)s * D + i, av[i]);
                publish(a.f_attf + s, ep);
            }
            AB_MARK();
            // ---- FWO + residual ----
            ++ep;
            for (int u = wg; u < TMF * XT; u += G) {
                const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTF, n0 = nj * 16;
                linear_tile<WT, MTF, 1, D, false>(AATTF, m0, B, reinterpret_cast<const WT*>(L.wo), n0, D, nullptr,
                                                  [&] { wait_flags(a.f_attf + m0, min(16 * MTF, B - m0), ep - 1, a.fail, 10); }, red);
                // hidden = pre-norm state of the content token (forward_generate :340-341): tap, and the fast AR's first input
                if (from_slow) epi_residual<MTF>(red, m0, B, n0, a.xs, 2 * D, D, a.xf, a.hidden);
                else epi_residual<MTF>(red, m0, B, n0, a.xf, D, 0, a.xf, nullptr);
                publish(a.f_xf + mi * XT + nj, ep);
            }
            AB_MARK();
            // ---- FW13 ----
            ++ep;
            for (int u = wg; u < TMF * GT; u += G) {
                const int mi = u / GT, nj = u - mi * GT, m0 = mi * 16 * MTF, n0 = nj * 32;
                linear_tile<WT, MTF, 2, D, true>(AXF, m0, B, reinterpret_cast<const WT*>(L.w13), n0, 2 * I, L.ffn_norm,
                                                 [&] { wait_flags(a.f_xf + mi * XT + 12 * wave, 12, ep - 1, a.fail, 11); }, red);
                for (int i = wave; i < MTF; i += 4) {
                    f32x4 t[2];
                    tile_sum<MTF, 2>(red, i, lane, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + 16 * i + rq + r;
                        if (m >= B) continue;
                        const float inv = row_inv<MTF, 2>(red, i, rq + r, D, 1e-5f);
                        st_sc1(a.gf + (long)m * I + nj * 16 + col, silu_f(t[0][r] * inv) * (t[1][r] * inv));
                    }
                }
                publish(a.f_gf + mi * GT + nj, ep);
            }
            AB_MARK();
            // ---- FW2 + residual ----
            ++ep;
            for (int u = wg; u < TMF * XT; u += G) {
                const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTF, n0 = nj * 16;
                linear_tile<WT, MTF, 1, I, false>(AGF, m0, B, reinterpret_cast<const WT*>(L.w2), n0, D, nullptr,
                                                  [&] { wait_flags(a.f_gf + mi * GT + 36 * wave, 36, ep - 1, a.fail, 12); }, red);
                epi_residual<MTF>(red, m0, B, n0, a.xf, D, 0, a.xf, nullptr);
                publish(a.f_xf + mi * XT + nj, ep);
            }
            AB_MARK();
        }
        // ---- HEAD: fast_norm + codebook logits ----
        ++ep;
        for (int u = wg; u < TMF * VT; u += G) {
            const int mi = u / VT, nj = u - mi * VT, m0 = mi * 16 * MTF, n0 = nj * 16;
            linear_tile<WT, MTF, 1, D, true>(AXF, m0, B, reinterpret_cast<const WT*>(a.fast_out_w), n0, V, a.fast_norm,
                                             [&] { wait_flags(a.f_xf + mi * XT + 12 * wave, 12, ep - 1, a.fail, 13); }, red);
            for (int i = wave; i < MTF; i += 4) {
                f32x4 t[1];
                tile_sum<MTF, 1>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r, n = n0 + col;
                    if (m < B && n < V) st_sc1(a.fast_logits + ((long)m * NCB + cb) * V + n, t[0][r] * row_inv<MTF, 1>(red, i, rq + r, D, 1e-5f));
                }
            }
            publish(a.f_log + mi * LOGT + nj, ep);
        }
        AB_MARK();
        // ---- SAMPLE: nucleus sample of one stream per unit, next input row = fast_emb[token] ----
        ++ep;
        for (int u = wg; u < B; u += G) {
            const int s = u, mi = s / (16 * MTF);
            wait_flags(a.f_log + mi * LOGT, VT, ep - 1, a.fail, 14);
            const float* lg = a.fast_logits + ((long)s * NCB + cb) * V;
            float lv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) lv[r] = (tid + 256 * r) < V ? ld_sc1(lg + tid + 256 * r) : -INFINITY;
            const float* nz = a.noise ? a.noise + (long)s * a.noise_ld + a.vocab + (long)cb * V : nullptr;
            const int raw = nucleus_sample<4, 4>(lv, V, tid, nz, a.seed[s], a.nframes[s], 1, cb * V, a.inv_temp, a.top_p, reinterpret_cast<double*>(red));
            int t = raw;
            if (use_forced) t = a.forced[((long)s * NCB + cb) * a.chunk + a.ci];
            if (tid == 0) { a.tok_raw[s * NCB + cb] = raw; st_sc1(a.tok + s * NCB + cb, t); }
            if (cb + 1 < NCB)
                for (int i = tid; i < D; i += 256) st_sc1(a.xf + (long)s * D + i, a.fast_emb[(long)t * D + i]);
            publish(a.f_row + s, ep);
        }
        e_row = ep;
        AB_MARK();
    }

    // ======================================= frame bookkeeping, unit = stream =======================================
    for (int u = wg; u < B; u += G) {
        const int s = u;
        wait_flags(a.f_row + s, 1, e_row, a.fail, 15);
        int* toks = reinterpret_cast<int*>(red + 512);       // (the samplers' scratch sits below)
        if (tid < NCB) toks[tid] = ld_sc1(a.tok + s * NCB + tid);
        __syncthreads();
        const int frame = a.nframes[s];
        // cached_new_audio_emb = embed(codes) (dual_ar_stream.py:834, 245-255): codebooks summed in order
        for (int i = tid; i < D; i += 256) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < NCB; ++q) acc += a.codebook_emb[((long)toks[q] + (long)q * V) * D + i];
            a.cached_audio_emb[(long)s * D + i] = acc;
        }
        if (tid < NCB) {
            a.pred_hist[((long)s * NCB + tid) * a.hist_cap + (frame & (a.hist_cap - 1))] = toks[tid];
            a.step_audio[((long)s * NCB + tid) * a.chunk + a.ci] = toks[tid];
        }
        if (!a.skip_semantic) {
            const int mi = s / (16 * MTF);
            for (int k = 0; k < (a.vocab + 15) / 16; k += 64) wait_flags(a.f_sem + mi * SEMT + k, min(64, (a.vocab + 15) / 16 - k), e_sem, a.fail, 16);
            float lv[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int e = tid + 256 * r;
                lv[r] = e < a.vocab ? ld_sc1(a.slow_logits + (long)s * a.vocab + e) : -INFINITY;
            }
            __syncthreads();
            const int sm = nucleus_sample<4, 32>(lv, a.vocab, tid, a.noise ? a.noise + (long)s * a.noise_ld : nullptr, a.seed[s], frame, 0, 0, a.inv_temp,
                                                 a.top_p, reinterpret_cast<double*>(red));
            if (tid == 0) a.sem[s] = sm;
        }
        __syncthreads();
        if (tid == 0) {
            a.nframes[s] = frame + 1;
            a.last_pos[s] += 2;
        }
    }
    AB_MARK();
    // the last workgroup out advances the epoch for the next launch (every workgroup has read it by then)
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)G - 1) {
            __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.epoch, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

constexpr size_t lds_floats(int MTS, int MTF) {
    const size_t lin = red_floats(MTS > MTF ? MTS : MTF, 2);
    const size_t att = 2 * 16 * 68, fatt = 4 * D, smp = 256;
    size_t m = lin;
    if (att > m) m = att;
    if (fatt > m) m = fatt;
    if (smp > m) m = smp;
    return m;
}

template <typename WT, typename KVT, int MTS, int MTF>
int launch_cfg(const ArBatchArgs& a, hipStream_t st) {
    const size_t smem = lds_floats(MTS, MTF) * sizeof(float);
    hipLaunchKernelGGL((ar_batch_kernel<WT, KVT, MTS, MTF>), dim3(a.G), dim3(256), smem, st, a);
    SVA_HIP(hipGetLastError());
    return 0;
}

template <typename WT, typename KVT, int MTS, int MTF>
int occupancy_cfg(int* blocks_per_cu) {
    SVA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)ar_batch_kernel<WT, KVT, MTS, MTF>, 256,
                                                         lds_floats(MTS, MTF) * sizeof(float)));
    return 0;
}

}  // namespace

void ar_batch_tiles(int B, int* mts, int* mtf) {
    // 16 MT rows per tile: one tile up to 64 rows, then 64-row tiles
    *mts = 2 * B <= 16 ? 1 : 2 * B <= 32 ? 2 : 4;
    *mtf = B <= 16 ? 1 : B <= 32 ? 2 : 4;
}

size_t ar_batch_flag_words(int B, size_t offs[11]) {
    int mts, mtf;
    ar_batch_tiles(B, &mts, &mtf);
    const size_t TMS = (2 * B + 16 * mts - 1) / (16 * mts), TMF = (B + 16 * mtf - 1) / (16 * mtf);
    const size_t sizes[11] = {TMS * XT, TMS * QT, TMS * GT, (size_t)H * B, TMF * XT, TMF * QT, TMF * GT, (size_t)B, TMF * LOGT, (size_t)B, TMF * SEMT};
    size_t o = 0;
    for (int i = 0; i < 11; ++i) {
        offs[i] = o;
        o += (sizes[i] + 63) / 64 * 64;
    }
    return o;
}

int ar_batch_wanted_workgroups(int B) {
    int mts, mtf;
    ar_batch_tiles(B, &mts, &mtf);
    const int TMS = (2 * B + 16 * mts - 1) / (16 * mts);
    return QT * TMS;
}

#define AB_DISPATCH(FN, ...)                                                                                   \
    do {                                                                                                       \
        int mts, mtf;                                                                                          \
        ar_batch_tiles(B_, &mts, &mtf);                                                                        \
        if (wt_half) {                                                                                         \
            if (mts == 1) return FN<__half, __half, 1, 1>(__VA_ARGS__);                                        \
            if (mts == 2) return FN<__half, __half, 2, 1>(__VA_ARGS__);                                        \
            if (mtf == 2) return FN<__half, __half, 4, 2>(__VA_ARGS__);                                        \
            return FN<__half, __half, 4, 4>(__VA_ARGS__);                                                      \
        }                                                                                                      \
        if (mts == 1) return FN<float, float, 1, 1>(__VA_ARGS__);                                              \
        if (mts == 2) return FN<float, float, 2, 1>(__VA_ARGS__);                                              \
        if (mtf == 2) return FN<float, float, 4, 2>(__VA_ARGS__);                                              \
        return FN<float, float, 4, 4>(__VA_ARGS__);                                                            \
    } while (0)

int ar_batch_occupancy(int wt_half, int B_, int* blocks_per_cu) { AB_DISPATCH(occupancy_cfg, blocks_per_cu); }

int launch_ar_batch(const ArBatchArgs& a, int wt_half, hipStream_t st) {
    const int B_ = a.B;
    SVA_CHECK(a.B >= 1 && a.B <= AR_BATCH_MAX_STREAMS && a.G >= 1, "ar_batch: 1..128 streams");
    SVA_CHECK(a.vocab <= 8192 && a.vocab <= SEMT * 16 && a.codebook_size <= 1024 && a.codebook_size <= LOGT * 16 && (a.hist_cap & (a.hist_cap - 1)) == 0,
              "ar_batch: unsupported head sizes");
    AB_DISPATCH(launch_cfg, a, st);
}

}  // namespace sva
