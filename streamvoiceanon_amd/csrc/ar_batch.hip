// Persistent batched decode kernel of the dual AR: ONE launch per frame for all B streams of a batch.
//
// decode_one_token_ar (modules/dual_ar_stream.py:1168-1219) for B streams is 12 slow layers on 2 B rows, the semantic head and
// 8 x (4 fast layers on B rows + codebook head + nucleus sample): ~240 dependent matrix products that each move a few MB of
// weights.  As separate launches (engine.hip: ar_layers_pass / ar_frame_tail, ~265 kernels of 5-25 us) the frame costs 2.5-3.8 ms
// at 8-64 streams although the weight stream (0.26 GB fp16 / 0.52 GB fp32) and the matrix work are worth a fraction of that.
// Here G workgroups of 8 waves stay resident for the whole frame and run a STATIC schedule of phases; phase p of a layer is cut
// into independent units that workgroup w takes round-robin (unit w, w + G, ...):
//   * linear phases (wqkv, wo, w1|w3, w2, heads): unit = (row tile of 16 MT rows, 16 NT output columns).  The K axis is split over
//     the 8 waves of the workgroup; a wave requests the WHOLE weight slice of its unit (3 or 9 32-wide K blocks per column tile)
//     BEFORE it looks at its input, so the weight stream's latency overlaps the hand-off of the previous phase's output -- the one
//     thing a launch boundary cannot do.  Lanes load weight and activation fragments straight into MFMA operands (no LDS in the K
//     loop), partial tiles are reduced through LDS once.  fp16 weights: v_mfma_f32_16x16x32_f16 with the fp32 activations split
//     exactly into hi + lo halves; fp32 weights: v_mfma_f32_16x16x4_f32.  RMSNorm is folded in (weight into the operand, 1 / rms
//     onto the accumulators); RoPE + KV write, residual add and SwiGLU are epilogues.
//   * slow attention: unit = (stream, head), both new rows against the cached keys (8 waves split the keys, merged in LDS);
//     fast attention (<= 8 codebook positions): unit = stream; samplers and bookkeeping: unit = stream.
//   * hand-off between phases = the data is the flag (cdna_hip_programming.md Guideline 16, form R2, as in ar_decode.hip): every
//     activation element crosses workgroups as an 8-byte {tag = epoch of the producing phase, value} granule, stored with a relaxed
//     agent-scope (sc1, write-through) store and read with 16-byte sc1 loads that are repeated until every tag of the wave's
//     fragment matches.  One memory round trip per edge: no flag words, no drains, no fences; nothing depends on workgroup
//     placement or dispatch order.  (The first version of this kernel published a flag per tile behind a drain + barrier and polled
//     the flag before loading the tile: three dependent round trips, 5.5-6 us per phase, 2.2-2.4 ms per frame at 8-12 streams.)
//     Buffers that feed a linear phase are laid out FRAGMENT-MAJOR: element (row m, k) sits at granule
//     (((k >> 3) * 4 + ((k >> 1) & 3)) * ROWS + m) * 2 + (k & 1), so that one 16-byte load instruction of a wave (lane = row x
//     k-octet, the MFMA operand layout) reads whole 256-byte runs -- 16 rows x 2 granules -- instead of 16 bytes out of every
//     64-byte row fragment: every unit of a phase re-reads the whole row tile (48-144 units x 16 MT rows x K x 8 bytes: as many
//     bytes as the weights from 8 streams on), and these sc1 reads are served past the L2, a cache line at a time.
//     All waits are bounded (timeout -> *fail, the kernel runs to its end with garbage instead of hanging).  A buffer is
//     rewritten only after every reader of its previous contents has finished: every unit of a consuming phase reads ALL column
//     tiles of its row tile, and the next writer's inputs depend (transitively) on the outputs of all those units.
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

constexpr int NWV = 8, NTH = NWV * 64;             // waves / threads per workgroup
constexpr int SPIN_LIMIT = 1 << 15;                // polls (a memory round trip each) before a wait gives up: tens of ms
constexpr int QT = 3 * D / 16, XT = D / 16, GT = I / 16;       // column tiles of the qkv (144), x / att (48) and SwiGLU (144) buffers
constexpr int HDR = 64 + 2 * AR_BATCH_MAX_STREAMS;  // LDS header (words): sampled tokens of the workgroup's streams [8][8], positions of the 2 B new tokens
constexpr int LOGP = 1024, SEMP = 8192;            // row pitch (granules) of the codebook / semantic logits

__device__ __forceinline__ void st_g(u64* p, unsigned ep, float v) { store_granule(p, ep, v); }
__device__ __forceinline__ u64 ld_g(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float g_val(u64 g) { return __uint_as_float((unsigned)g); }
__device__ __forceinline__ unsigned g_tag(u64 g) { return (unsigned)(g >> 32); }

// A fragment-major granule buffer of ROWS rows (see the header): logical row m lives at buffer row m * rs + ro
struct GBuf {
    u64* p;
    int rows, rs, ro;
};
__device__ __forceinline__ long gidx(const GBuf& b, int m, int k) {
    return ((long)((k >> 3) * 4 + ((k >> 1) & 3)) * b.rows + (m * b.rs + b.ro)) * 2 + (k & 1);
}
// A operand of a linear phase: a fragment-major granule buffer written earlier in this launch (sc1 loads; row_stride / off = bytes of
// one buffer row / of the view's first row inside a 16-byte column of ROWS pairs, kstep = bytes between consecutive 16-byte pieces of
// a k-octet), or plain fp32 rows written by an earlier launch (row_stride / off in bytes)
struct ASrc {
    __amdgpu_buffer_rsrc_t rs;
    int row_stride, off, kstep;
};
__device__ __forceinline__ ASrc make_gsrc(const GBuf& b, int K) {
    ASrc a;
    a.rs = __builtin_amdgcn_make_buffer_rsrc(b.p, 0, (int)((long)b.rows * K * 8), 0x00020000);
    a.row_stride = b.rs * 16;
    a.off = b.ro * 16;
    a.kstep = b.rows * 16;
    return a;
}
__device__ __forceinline__ ASrc make_psrc(const float* base, long elems, int row_stride_elems, int off_elems) {
    ASrc a;
    a.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
    a.row_stride = row_stride_elems * 4;
    a.off = off_elems * 4;
    a.kstep = 0;
    return a;
}
__device__ __forceinline__ v4i ld_g16(const __amdgpu_buffer_rsrc_t& rs, int voff) {
    return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 16));      // aux 16 = sc1
}
__device__ __forceinline__ float4 ld_p16(const __amdgpu_buffer_rsrc_t& rs, int voff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
}

// the thread index as a value the optimiser cannot see through: per-lane offsets derived from it stay inside the phase that uses them
// (from the plain threadIdx.x they are hoisted out of the layer loops, live through the whole kernel and end up in scratch, where
// every reload waits for all loads requested before it -- the in-order return that makes the weight prefetches free otherwise)
__device__ __forceinline__ int otid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// bounded poll bookkeeping of one wave: returns true when the wait must end (timeout, or another wait timed out earlier)
__device__ __forceinline__ bool spin_over(int& spins, int* fail, int code) {
    ++spins;
    if (spins > SPIN_LIMIT) {
        if ((threadIdx.x & 63) == 0) *fail = code;
        return true;
    }
    return (spins & 7) == 0 && *reinterpret_cast<volatile int*>(fail) != 0;
}

// L2 prefetch of the weight rows [n0, n0 + ncols) x K of a LATER unit of this workgroup: one 4-byte load per 128-byte line, results
// never used (kept in pf until the next call: by then a completed poll has proven them landed).  The unit's own 16/32-byte loads then hit
// this XCD's L2 instead of waiting for HBM on the critical path between two hand-offs.
struct Pf { int v0 = 0, v1 = 0, v2 = 0; };
__device__ __forceinline__ void pf_retire(Pf& pf) { asm volatile("" :: "v"(pf.v0), "v"(pf.v1), "v"(pf.v2)); }
template <typename WT>
__device__ __forceinline__ void touch_w(Pf& pf, const void* W, int n0, int ncols, int N, int K) {
    pf_retire(pf);
    const int t0 = otid();
    const int lpr = K * (int)sizeof(WT) / 128, total = ncols * lpr;            // <= 32 x 24 or 16 x 72 lines: at most 3 per thread
    const char* base = reinterpret_cast<const char*>(W);
    auto one = [&](int i, int& dst) {
        if (i < total) {
            const int r = i / lpr, cidx = i - r * lpr;
            int n = n0 + r;
            if (n > N - 1) n = N - 1;
            dst = *reinterpret_cast<const int*>(base + (long)n * K * sizeof(WT) + cidx * 128);
        }
    };
    one(t0, pf.v0);
    one(t0 + NTH, pf.v1);
    one(t0 + 2 * NTH, pf.v2);
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

// SVA_DEBUG=ar_timing=1: stamps of one unit (thread 0 of workgroup 0)
struct Tm {
    bool on = false;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    int sweeps = 0;
};

// Partial tiles of C[m0 .. m0 + 16 MT) x [n0 .. n0 + 16 NT) = A[rows] (x RMSNorm weight) . W[cols]^T, K split over the 8 waves; on
// return the per-wave partial tiles sit in `red` ([8][MT * NT][64][4] floats, then [8][MT][16] row sums of squares) behind a barrier.
// GRAN: A is a granule buffer whose tags must equal `want`; otherwise plain fp32 rows of an earlier launch.  The wave's K slice is
// worked in chunks of ACH 32-wide blocks: the weights of the first two chunks are requested up front (BEFORE the input is looked at:
// they overlap the hand-off; an earlier unit's post() has pulled them into this XCD's L2), chunk c + 2 when chunk c has been
// multiplied.  pre() runs in every wave after the first weight requests and before the first activation load (prefetch of what the
// epilogue needs); post() once the wave's last activation chunk has landed (L2 prefetch of a later unit's weights: nothing of this
// unit may queue behind it -- loads return in order, so the epilogue must not wait for any load of its own).
template <typename WT, int MT, int NT, int K, bool RMS, bool GRAN, int NP = 1, typename PreF, typename PostF>
__device__ __forceinline__ void linear_tile(const ASrc& A, unsigned want, int m0, int M, const WT* __restrict__ W, int n0, int N,
                                            const float* __restrict__ rms_w, PreF&& pre, PostF&& post, float* red, int* fail, int code, Tm& tm) {
    if (tm.on) tm.t0 = wall_clock64();
    constexpr int NKW = K / (32 * NWV);                 // 32-wide K blocks per wave: 3 (K = 768) or 9 (K = 2304)
    constexpr int ACH = MT >= 4 ? 1 : 3;                // K blocks of A requested (and validated) together
    constexpr int NCH = NKW / ACH;
    static_assert(K % (32 * NWV) == 0 && NKW % ACH == 0, "K blocks per wave");
    // NP = 2 (single-chunk K only): the NT column tiles are multiplied in two passes of NT / 2 against the same landed A operands, the
    // second pass's weights requested while the first multiplies -- twice the columns per unit without twice the weight registers
    constexpr int NTC = NT / NP;
    static_assert(NP == 1 || (NP == 2 && NCH == 1 && NT % 2 == 0), "two column passes need a single K chunk");
    // (an opaque copy of the thread index: otherwise every per-lane offset of every phase is hoisted out of the layer loops and
    // lives -- spilled to scratch -- through the whole kernel; a scratch reload waits for everything requested before it)
    const int tid_ = otid();
    const int lane = tid_ & 63, wave = tid_ >> 6, fr = lane & 15, fk = lane >> 4;
    const int kbase = wave * (K / NWV) + 8 * fk;
    WReg<WT> wv[2][ACH][NTC];
    float4 nv[2][ACH][2];
    const WT* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + 16 * j + fr;
        if (n > N - 1) n = N - 1;
        wp[j] = W + (long)n * K + kbase;
    }
    auto issue_w = [&](int buf, int ch, int pass) {
#pragma unroll
        for (int d = 0; d < ACH; ++d) {
#pragma unroll
            for (int j = 0; j < NTC; ++j) wv[buf][d][j].load(wp[pass * NTC + j] + (ch * ACH + d) * 32);
            if constexpr (RMS) {
                if (pass == 0) {
                    nv[buf][d][0] = *reinterpret_cast<const float4*>(rms_w + kbase + (ch * ACH + d) * 32);
                    nv[buf][d][1] = *reinterpret_cast<const float4*>(rms_w + kbase + (ch * ACH + d) * 32 + 4);
                }
            }
        }
    };
    issue_w(0, 0, 0);
    if constexpr (NCH > 1) issue_w(1, 1, 0);
    int aoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = m0 + 16 * i + fr;
        if (m > M - 1) m = M - 1;
        // granules: piece q of k-octet o at ((o * 4 + q) * ROWS + row) * 16 bytes; this lane's octets: wave * (K / 64) + fk + 4 * block
        aoff[i] = GRAN ? A.off + m * A.row_stride + (wave * (K / 64) + fk) * 4 * A.kstep : A.off + m * A.row_stride + kbase * 4;
    }
    pre();
    if (tm.on) tm.t1 = wall_clock64();
    int sweeps = 0;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ssq[i] = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = ch * ACH;
        float4 av[ACH][MT][2];
        if constexpr (GRAN) {
            v4i g[ACH][MT][4];
            int spins = 0;
            while (true) {
                asm volatile("" ::: "memory");          // (the buffer-load builtin is an ordinary read to the optimiser: keep it inside the loop)
#pragma unroll
                for (int d = 0; d < ACH; ++d)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) g[d][i][q] = ld_g16(A.rs, aoff[i] + ((c + d) * 16 + q) * A.kstep);
                bool ok = true;
#pragma unroll
                for (int d = 0; d < ACH; ++d)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) ok = ok & ((unsigned)g[d][i][q].y == want) & ((unsigned)g[d][i][q].w == want);
                if (tm.on && ch == 0 && sweeps == 0) tm.t2 = wall_clock64();
                ++sweeps;
                if (__all(ok)) break;
                if (spin_over(spins, fail, code)) break;
                // not there yet: wait on ONE piece per lane with pauses (a full sweep per poll from every waiting wave of the chip is
                // terabytes per second of fabric traffic beside the producers' weight streams), then sweep again
                bool over = false;
                while (true) {
                    __builtin_amdgcn_s_sleep(8);
                    asm volatile("" ::: "memory");
                    const v4i t = ld_g16(A.rs, aoff[MT - 1] + ((c + ACH - 1) * 16 + 3) * A.kstep);
                    if (__all(((unsigned)t.y == want) & ((unsigned)t.w == want))) break;
                    if (spin_over(spins, fail, code)) { over = true; break; }
                }
                if (over) break;
            }
#pragma unroll
            for (int d = 0; d < ACH; ++d)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    av[d][i][0] = make_float4(__int_as_float(g[d][i][0].x), __int_as_float(g[d][i][0].z), __int_as_float(g[d][i][1].x), __int_as_float(g[d][i][1].z));
                    av[d][i][1] = make_float4(__int_as_float(g[d][i][2].x), __int_as_float(g[d][i][2].z), __int_as_float(g[d][i][3].x), __int_as_float(g[d][i][3].z));
                }
        } else {
#pragma unroll
            for (int d = 0; d < ACH; ++d)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    av[d][i][0] = ld_p16(A.rs, aoff[i] + (c + d) * 128);
                    av[d][i][1] = ld_p16(A.rs, aoff[i] + (c + d) * 128 + 16);
                }
        }
        if (ch == NCH - 1) {
            if (tm.on) { tm.t3 = wall_clock64(); tm.sweeps = sweeps; }
            post();
        }
        if constexpr (NP == 1) {
#pragma unroll
            for (int d = 0; d < ACH; ++d) {
                AOps<WT, MT> o;
                prep_block<WT, MT, RMS>(av[d], nv[ch & 1][d], o, ssq);
                mma_block<WT, MT, NT>(o, wv[ch & 1][d], acc);
            }
            if (ch + 2 < NCH) issue_w(ch & 1, ch + 2, 0);
        } else {
            issue_w(1, 0, 1);
            AOps<WT, MT> o[ACH];
#pragma unroll
            for (int d = 0; d < ACH; ++d) prep_block<WT, MT, RMS>(av[d], nv[0][d], o[d], ssq);
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                f32x4 accp[MT][NTC];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTC; ++j) accp[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < ACH; ++d) mma_block<WT, MT, NTC>(o[d], wv[pass][d], accp);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTC; ++j) acc[i][pass * NTC + j] = accp[i][j];
            }
        }
    }
    __syncthreads();            // the previous unit's epilogue has read `red`
    // cross-wave reduction of the K slices
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + j) * 64 + lane) * 4]) = acc[i][j];
    if constexpr (RMS) {
        float* redss = red + NWV * MT * NT * 256;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fk == 0) redss[(wave * MT + i) * 16 + fr] = v;
        }
    }
    __syncthreads();
    if (tm.on) tm.t4 = wall_clock64();
}

// sum of the eight waves' partials of row sub-tile i: t[j][r] = element (row 4 (lane >> 4) + r, column lane & 15) of column tile j
template <int MT, int NT>
__device__ __forceinline__ void tile_sum(const float* red, int i, int lane, f32x4 (&t)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x4 s = *reinterpret_cast<const f32x4*>(&red[((i * NT + j) * 64 + lane) * 4]);
#pragma unroll
        for (int w = 1; w < NWV; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * (MT * NT) + i * NT + j) * 64 + lane) * 4]);
        t[j] = s;
    }
}
template <int MT, int NT>
__device__ __forceinline__ float row_inv(const float* red, int i, int row16, int K, float eps) {
    const float* redss = red + NWV * MT * NT * 256;
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) tot += redss[(w * MT + i) * 16 + row16];
    return 1.f / sqrtf(tot / (float)K + eps);
}

constexpr size_t red_floats(int MT, int NT) { return (size_t)NWV * MT * NT * 256 + NWV * MT * 16; }

// residual granules of the lane's four rows of sub-tile i = wave, requested ahead of the unit's K loop (plain fp32 rows of an earlier
// launch when !GRAN).  They are usually in place by then, but nothing this workgroup has seen proves it: epi_residual checks the tags
template <int MT>
__device__ __forceinline__ void res_prefetch(const GBuf& res, int m0, int M, int n0, u64 (&rg)[4]) {
    const int t_ = otid(), lane = t_ & 63, wave = t_ >> 6, col = lane & 15, rq = (lane >> 4) * 4;
    if (wave < MT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int m = m0 + 16 * wave + rq + r;
            if (m > M - 1) m = M - 1;
            rg[r] = ld_g(res.p + gidx(res, m, n0 + col));
        }
    }
}
template <int MT>
__device__ __forceinline__ void res_prefetch_plain(const float* res, int m0, int M, int n0, u64 (&rg)[4]) {
    const int t_ = otid(), lane = t_ & 63, wave = t_ >> 6, col = lane & 15, rq = (lane >> 4) * 4;
    if (wave < MT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int m = m0 + 16 * wave + rq + r;
            if (m > M - 1) m = M - 1;
            rg[r] = (u64)__float_as_uint(res[(long)m * D + n0 + col]);
        }
    }
}
// wo / w2 epilogue: out[m][n] = res[m][n] + acc.  check: the residual granules must carry tag `want` (re-read until they do: by the
// time the unit's A operand has been validated they are in place, so this loop does not iterate in practice)
template <int MT>
__device__ __forceinline__ void epi_residual(const float* red, int m0, int M, int n0, u64 (&rg)[4], bool check, unsigned want, const GBuf& res,
                                             const GBuf& out, unsigned ep, float* tap, int* fail) {
    const int t_ = otid(), lane = t_ & 63, wave = t_ >> 6, col = lane & 15, rq = (lane >> 4) * 4;
    if (wave < MT) {
        const int i = wave;
        f32x4 t[1];
        tile_sum<MT, 1>(red, i, lane, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + rq + r;
            if (m >= M) continue;
            const int n = n0 + col;
            u64 g = rg[r];
            if (check) {
                int spins = 0;
                while (g_tag(g) != want) {
                    g = ld_g(res.p + gidx(res, m, n));
                    if (++spins > SPIN_LIMIT) { *fail = 17; break; }
                }
            }
            const float rv = g_val(g);
            if (tap) tap[(long)m * D + n] = rv;
            st_g(out.p + gidx(out, m, n), ep, rv + t[0][r]);
        }
    }
}

#define AB_MARK() do { if (a.dbg && wg == 0 && tid == 0 && nmark < 1000) a.dbg[nmark] = wall_clock64(); ++nmark; } while (0)

template <typename WT, typename KVT, int MTS, int MTF>
__global__ __launch_bounds__(NTH, 2) void ar_batch_kernel(const ArBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* const toks = reinterpret_cast<int*>(lds);         // [8][8] sampled tokens of the 8 streams this workgroup samples (wave = stream)
    int* const pos_s = toks + 64;                          // [2 B] cache positions of the frame's new tokens (dual_ar_stream.py:821-824)
    float* const red = lds + HDR;
    const int wg = blockIdx.x, G = a.G;
    // thread indices are re-derived from an opaque copy in every phase (AB_IDS): see linear_tile
#define AB_IDS() const int tid = otid(), lane = tid & 63, wave = tid >> 6, col = lane & 15, rq = (lane >> 4) * 4; \
    (void)lane; (void)wave; (void)col; (void)rq
    const int tid = threadIdx.x;
    const int B = a.B, M2 = 2 * B;
    const int TMS = (M2 + 16 * MTS - 1) / (16 * MTS), TMF = (B + 16 * MTF - 1) / (16 * MTF);
    const int V = a.codebook_size, VT = (V + 15) / 16, ST = (a.vocab + 15) / 16;
    // column tiles per unit of the wide phases (wqkv: 144 tiles, w1 | w3: 144 gate / up pairs): two each up to 32 rows, so that EVERY
    // linear phase has at most 72 units per row tile and a launch of 72-96 workgroups takes each phase in one round
    constexpr int QNS = MTS >= 4 ? 1 : 2, GNS = MTS >= 4 ? 1 : 2, QNF = MTF >= 4 ? 1 : 2, GNF = MTF >= 4 ? 1 : 2;
    constexpr int QUS = QT / QNS, GUS = GT / GNS, QUF = QT / QNF, GUF = GT / GNF;
    __builtin_amdgcn_s_setprio(3);
    unsigned ep = *a.epoch;
    int nmark = 0;
    const int use_forced = *a.use_forced;
    const long SH = (long)a.S * 64;
    const ASrc AIN = make_psrc(a.xs_in, (long)M2 * D, D, 0);
    const int RS = TMS * 16 * MTS, RF = TMF * 16 * MTF;               // rows of the slow / fast fragment-major buffers
    const GBuf GXS{a.gxs, RS, 1, 0}, GATT{a.gatt, RS, 1, 0}, GG{a.gg, RS, 1, 0};
    const GBuf GHID{a.gxs, RS, 2, 1};                                  // content-token rows of the slow residual stream
    const GBuf GXF{a.gxf, RF, 1, 0}, GATTF{a.gattf, RF, 1, 0}, GGF{a.ggf, RF, 1, 0};
    const ASrc AXS = make_gsrc(GXS, D), AATT = make_gsrc(GATT, D), AG = make_gsrc(GG, I), AHID = make_gsrc(GHID, D);
    const ASrc AXF = make_gsrc(GXF, D), AATTF = make_gsrc(GATTF, D), AGF = make_gsrc(GGF, I);
    // SVA_DEBUG=ar_timing=1: inside the linear phases, thread 0 of workgroup 0 accumulates per phase kind (dbg[512 + 8 kind + j]):
    // j = 0 weights requested, 1 first sweep of the input back, 2 input validated, 3 partial tiles reduced, 4 epilogue stored (all in
    // 10 ns ticks since the unit began, summed over the frame), 5 sweeps, 6 units
    Tm tm;
    tm.on = a.dbg && wg == 0 && tid == 0;
    if (tm.on)
        for (int i = 512; i < 512 + 8 * 9; ++i) a.dbg[i] = 0;
#define AB_ACC(kind) do { if (tm.on && u == wg) { const long long now = wall_clock64(); long long* q_ = a.dbg + 512 + 8 * (kind); \
        q_[0] += tm.t1 - tm.t0; q_[1] += tm.t2 - tm.t0; q_[2] += tm.t3 - tm.t0; q_[3] += tm.t4 - tm.t0; q_[4] += now - tm.t0; q_[5] += tm.sweeps; q_[6] += 1; } } while (0)
    Pf pf;
    // L2 prefetch of the weights of this workgroup's FIRST unit (u = wg) of a later linear phase with `tiles` column tiles of `ncols` rows
    auto touch = [&](const void* W, int tm, int tiles, int ncols, int N, int K) {
        if (wg < tm * tiles) touch_w<WT>(pf, W, (wg % tiles) * ncols, ncols, N, K);
    };
    for (int m = tid; m < M2; m += NTH) pos_s[m] = a.last_pos[m >> 1] + 1 + (m & 1);
    __syncthreads();
    AB_MARK();

    // ======================================= slow AR: 12 layers on 2 B rows =======================================
    unsigned e_x = 0;           // epoch of the phase that last wrote gxs
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const ArLayerW& L = a.slow[l];
        KVT* const kvl = reinterpret_cast<KVT*>(a.kv_slow) + (long)l * a.kv_layer_stride;
        // ---- QKV: RMSNorm + wqkv + RoPE + KV write ----
        ++ep;
        for (int u = wg; u < TMS * QUS; u += G) {
            AB_IDS();
            const int mi = u / QUS, nj = u - mi * QUS, m0 = mi * 16 * MTS, n0 = nj * 16 * QNS;
            float2 rcs[QNS][4];         // RoPE factors of the lane's rows (requested with the weights: the epilogue loads nothing)
            auto pre = [&] {
                if (wave < MTS) {
#pragma unroll
                    for (int j = 0; j < QNS; ++j) {
                        const int nt = n0 + 16 * j, region = nt / D, d = (nt + col - region * D) & 63;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = m0 + 16 * wave + rq + r;
                            rcs[j][r] = *reinterpret_cast<const float2*>(a.rope_slow + ((long)pos_s[m < M2 ? m : M2 - 1] * 32 + (d >> 1)) * 2);
                        }
                    }
                }
            };
            auto post = [&] { if (u == wg) touch(L.wo, TMS, XT, 16, D, D); };
            if (l == 0) linear_tile<WT, MTS, QNS, D, true, false, QNS>(AIN, 0u, m0, M2, reinterpret_cast<const WT*>(L.wqkv), n0, 3 * D, L.attn_norm, pre, post, red, a.fail, 1, tm);
            else linear_tile<WT, MTS, QNS, D, true, true, QNS>(AXS, e_x, m0, M2, reinterpret_cast<const WT*>(L.wqkv), n0, 3 * D, L.attn_norm, pre, post, red, a.fail, 1, tm);
            if (wave < MTS) {
                const int i = wave;
                f32x4 t[QNS];
                tile_sum<MTS, QNS>(red, i, lane, t);
#pragma unroll
                for (int j = 0; j < QNS; ++j) {
                    const int nt = n0 + 16 * j, region = nt / D, nn = nt + col - region * D, h = nn >> 6, d = nn & 63;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + 16 * i + rq + r, mc = m < M2 ? m : M2 - 1;
                        const int s = mc >> 1, pos = pos_s[mc];
                        float v = t[j][r] * row_inv<MTS, QNS>(red, i, rq + r, D, 1e-5f);
                        const float pv = lane_xor_f<1>(v);
                        if (region < 2) {
                            const float c = rcs[j][r].x, sn = rcs[j][r].y;
                            v = (d & 1) ? v * c + pv * sn : v * c - pv * sn;
                        }
                        if (m < M2) {
                            st_g(a.gqkv + (long)m * 3 * D + nt + col, ep, v);
                            if (region >= 1) st_kv<KVT>(kvl + (long)s * a.kv_slot_stride + ((long)(region - 1) * H + h) * SH + (long)pos * 64 + d, v);
                        }
                    }
                }
            }
            AB_ACC(0);
        }
        AB_MARK();
        // ---- ATT: (stream, head): both new rows against keys 0 .. p0 (+ 1) ----
        ++ep;
        for (int u = wg; u < B * H; u += G) {
            AB_IDS();
            const int s = u / H, h = u - s * H;
            const int p0 = pos_s[2 * s];                         // positions of the two new tokens: p0, p0 + 1
            const int grp = lane >> 4, li = lane & 15;
            const KVT* kc = kvl + (long)s * a.kv_slot_stride + (long)h * SH + li * 4;
            const KVT* vc = kc + (long)H * SH;
            const int lo = (int)((long)wave * p0 / NWV), hi = (int)((long)(wave + 1) * p0 / NWV);       // this wave's cached keys
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
            // q of both rows (every wave); the frame's own two keys / values (last wave: group 0 = row 0, group 1 = row 1)
            const __amdgpu_buffer_rsrc_t rq_ = __builtin_amdgcn_make_buffer_rsrc(a.gqkv + (long)(2 * s) * 3 * D, 0, 2 * 3 * D * 8, 0x00020000);
            const int qo = (h * 64 + li * 4) * 8;
            const bool newk = wave == NWV - 1 && grp < 2;
            const int ko = ((grp & 1) * 3 * D + D + h * 64 + li * 4) * 8;      // (every wave loads them: a load under a branch is waited for on the spot)
            v4i gq[4], gk[4];
            {
                int spins = 0;
                while (true) {
                    asm volatile("" ::: "memory");
                    gq[0] = ld_g16(rq_, qo); gq[1] = ld_g16(rq_, qo + 16);
                    gq[2] = ld_g16(rq_, qo + 3 * D * 8); gq[3] = ld_g16(rq_, qo + 3 * D * 8 + 16);
                    gk[0] = ld_g16(rq_, ko); gk[1] = ld_g16(rq_, ko + 16);
                    gk[2] = ld_g16(rq_, ko + D * 8); gk[3] = ld_g16(rq_, ko + D * 8 + 16);
                    const unsigned w_ = ep - 1;
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok & ((unsigned)gq[q].y == w_) & ((unsigned)gq[q].w == w_) & ((unsigned)gk[q].y == w_) & ((unsigned)gk[q].w == w_);
                    if (__all(ok)) break;
                    if (spin_over(spins, a.fail, 2)) break;
                }
            }
            const float4 q0 = make_float4(__int_as_float(gq[0].x) * 0.125f, __int_as_float(gq[0].z) * 0.125f, __int_as_float(gq[1].x) * 0.125f, __int_as_float(gq[1].z) * 0.125f);
            const float4 q1 = make_float4(__int_as_float(gq[2].x) * 0.125f, __int_as_float(gq[2].z) * 0.125f, __int_as_float(gq[3].x) * 0.125f, __int_as_float(gq[3].z) * 0.125f);
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
            if (newk) {
                // the two keys of this frame, from the freshly published rows: group 0 = key p0 (both rows), group 1 = key p0 + 1 (row 1 only)
                const float4 kk = make_float4(__int_as_float(gk[0].x), __int_as_float(gk[0].z), __int_as_float(gk[1].x), __int_as_float(gk[1].z));
                const float4 vv = make_float4(__int_as_float(gk[2].x), __int_as_float(gk[2].z), __int_as_float(gk[3].x), __int_as_float(gk[3].z));
                const float s1 = row16_sum(q1.x * kk.x + q1.y * kk.y + q1.z * kk.z + q1.w * kk.w);
                upd(s1, vv, mr1, ls1, o1);
                if (grp == 0) {
                    const float s0 = row16_sum(q0.x * kk.x + q0.y * kk.y + q0.z * kk.z + q0.w * kk.w);
                    upd(s0, vv, mr0, ls0, o0);
                }
            }
            __syncthreads();                                // the previous unit has read `red`
            float* part = red;                              // [2][32][68]
            {
                float* p0_ = part + (wave * 4 + grp) * 68;
                float* p1_ = part + (4 * NWV + wave * 4 + grp) * 68;
                *reinterpret_cast<float4*>(p0_ + li * 4) = o0;
                *reinterpret_cast<float4*>(p1_ + li * 4) = o1;
                if (li == 0) { p0_[64] = mr0; p0_[65] = ls0; p1_[64] = mr1; p1_[65] = ls1; }
            }
            __syncthreads();
            if (tid < 128) {
                const int row = tid >> 6, dd = tid & 63;
                const float* pr = part + row * 4 * NWV * 68;
                float Mx = -INFINITY;
#pragma unroll
                for (int g2 = 0; g2 < 4 * NWV; ++g2)
                    if (pr[g2 * 68 + 65] > 0.f) Mx = fmaxf(Mx, pr[g2 * 68 + 64]);
                float val = 0.f, den = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < 4 * NWV; ++g2) {
                    const float lg2 = pr[g2 * 68 + 65];
                    const float wgt = lg2 > 0.f ? expf(pr[g2 * 68 + 64] - Mx) : 0.f;
                    den = fmaf(wgt, lg2, den);
                    val = fmaf(wgt, pr[g2 * 68 + dd], val);
                }
                st_g(a.gatt + gidx(GATT, 2 * s + row, h * 64 + dd), ep, val / den);
            }
        }
        AB_MARK();
        // ---- WO + residual ----
        ++ep;
        for (int u = wg; u < TMS * XT; u += G) {
            AB_IDS();
            const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTS, n0 = nj * 16;
            u64 rg[4];
            linear_tile<WT, MTS, 1, D, false, true>(AATT, ep - 1, m0, M2, reinterpret_cast<const WT*>(L.wo), n0, D, nullptr,
                                                    [&] {
                                                        if (l == 0) res_prefetch_plain<MTS>(a.xs_in, m0, M2, n0, rg);
                                                        else res_prefetch<MTS>(GXS, m0, M2, n0, rg);
                                                    }, [&] { if (u == wg) touch(L.w13, TMS, GUS, 32 * GNS, 2 * I, D); }, red, a.fail, 3, tm);
            epi_residual<MTS>(red, m0, M2, n0, rg, l > 0, e_x, GXS, GXS, ep, nullptr, a.fail);
            AB_ACC(1);
        }
        AB_MARK();
        // ---- W13: RMSNorm + w1 | w3 + SwiGLU ----
        ++ep;
        for (int u = wg; u < TMS * GUS; u += G) {
            AB_IDS();
            const int mi = u / GUS, nj = u - mi * GUS, m0 = mi * 16 * MTS, n0 = nj * 32 * GNS;
            linear_tile<WT, MTS, 2 * GNS, D, true, true, GNS>(AXS, ep - 1, m0, M2, reinterpret_cast<const WT*>(L.w13), n0, 2 * I, L.ffn_norm, [] {},
                                                         [&] { if (u == wg) touch(L.w2, TMS, XT, 16, D, I); }, red, a.fail, 4, tm);
            if (wave < MTS) {
                const int i = wave;
                f32x4 t[2 * GNS];
                tile_sum<MTS, 2 * GNS>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r;
                    if (m >= M2) continue;
                    const float inv = row_inv<MTS, 2 * GNS>(red, i, rq + r, D, 1e-5f);
#pragma unroll
                    for (int j = 0; j < GNS; ++j)
                        st_g(a.gg + gidx(GG, m, (nj * GNS + j) * 16 + col), ep, silu_f(t[2 * j][r] * inv) * (t[2 * j + 1][r] * inv));
                }
            }
            AB_ACC(2);
        }
        AB_MARK();
        // ---- W2 + residual ----
        ++ep;
        for (int u = wg; u < TMS * XT; u += G) {
            AB_IDS();
            const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTS, n0 = nj * 16;
            u64 rg[4];
            linear_tile<WT, MTS, 1, I, false, true>(AG, ep - 1, m0, M2, reinterpret_cast<const WT*>(L.w2), n0, D, nullptr,
                                                    [&] { res_prefetch<MTS>(GXS, m0, M2, n0, rg); },
                                                    [&] {
                                                        if (u != wg) return;
                                                        if (l + 1 < AR_SLOW_LAYERS) touch(a.slow[l + 1].wqkv, TMS, QUS, 16 * QNS, 3 * D, D);
                                                        else if (!a.skip_semantic) touch(a.out_w, TMF, ST, 16, a.vocab, D);
                                                        else touch(a.fast[0].wqkv, TMF, QUF, 16 * QNF, 3 * D, D);
                                                    }, red, a.fail, 5, tm);
            epi_residual<MTS>(red, m0, M2, n0, rg, true, ep - 2, GXS, GXS, ep, nullptr, a.fail);
            AB_ACC(3);
        }
        e_x = ep;
        AB_MARK();
    }
    // ---- semantic-token logits (dual_ar_stream.py:1181-1186; the sample is discarded by every caller, :833) ----
    unsigned e_sem = 0;
    if (!a.skip_semantic) {
        ++ep;
        e_sem = ep;
        for (int u = wg; u < TMF * ST; u += G) {
            AB_IDS();
            const int mi = u / ST, nj = u - mi * ST, m0 = mi * 16 * MTF, n0 = nj * 16;
            linear_tile<WT, MTF, 1, D, true, true>(AHID, e_x, m0, B, reinterpret_cast<const WT*>(a.out_w), n0, a.vocab, a.out_norm, [] {},
                                                   [&] { if (u == wg) touch(a.fast[0].wqkv, TMF, QUF, 16 * QNF, 3 * D, D); }, red, a.fail, 6, tm);
            if (wave < MTF) {
                const int i = wave;
                f32x4 t[1];
                tile_sum<MTF, 1>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r, n = n0 + col;
                    if (m < B && n < a.vocab) st_g(a.gsem + (long)m * SEMP + n, ep, t[0][r] * row_inv<MTF, 1>(red, i, rq + r, D, 1e-5f));
                }
            }
        }
        // the semantic sample itself, unit = stream (8 waves x 16 logits per lane).  Here and not at the end of the frame: it reads the
        // stream's frame counter, which the wave that samples the stream's last codebook advances; every workgroup that runs such a
        // unit also has a unit in the first fast phase, so the advance is ordered behind this read
        for (int u = wg; u < B; u += G) {
            AB_IDS();
            const int s = u;
            float lv[16];
            const u64* lg = a.gsem + (long)s * SEMP;
            int spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < 16; ++r) {          // (no load under a branch: it would be waited for on the spot)
                    const int e = tid + NTH * r;
                    const u64 g = ld_g(lg + (e < a.vocab ? e : a.vocab - 1));
                    ok = ok & (g_tag(g) == e_sem);
                    lv[r] = e < a.vocab ? g_val(g) : -INFINITY;
                }
                if (__all(ok)) break;
                if (spin_over(spins, a.fail, 16)) break;
                __builtin_amdgcn_s_sleep(8);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (tid + NTH * r < a.vocab) a.slow_logits[(long)s * a.vocab + tid + NTH * r] = lv[r];
            __syncthreads();                   // the previous unit has read `red`
            const int sm = nucleus_sample<NWV, 16>(lv, a.vocab, tid, a.noise ? a.noise + (long)s * a.noise_ld : nullptr, a.seed[s], a.nframes[s], 0, 0,
                                                   a.inv_temp, a.top_p, reinterpret_cast<double*>(red));
            if (tid == 0) a.sem[s] = sm;
        }
        AB_MARK();
    }

    // ======================================= fast AR: 8 codebooks x 4 layers on B rows =======================================
    constexpr unsigned CBP = 5 * AR_FAST_LAYERS + 2;         // phases per codebook
    unsigned e_row = 0;         // epoch of the previous codebook's sampler (which wrote the gxf rows)
    for (int cb = 0; cb < NCB; ++cb) {
        for (int l = 0; l < AR_FAST_LAYERS; ++l) {
            const ArLayerW& L = a.fast[l];
            const bool first = l == 0, from_slow = first && cb == 0;
            // ---- FQKV: RMSNorm + wqkv + RoPE (position = codebook index) + K / V of this position ----
            ++ep;
            for (int u = wg; u < TMF * QUF; u += G) {
                AB_IDS();
                const int mi = u / QUF, nj = u - mi * QUF, m0 = mi * 16 * MTF, n0 = nj * 16 * QNF;
                float2 cs[QNF];
#pragma unroll
                for (int j = 0; j < QNF; ++j) {
                    const int nt = n0 + 16 * j, region = nt / D, d = (nt + col - region * D) & 63;
                    cs[j] = *reinterpret_cast<const float2*>(a.rope_fast + (cb * 32 + (d >> 1)) * 2);
                }
                linear_tile<WT, MTF, QNF, D, true, true, QNF>(from_slow ? AHID : AXF, from_slow ? e_x : first ? e_row : ep - 1, m0, B,
                                                         reinterpret_cast<const WT*>(L.wqkv), n0, 3 * D, L.attn_norm, [] {},
                                                         [&] { if (u == wg) touch(L.wo, TMF, XT, 16, D, D); }, red, a.fail, 7, tm);
                if (wave < MTF) {
                    const int i = wave;
                    f32x4 t[QNF];
                    tile_sum<MTF, QNF>(red, i, lane, t);
#pragma unroll
                    for (int j = 0; j < QNF; ++j) {
                        const int nt = n0 + 16 * j, region = nt / D, d = (nt + col - region * D) & 63;
                        const float c = cs[j].x, sn = cs[j].y;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = m0 + 16 * i + rq + r;
                            float v = t[j][r] * row_inv<MTF, QNF>(red, i, rq + r, D, 1e-5f);
                            const float pv = lane_xor_f<1>(v);
                            if (region < 2) v = (d & 1) ? v * c + pv * sn : v * c - pv * sn;
                            if (m < B) {
                                st_g(a.gqkvf + (long)m * 3 * D + nt + col, ep, v);
                                if (region >= 1) st_g(a.gkvf + (((long)l * B + m) * NCB + cb) * 2 * D + (nt + col - D), ep, v);
                            }
                        }
                    }
                }
                AB_ACC(4);
            }
            AB_MARK();
            // ---- FATT: attention over the <= 8 codebook positions, unit = stream (wave w: heads w, w + 8) ----
            ++ep;
            for (int u = wg; u < B; u += G) {
                AB_IDS();
                const int s = u;
                __syncthreads();                   // the previous unit has read `red`
                float* big = red;                  // [2304] this stream's q | k | v row
                float* hist = red + 3 * D;         // [cb][k 768 | v 768] the earlier positions of this frame
                float* av = hist + 7 * 2 * D;      // [768]
                {
                    // q | k | v of this position (tag: this layer's FQKV) and K / V of positions t < cb (tag: the same phase, cb - t codebooks ago)
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.gqkvf + (long)s * 3 * D, 0, 3 * D * 8, 0x00020000);
                    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.gkvf + (((long)l * B + s) * NCB) * 2 * D, 0, NCB * 2 * D * 8, 0x00020000);
                    constexpr int NQ = 3 * D / 2, NH1 = 2 * D / 2;       // 16-byte pairs of the row, of one position
                    const int nh = cb * NH1;
                    v4i gq[3], gh[11];
                    int spins = 0;
                    while (true) {
                        asm volatile("" ::: "memory");
                        bool ok = true;         // (every load unconditional, clamped: a load under a branch is waited for on the spot)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int p = tid + k * NTH;
                            gq[k] = ld_g16(rs, (p < NQ ? p : NQ - 1) * 16);
                            ok = ok & ((unsigned)gq[k].y == ep - 1) & ((unsigned)gq[k].w == ep - 1);
                        }
#pragma unroll
                        for (int k = 0; k < 11; ++k) {
                            const int p = tid + k * NTH, pc = p < nh ? p : 0;
                            gh[k] = ld_g16(rh, pc * 16);
                            const unsigned want = ep - 1 - (unsigned)(cb - pc / NH1) * CBP;
                            ok = ok & ((p >= nh) | (((unsigned)gh[k].y == want) & ((unsigned)gh[k].w == want)));
                        }
                        if (__all(ok)) break;
                        if (spin_over(spins, a.fail, 9)) break;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int p = tid + k * NTH;
                        if (p < NQ) *reinterpret_cast<float2*>(big + 2 * p) = make_float2(__int_as_float(gq[k].x), __int_as_float(gq[k].z));
                    }
#pragma unroll
                    for (int k = 0; k < 11; ++k) {
                        const int p = tid + k * NTH;
                        if (p < nh) *reinterpret_cast<float2*>(hist + 2 * p) = make_float2(__int_as_float(gh[k].x), __int_as_float(gh[k].z));
                    }
                }
                __syncthreads();
                const int kg = lane >> 4, kli = lane & 15;
                for (int h = wave; h < H; h += NWV) {
                    const int hb = h * 64;
                    const float4 q4 = *reinterpret_cast<const float4*>(big + hb + 4 * kli);
                    float sc2[2];
#pragma unroll
                    for (int rnd = 0; rnd < 2; ++rnd) {
                        const int t = kg + 4 * rnd;
                        float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (t < cb) k4 = *reinterpret_cast<const float4*>(hist + t * 2 * D + hb + 4 * kli);
                        else if (t == cb) k4 = *reinterpret_cast<const float4*>(big + D + hb + 4 * kli);
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
                        if (t <= cb) acc = fmaf(e, t < cb ? hist[t * 2 * D + D + hb + lane] : big[2 * D + hb + lane], acc);
                    }
                    av[hb + lane] = acc * inv;
                }
                __syncthreads();
                for (int i = tid; i < D; i += NTH) st_g(a.gattf + gidx(GATTF, s, i), ep, av[i]);
            }
            AB_MARK();
            // ---- FWO + residual ----
            ++ep;
            for (int u = wg; u < TMF * XT; u += G) {
                AB_IDS();
                const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTF, n0 = nj * 16;
                u64 rg[4];
                linear_tile<WT, MTF, 1, D, false, true>(AATTF, ep - 1, m0, B, reinterpret_cast<const WT*>(L.wo), n0, D, nullptr,
                                                        [&] {
                                                            if (from_slow) res_prefetch<MTF>(GHID, m0, B, n0, rg);
                                                            else res_prefetch<MTF>(GXF, m0, B, n0, rg);
                                                        }, [&] { if (u == wg) touch(L.w13, TMF, GUF, 32 * GNF, 2 * I, D); }, red, a.fail, 10, tm);
                // hidden = pre-norm state of the content token (forward_generate :340-341): tap, and the fast AR's first input
                if (from_slow) epi_residual<MTF>(red, m0, B, n0, rg, true, e_x, GHID, GXF, ep, a.hidden, a.fail);
                else epi_residual<MTF>(red, m0, B, n0, rg, true, first ? e_row : ep - 3, GXF, GXF, ep, nullptr, a.fail);
                AB_ACC(5);
            }
            AB_MARK();
            // ---- FW13 ----
            ++ep;
            for (int u = wg; u < TMF * GUF; u += G) {
                AB_IDS();
                const int mi = u / GUF, nj = u - mi * GUF, m0 = mi * 16 * MTF, n0 = nj * 32 * GNF;
                linear_tile<WT, MTF, 2 * GNF, D, true, true, GNF>(AXF, ep - 1, m0, B, reinterpret_cast<const WT*>(L.w13), n0, 2 * I, L.ffn_norm, [] {},
                                                             [&] { if (u == wg) touch(L.w2, TMF, XT, 16, D, I); }, red, a.fail, 11, tm);
                if (wave < MTF) {
                    const int i = wave;
                    f32x4 t[2 * GNF];
                    tile_sum<MTF, 2 * GNF>(red, i, lane, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + 16 * i + rq + r;
                        if (m >= B) continue;
                        const float inv = row_inv<MTF, 2 * GNF>(red, i, rq + r, D, 1e-5f);
#pragma unroll
                        for (int j = 0; j < GNF; ++j)
                            st_g(a.ggf + gidx(GGF, m, (nj * GNF + j) * 16 + col), ep, silu_f(t[2 * j][r] * inv) * (t[2 * j + 1][r] * inv));
                    }
                }
                AB_ACC(6);
            }
            AB_MARK();
            // ---- FW2 + residual ----
            ++ep;
            for (int u = wg; u < TMF * XT; u += G) {
                AB_IDS();
                const int mi = u / XT, nj = u - mi * XT, m0 = mi * 16 * MTF, n0 = nj * 16;
                u64 rg[4];
                linear_tile<WT, MTF, 1, I, false, true>(AGF, ep - 1, m0, B, reinterpret_cast<const WT*>(L.w2), n0, D, nullptr,
                                                        [&] { res_prefetch<MTF>(GXF, m0, B, n0, rg); },
                                                        [&] {
                                                            if (u != wg) return;
                                                            if (l + 1 < AR_FAST_LAYERS) touch(a.fast[l + 1].wqkv, TMF, QUF, 16 * QNF, 3 * D, D);
                                                            else touch(a.fast_out_w, TMF, VT, 16, V, D);
                                                        }, red, a.fail, 12, tm);
                epi_residual<MTF>(red, m0, B, n0, rg, true, ep - 2, GXF, GXF, ep, nullptr, a.fail);
                AB_ACC(7);
            }
            AB_MARK();
        }
        // ---- HEAD: fast_norm + codebook logits ----
        ++ep;
        for (int u = wg; u < TMF * VT; u += G) {
            AB_IDS();
            const int mi = u / VT, nj = u - mi * VT, m0 = mi * 16 * MTF, n0 = nj * 16;
            linear_tile<WT, MTF, 1, D, true, true>(AXF, ep - 1, m0, B, reinterpret_cast<const WT*>(a.fast_out_w), n0, V, a.fast_norm, [] {},
                                                   [&] { if (u == wg && cb + 1 < NCB) touch(a.fast[0].wqkv, TMF, QUF, 16 * QNF, 3 * D, D); }, red, a.fail, 13, tm);
            if (wave < MTF) {
                const int i = wave;
                f32x4 t[1];
                tile_sum<MTF, 1>(red, i, lane, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + rq + r, n = n0 + col;
                    if (m < B && n < V) st_g(a.glog + (long)m * LOGP + n, ep, t[0][r] * row_inv<MTF, 1>(red, i, rq + r, D, 1e-5f));
                }
            }
            AB_ACC(8);
        }
        AB_MARK();
        // ---- SAMPLE: unit = 8 streams, one per wave (16 logits per lane, no barrier inside the sampler: ar_decode.hip); the next
        // input row = fast_emb[token] ----
        ++ep;
        for (int u = wg; u < (B + NWV - 1) / NWV; u += G) {
            AB_IDS();
            const int s = u * NWV + wave;
            if (s < B) {
                float lv[16];
                const u64* lg = a.glog + (long)s * LOGP;
                int spins = 0;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {      // (no load under a branch: it would be waited for on the spot)
                        const int e = lane + 64 * r;
                        const u64 g = ld_g(lg + (e < V ? e : V - 1));
                        ok = ok & (g_tag(g) == ep - 1);
                        lv[r] = e < V ? g_val(g) : -INFINITY;
                    }
                    if (__all(ok)) break;
                    if (spin_over(spins, a.fail, 14)) break;
                    __builtin_amdgcn_s_sleep(8);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (lane + 64 * r < V) a.fast_logits[((long)s * NCB + cb) * V + lane + 64 * r] = lv[r];
                const float* nz = a.noise ? a.noise + (long)s * a.noise_ld + a.vocab + (long)cb * V : nullptr;
                const int raw = nucleus_sample<1, 16>(lv, V, lane, nz, a.seed[s], a.nframes[s], 1, cb * V, a.inv_temp, a.top_p, nullptr);
                int t = raw;
                if (use_forced) t = a.forced[((long)s * NCB + cb) * a.chunk + a.ci];
                if (lane == 0) { a.tok_raw[s * NCB + cb] = raw; a.tok[s * NCB + cb] = t; toks[wave * NCB + cb] = t; }
                if (cb + 1 < NCB)
                    for (int i = lane; i < D; i += 64) st_g(a.gxf + gidx(GXF, s, i), ep, a.fast_emb[(long)t * D + i]);
            }
        }
        e_row = ep;
        AB_MARK();
    }

    // ======================================= frame bookkeeping: the wave that sampled a stream =======================================
    for (int u = wg; u < (B + NWV - 1) / NWV; u += G) {
        AB_IDS();
        const int s = u * NWV + wave;
        if (s < B) {
            const int* tk = toks + wave * NCB;                  // (written by this wave: program order)
            const int frame = a.nframes[s];
            // cached_new_audio_emb = embed(codes) (dual_ar_stream.py:834, 245-255): codebooks summed in order
            for (int i = lane; i < D; i += 64) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < NCB; ++q) acc += a.codebook_emb[((long)tk[q] + (long)q * V) * D + i];
                a.cached_audio_emb[(long)s * D + i] = acc;
            }
            if (lane < NCB) {
                a.pred_hist[((long)s * NCB + lane) * a.hist_cap + (frame & (a.hist_cap - 1))] = tk[lane];
                a.step_audio[((long)s * NCB + lane) * a.chunk + a.ci] = tk[lane];
            }
            if (lane == 0) {
                a.nframes[s] = frame + 1;
                a.last_pos[s] += 2;
            }
        }
    }
    AB_MARK();
    // the last workgroup out advances the epoch for the next launch (every workgroup has read it by then)
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)G - 1) {
            __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.epoch, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.fail_host) {      // a timeout of this launch reaches the host without a synchronising call (sva_step_device_on)
                const int f = *reinterpret_cast<volatile int*>(a.fail);
                if (f) *reinterpret_cast<volatile int*>(a.fail_host) = f;
            }
        }
    }
}

constexpr size_t lds_floats(int MTS, int MTF) {
    const int mt = MTS > MTF ? MTS : MTF;
    const size_t lin = red_floats(mt, mt >= 4 ? 2 : 4);
    const size_t att = 2 * 4 * NWV * 68, fatt = 3 * D + 7 * 2 * D + D, smp = 256;
    size_t m = lin;
    if (att > m) m = att;
    if (fatt > m) m = fatt;
    if (smp > m) m = smp;
    return HDR + m;
}

template <typename WT, typename KVT, int MTS, int MTF>
int launch_cfg(const ArBatchArgs& a, hipStream_t st) {
    const size_t smem = lds_floats(MTS, MTF) * sizeof(float);
    static bool attr_set = false;              // (more than 64 KiB of dynamic LDS needs the opt-in; idempotent, so a race only repeats it)
    if (!attr_set) {
        SVA_HIP(hipFuncSetAttribute((const void*)ar_batch_kernel<WT, KVT, MTS, MTF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((ar_batch_kernel<WT, KVT, MTS, MTF>), dim3(a.G), dim3(NTH), smem, st, a);
    SVA_HIP(hipGetLastError());
    return 0;
}

template <typename WT, typename KVT, int MTS, int MTF>
int occupancy_cfg(int* blocks_per_cu) {
    const size_t smem = lds_floats(MTS, MTF) * sizeof(float);
    SVA_HIP(hipFuncSetAttribute((const void*)ar_batch_kernel<WT, KVT, MTS, MTF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    SVA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)ar_batch_kernel<WT, KVT, MTS, MTF>, NTH, smem));
    return 0;
}

}  // namespace

void ar_batch_tiles(int B, int* mts, int* mtf) {
    // 16 MT rows per tile: one tile up to 64 rows, then 64-row tiles
    *mts = 2 * B <= 16 ? 1 : 2 * B <= 32 ? 2 : 4;
    *mtf = B <= 16 ? 1 : B <= 32 ? 2 : 4;
}

size_t ar_batch_granule_words(int B, size_t offs[AR_BATCH_NBUF]) {
    int mts, mtf;
    ar_batch_tiles(B, &mts, &mtf);
    const size_t b = (size_t)B;
    const size_t rs = (size_t)(2 * B + 16 * mts - 1) / (16 * mts) * 16 * mts, rf = (size_t)(B + 16 * mtf - 1) / (16 * mtf) * 16 * mtf;      // fragment-major buffers: whole row tiles
    const size_t sizes[AR_BATCH_NBUF] = {rs * D, 2 * b * 3 * D, rs * D, rs * I, rf * D, b * 3 * D, rf * D, rf * I,
                                         (size_t)AR_FAST_LAYERS * b * NCB * 2 * D, b * LOGP, b * SEMP};
    size_t o = 0;
    for (int i = 0; i < AR_BATCH_NBUF; ++i) {
        offs[i] = o;
        o += (sizes[i] + 31) / 32 * 32;
    }
    return o;
}

int ar_batch_wanted_workgroups(int B) {
    int mts, mtf;
    ar_batch_tiles(B, &mts, &mtf);
    const int TMS = (2 * B + 16 * mts - 1) / (16 * mts);
    return (mts >= 4 ? QT : QT / 2) * TMS;
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
    SVA_CHECK((a.B + NWV - 1) / NWV <= a.G && 3 * 48 * ((a.B + 63) / 64) >= 1, "ar_batch: one group of 8 streams per workgroup");
    SVA_CHECK(a.vocab <= SEMP && a.codebook_size <= 2 * NTH && a.codebook_size <= LOGP && (a.hist_cap & (a.hist_cap - 1)) == 0,
              "ar_batch: unsupported head sizes");
    AB_DISPATCH(launch_cfg, a, st);
}

}  // namespace sva
