// Single-stream persistent decode kernel of the dual AR, second generation: decode_one_token_ar
// (modules/dual_ar_stream.py:1168-1219) of ONE stream in one launch.  What the phase timeline of the first kernel (ar_decode.hip,
// profiles/r02_ar_timing_*.log) and the edge probe (tools/micro/ar_edge2.hip, profiles/r03_ar_edge2.log) showed: an all-to-all
// hand-off through the fabric costs 2.4-2.8 us whatever its size once the polling waves' own memory queues are quiet, weight loads
// requested a whole layer ahead add only ~0.3 us to it, and one polling wave per workgroup is far slower than four.  So:
//   * 192 workgroups x 4 waves.  A wave owns ONE output row of wo / w2, three rows of wqkv and 3 + 3 rows of w1 | w3: its share of
//     a layer is 156 fp32 registers.  wqkv and wo are requested a LAYER ahead of their use (the set a phase has consumed is
//     re-requested for the next layer at once), w1 | w3 two phases ahead and w2 one phase ahead (those two share registers:
//     keeping all four resident spills in fp32) -- no phase starts by waiting for a cold weight stream, and the arithmetic per
//     phase is half of the 96-workgroup kernel's.
//   * Hand-offs: 8-byte {tag = phase epoch, value} granules (sc1 stores, Guideline 16 form R2), polled by all 256 threads with
//     16-byte sc1 loads (two granules per load, each validated by its own tag).
//   * Narrow gathers where the dependency is narrow: the attention workgroup of (head, key slice) takes the 384 q / k / v values of
//     its head, not the 4608 of the phase; RoPE is applied by the consumer (a wave's three wqkv rows split an (even, odd) pair).
//   * Slow attention: 12 heads x 16 key slices, both tokens per workgroup (they share every cached K / V row); merge by the 24
//     (token, head) workgroups.
//   * Fast AR: every workgroup keeps ITS OWN copy of the fast K / V rows of the frame (global scratch, written and read by the same
//     workgroup): no cross-workgroup visibility question at all.  The nucleus sampler runs in each wave on its own (16 logits per
//     lane, no barrier), fused with the first projection of the next codebook step.
//   * The semantic head (8192 x 768, its sample discarded by every caller of decode_one) runs on 32 extra workgroups off the
//     critical path: they take the hidden state, publish the logits among themselves, one of them samples.
// All spins are bounded (a timeout sets *fail and lets the kernel run to its end with garbage instead of hanging).
#include "ar_decode.h"
#include "ar_device.h"
#include "sva_common.h"

namespace sva {
namespace {

using namespace ardev;

constexpr int NWG = AR2_WGS, NSEM = AR2_SEM_WGS, NT = AR2_THREADS;
constexpr int NWV = NWG * 4;                       // compute waves: 768 = D
static_assert(NWV == D && 3 * NWV == I, "row ownership: one wo / w2 row, three wqkv / w1 / w3 rows per compute wave");
constexpr int KSL = NWG / H;                       // key slices per head in the slow attention (16)
constexpr int SPIN_LIMIT = 1 << 16;                // polls before a gather gives up (~50 ms)

// granule buffers (u64 words from the stream's base)
constexpr int G_XA = 0, G_XB = G_XA + 2 * D, G_QKV = G_XB + 2 * D, G_H = G_QKV + 2 * I, G_ATT = G_H + 2 * I, G_A = G_ATT + NWG * 2 * 66,
              G_LOG = G_A + 2 * D, G_SEM = G_LOG + 1024, G_ACK = G_SEM + 8192, G_END = G_ACK + 64;
// LDS (floats)
constexpr int L_XA = 0, L_XB = L_XA + 2 * D, L_BIG = L_XB + 2 * D, L_AV = L_BIG + 2 * I, L_ATTP = L_AV + 2 * D, L_LG = L_ATTP + 1088,
              L_NA = L_LG + 1024, L_NC = L_NA + D, L_NH = L_NC + D, L_ROPE = L_NH + D, L_QKH = L_ROPE + NCB * 64, L_QKR = L_QKH + 384,
              L_END = L_QKR + 256;
constexpr int FAST_KV_WG = AR_FAST_LAYERS * 7 * 2 * D;      // floats of a workgroup's private fast K / V rows (positions 0..6)

typedef int v4i __attribute__((ext_vector_type(4)));

// all 256 threads: wait for npairs pairs of granules (tags == ep); pair i sits at granule offset goff(i) (even) of the stream's granule
// block and unpacks into dst[doff(i)], dst[doff(i) + 1].  One 16-byte sc1 load per pair.  stage_src: a [768] vector (RMSNorm weight)
// the first 192 threads copy into stage_dst on the way -- its load is issued before the polls and lands under them.
template <int PER, typename GF, typename DF>
__device__ __forceinline__ void gather16(__amdgpu_buffer_rsrc_t rs, int npairs, unsigned ep, GF goff, DF doff, float* dst, int* fail, int code,
                                         const float* stage_src = nullptr, float* stage_dst = nullptr) {
    const int tid = threadIdx.x;
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (stage_src && tid < D / 4) sv = *reinterpret_cast<const float4*>(stage_src + 4 * tid);
    unsigned pending = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (tid + k * 256 < npairs) pending |= 1u << k;
    int spins = 0;
    while (pending) {
        v4i x[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (pending & (1u << k)) x[k] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, goff(tid + k * 256) * 8, 0, 16));
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if ((pending & (1u << k)) && (unsigned)x[k].y == ep && (unsigned)x[k].w == ep) {
                *reinterpret_cast<float2*>(dst + doff(tid + k * 256)) = make_float2(__int_as_float(x[k].x), __int_as_float(x[k].z));
                pending &= ~(1u << k);
            }
        if (pending && (++spins & 255) == 0) {        // a rare look at the shared timeout word: once one workgroup has given up, all run through
            if (*reinterpret_cast<volatile int*>(fail)) break;
            if (spins > SPIN_LIMIT) { *fail = code; break; }
        }
    }
    if (stage_src && tid < D / 4) *reinterpret_cast<float4*>(stage_dst + 4 * tid) = sv;
}
// contiguous granules [g0, g0 + 2 npairs) -> dst[0 ...]
template <int PER>
__device__ __forceinline__ void gather_lin(__amdgpu_buffer_rsrc_t rs, int g0, int npairs, unsigned ep, float* dst, int* fail, int code,
                                           const float* stage_src = nullptr, float* stage_dst = nullptr) {
    gather16<PER>(rs, npairs, ep, [&](int i) { return g0 + 2 * i; }, [&](int i) { return 2 * i; }, dst, fail, code, stage_src, stage_dst);
}

template <typename WT> struct LayerRegs {        // a compute wave's rows of one layer
    WFrag<WT, D> qkv[3];
    WFrag<WT, D> wo[1];
    WFrag<WT, D> w13[6];
    WFrag<WT, I> w2[1];
};
template <typename WT> __device__ __forceinline__ void ld_qkv(LayerRegs<WT>& R, const ArLayerW& L, int gw, int lane) {
#pragma unroll
    for (int r = 0; r < 3; ++r) R.qkv[r].load(L.wqkv, 3L * gw + r, lane);
    asm volatile("" ::: "memory");
}
template <typename WT> __device__ __forceinline__ void ld_wo(LayerRegs<WT>& R, const ArLayerW& L, int gw, int lane) {
    R.wo[0].load(L.wo, gw, lane);
    asm volatile("" ::: "memory");
}
template <typename WT> __device__ __forceinline__ void ld_w13(LayerRegs<WT>& R, const ArLayerW& L, int gw, int lane) {
    // packed order of engine.hip w13(): the 96-wave blocks [w1 rows 6b..6b+5 | w3 rows 6b..6b+5]; wave gw owns w1 / w3 rows 3gw..3gw+2
    const long base = (long)(gw >> 1) * 12 + (gw & 1) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        R.w13[r].load(L.w13, base + r, lane);
        R.w13[3 + r].load(L.w13, base + 6 + r, lane);
    }
    asm volatile("" ::: "memory");
}
template <typename WT> __device__ __forceinline__ void ld_w2(LayerRegs<WT>& R, const ArLayerW& L, int gw, int lane) {
    R.w2[0].load(L.w2, gw, lane);
    asm volatile("" ::: "memory");
}

// SVA_AR_TIMING=1: workgroup 0 stamps wall_clock64() (100 MHz): [2k] = phase input gathered, [2k + 1] = phase
// outputs computed (thread 0)
#define MARK_W() do { if (s_dbg && wg == 0 && tid == 0) { s_dbg[nmark] = wall_clock64(); } ++nmark; } while (0)

template <typename WT, typename KVT>
__global__ __launch_bounds__(NT, 2) void ar_decode2_kernel(const ArDecodeArgs a) {
    // one stream per launch: the host passes this stream's pointers (launch_ar_decode2 advances them by the slot strides), so every
    // pointer below is a kernel argument the compiler can re-read from the kernarg segment instead of keeping it in a register
    const long long* const s_codes = a.codes;
    float* const s_cached_audio_emb = a.cached_audio_emb;
    int* const s_last_pos = a.last_pos;
    int* const s_nframes = a.nframes;
    const unsigned long long* const s_seed = a.seed;
    KVT* const s_kv_slow = reinterpret_cast<KVT*>(a.kv_slow);
    u64* const gran = a.gx;
    unsigned* const s_epoch = a.epoch;
    long long* const s_dbg = a.dbg;
    float* const s_slow_logits = a.slow_logits;
    float* const s_fast_logits = a.fast_logits;
    float* const s_hidden = a.hidden;
    int* const s_sem = a.sem;
    int* const s_tok_raw = a.tok_raw;
    int* const s_tok = a.tok;
    int* const s_step_audio = a.step_audio;
    int* const s_pred_hist = a.pred_hist;
    int* const s_step_content = a.step_content;
    const float* const s_noise = a.noise;
    const int* const s_forced = a.forced;
    u64 *const gxA = gran + G_XA, *const gxB = gran + G_XB, *const gqkv = gran + G_QKV, *const gh = gran + G_H, *const gatt = gran + G_ATT,
        *const ga = gran + G_A, *const glog = gran + G_LOG, *const gsem = gran + G_SEM, *const gack = gran + G_ACK;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const xsA = lds + L_XA;            // [2][768] layer input (fast AR: row 1)
    float* const xsB = lds + L_XB;            // [2][768] post-attention state
    float* const big = lds + L_BIG;           // [2][2304] SwiGLU output | fast qkv;  slow attention: per-group partials [2][16][68]
    float* const av = lds + L_AV;             // [2][768] attention output
    float* const attp = lds + L_ATTP;         // [16][66] key-slice partials of one (token, head)
    float* const lg = lds + L_LG;             // [1024] codebook logits
    float* const nrmA = lds + L_NA;           // attention_norm of the current layer
    float* const nrmC = lds + L_NC;           // ffn_norm
    float* const nrmH = lds + L_NH;           // fast_norm
    float* const ropef = lds + L_ROPE;        // [8][32][2] RoPE table of the codebook positions
    float* const qkh = lds + L_QKH;           // [q0 q1 k0 k1 v0 v1][64] this head's new rows, raw
    float* const qkr = lds + L_QKR;           // [q0 q1 k0 k1][64] after RoPE
    float* const scr = big;

    const int tid = threadIdx.x, lane = tid & 63, wg = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = wg * 4 + wave;             // 0 .. 767
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, G_END * 8, 0x00020000);
    __builtin_amdgcn_s_setprio(3);
    unsigned ep = *s_epoch;
    const unsigned ep0 = ep;
    int nmark = 0;
    const int p0 = *s_last_pos + 1;           // positions of the two new tokens (dual_ar_stream.py:821-824)
    const int frame = *s_nframes;
    const unsigned long long seed = *s_seed;
    const int code = (int)s_codes[a.code_off];
    const int use_forced = *a.use_forced;
    const long SH = (long)a.S * 64;           // one head of the cache
    constexpr unsigned EP_HIDDEN = 6 * AR_SLOW_LAYERS;      // phases before the hidden state is published

    // ======================================= semantic head: workgroups NWG .. NWG + NSEM - 1 =======================================
    // dual_ar_stream.py:1181-1186; the sampled token is discarded by every caller (:833) -- off the frame's critical path
    if (wg >= NWG) {
        const int sw = wg - NWG;
        if (tid == 0) store_granule(gack + sw, ep0 + 1, 0.f);       // "header read": workgroup 0 rewrites it only after these
        gather_lin<2>(rs, G_XA + D, D / 2, ep0 + EP_HIDDEN, xsA, a.fail, 20);
        __syncthreads();
        {
            const int sgw = sw * 4 + wave;     // 0 .. 127; rows sgw + 128 j
            for (int c8 = 0; c8 * 8 * (NSEM * 4) < a.vocab; ++c8) {
                WFrag<WT, D> w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int row = sgw + NSEM * 4 * (c8 * 8 + j);
                    if (row > a.vocab - 1) row = a.vocab - 1;
                    w[j].load(a.out_w, row, lane);
                }
                float o[1][8];
                gemv<WT, D, 8, 1, true>(w, xsA, D, a.out_norm, 1e-5f, lane, o);
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (lane == j) mine = o[0][j];
                const int row = sgw + NSEM * 4 * (c8 * 8 + lane);
                if (lane < 8 && row < a.vocab) {
                    s_slow_logits[row] = mine;
                    store_granule(gsem + row, ep0 + EP_HIDDEN, mine);
                }
            }
        }
        if (sw != 0) return;
        __syncthreads();
        // the first semantic workgroup samples (32 logits per thread)
        float* const sl = lds;
        gather_lin<16>(rs, G_SEM, a.vocab / 2, ep0 + EP_HIDDEN, sl, a.fail, 21);
        __syncthreads();
        float l[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int e = tid + NT * r;
            l[r] = e < a.vocab ? sl[e] : -INFINITY;
        }
        __syncthreads();
        const int s = nucleus_sample<4, 32>(l, a.vocab, tid, s_noise, seed, frame, 0, 0, a.inv_temp, a.top_p, reinterpret_cast<double*>(lds + 8192));
        if (tid == 0) *s_sem = s;
        return;
    }

    __shared__ int toks[NCB];
    const int hd = wg >> 4, sl = wg & 15;       // slow attention: head and key slice of this workgroup
    float* const xc = xsA + D;                 // fast AR: layer input / FB residual
    float* const xo = xsB + D;                 // fast AR: post-attention state / FD residual

    // RoPE factors this thread applies in the slow attention phase (token (tid >> 6) & 1, pair (tid & 63) >> 1): fixed for the frame
    float rope_c, rope_s;
    {
        const int m = (tid >> 6) & 1, d = tid & 63;
        rope_c = a.rope_slow[((long)(p0 + m) * 32 + (d >> 1)) * 2];
        rope_s = a.rope_slow[((long)(p0 + m) * 32 + (d >> 1)) * 2 + 1];
        // tokens [cached_new_audio_emb, src_cond] (decode_one, :817-837)
        for (int i = tid; i < D; i += 256) {
            xsA[i] = s_cached_audio_emb[i];
            xsA[D + i] = a.content_emb[(long)code * D + i];
        }
        for (int i = tid; i < NCB * 64; i += 256) ropef[i] = a.rope_fast[i];
        for (int i = tid; i < D; i += 256) { nrmA[i] = a.slow[0].attn_norm[i]; nrmH[i] = a.fast_norm[i]; }
    }
    LayerRegs<WT> R;
    ld_qkv(R, a.slow[0], gw, lane);
    ld_wo(R, a.slow[0], gw, lane);
    KVT* const kv = s_kv_slow;

    // ---------------------------------------- slow AR: 12 layers on M = 2 rows ----------------------------------------
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const ArLayerW& Ln = (l + 1 < AR_SLOW_LAYERS) ? a.slow[l + 1] : a.fast[0];
        KVT* const kl = kv + (long)l * a.kv_layer_stride;
        const int grp = lane >> 4, li = lane & 15;
        const int Lk1 = p0 + 2;                                    // keys 0 .. p0 + 1 (token 0 sees 0 .. p0)
        const int seg = sl * 16 + wave * 4 + grp;                  // 256 key segments per head
        const int lo = (int)((long)seg * Lk1 / (KSL * 16)), hi = (int)((long)(seg + 1) * Lk1 / (KSL * 16));
        const int hc = hi < p0 ? hi : p0;                          // cached keys of this segment: [lo, hc)
        const KVT* const kc = kl + (long)hd * SH + li * 4;
        const KVT* const vc = kc + (long)H * SH;
        float4 pkk[4], pvv[4];
        // ---- A: RMSNorm + wqkv (raw: RoPE is applied by the consumers) ----
        if (l > 0) gather_lin<3>(rs, G_XA, D, ep, xsA, a.fail, 1);
        MARK_W();
        __syncthreads();
        ++ep;
        {
            float o[2][3];
            gemv<WT, D, 3, 2, true>(R.qkv, xsA, D, nrmA, 1e-5f, lane, o);
            MARK_W();
            float mine = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (lane == m * 3 + r) mine = o[m][r];
            if (lane < 6) {
                const int m = lane / 3, r = lane - m * 3;
                store_granule(gqkv + m * I + 3 * gw + r, ep, mine);
            }
            // the cached K / V rows of this workgroup's key segment do not depend on this step: request them now
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int tk = lo + i;
                if (tk > hc - 1) tk = hc - 1;
                if (tk < 0) tk = 0;
                pkk[i] = ld_kv4<KVT>(kc + (long)tk * 64);
                pvv[i] = ld_kv4<KVT>(vc + (long)tk * 64);
            }
            asm volatile("" ::: "memory");
            ld_qkv(R, Ln, gw, lane);
            ld_w13(R, a.slow[l], gw, lane);       // w1 | w3 (72 registers in fp32) are resident from here to phase C only
        }
        // ---- B1: attention of (head, key slice) for both tokens ----
        // q0 q1 k0 k1 v0 v1 of this head: 6 x 64 of the 4608 published values
        gather16<1>(rs, 192, ep, [&](int i) { const int sg = i >> 5; return G_QKV + (sg & 1) * I + (sg >> 1) * D + hd * 64 + 2 * (i & 31); },
                    [&](int i) { return 2 * i; }, qkh, a.fail, 2);
        MARK_W();
        __syncthreads();
        ++ep;
        {   // RoPE of q0 q1 k0 k1 (dual_ar_stream.py:1062-1079): thread = (row tid >> 6, dim tid & 63)
            const float x = qkh[tid], xp = qkh[tid ^ 1];
            qkr[tid] = (tid & 1) ? x * rope_c + xp * rope_s : x * rope_c - xp * rope_s;
        }
        __syncthreads();
        {
            float4 q0 = *reinterpret_cast<const float4*>(qkr + li * 4), q1 = *reinterpret_cast<const float4*>(qkr + 64 + li * 4);
            q0.x *= 0.125f; q0.y *= 0.125f; q0.z *= 0.125f; q0.w *= 0.125f;
            q1.x *= 0.125f; q1.y *= 0.125f; q1.z *= 0.125f; q1.w *= 0.125f;
            float mr0 = -INFINITY, ls0 = 0.f, mr1 = -INFINITY, ls1 = 0.f;
            float4 oa0 = make_float4(0.f, 0.f, 0.f, 0.f), oa1 = oa0;
            auto step = [&](const float4& kk, const float4& vv, bool row0) {
                float s0 = q0.x * kk.x + q0.y * kk.y + q0.z * kk.z + q0.w * kk.w;
                float s1 = q1.x * kk.x + q1.y * kk.y + q1.z * kk.z + q1.w * kk.w;
                s0 = row16_sum(s0);
                s1 = row16_sum(s1);
                if (row0) {
                    const float mn = fmaxf(mr0, s0);
                    const float corr = expf(mr0 - mn), p = expf(s0 - mn);
                    ls0 = ls0 * corr + p;
                    oa0.x = oa0.x * corr + p * vv.x; oa0.y = oa0.y * corr + p * vv.y; oa0.z = oa0.z * corr + p * vv.z; oa0.w = oa0.w * corr + p * vv.w;
                    mr0 = mn;
                }
                const float mn = fmaxf(mr1, s1);
                const float corr = expf(mr1 - mn), p = expf(s1 - mn);
                ls1 = ls1 * corr + p;
                oa1.x = oa1.x * corr + p * vv.x; oa1.y = oa1.y * corr + p * vv.y; oa1.z = oa1.z * corr + p * vv.z; oa1.w = oa1.w * corr + p * vv.w;
                mr1 = mn;
            };
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lo + i < hc) step(pkk[i], pvv[i], true);
            for (int tk = lo + 4; tk < hc; ++tk) {                  // contexts beyond 1024 positions
                const float4 kk = ld_kv4<KVT>(kc + (long)tk * 64), vv = ld_kv4<KVT>(vc + (long)tk * 64);
                step(kk, vv, true);
            }
            for (int tk = lo > p0 ? lo : p0; tk < hi; ++tk) {       // the two keys of this launch: from the gathered rows
                const int m = tk - p0;
                step(*reinterpret_cast<const float4*>(qkr + (2 + m) * 64 + li * 4), *reinterpret_cast<const float4*>(qkh + (4 + m) * 64 + li * 4), m == 0);
            }
            float* pg0 = scr + (wave * 4 + grp) * 68;
            float* pg1 = scr + (16 + wave * 4 + grp) * 68;
            *reinterpret_cast<float4*>(pg0 + li * 4) = oa0;
            *reinterpret_cast<float4*>(pg1 + li * 4) = oa1;
            if (li == 0) { pg0[64] = mr0; pg0[65] = ls0; pg1[64] = mr1; pg1[65] = ls1; }
            if (sl == 0) {      // KV write of the two new positions of this head (forward_generate: kv_cache.update)
                const int which = tid >> 7, m = (tid >> 6) & 1, d = tid & 63;
                const float v = which ? qkh[(4 + m) * 64 + d] : qkr[(2 + m) * 64 + d];
                st_kv<KVT>(kl + ((long)which * H + hd) * SH + (long)(p0 + m) * 64 + d, v);
            }
        }
        __syncthreads();
        MARK_W();
        if (tid < 132) {
            const int r = tid >= 66, i = tid - 66 * r;
            const float* sp = scr + r * 16 * 68;
            float M = -INFINITY;
#pragma unroll
            for (int g2 = 0; g2 < 16; ++g2)
                if (sp[g2 * 68 + 65] > 0.f) M = fmaxf(M, sp[g2 * 68 + 64]);
            float val = 0.f, den = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 16; ++g2) {
                const float lg2 = sp[g2 * 68 + 65];
                const float wgt = lg2 > 0.f ? expf(sp[g2 * 68 + 64] - M) : 0.f;
                den = fmaf(wgt, lg2, den);
                if (i < 64) val = fmaf(wgt, sp[g2 * 68 + i], val);
            }
            store_granule(gatt + ((long)wg * 2 + r) * 66 + i, ep, i < 64 ? val : (i == 64 ? M : den));
        }
        // ---- B1m: workgroup (head, r < 2) merges the 16 key slices of token r and publishes that head's output ----
        if (sl < 2)      // the 16 key-slice partials [66] of (token sl, head hd)
            gather16<3>(rs, 16 * 33, ep, [&](int i) { const int j = i / 33; return G_ATT + ((hd * 16 + j) * 2 + sl) * 66 + 2 * (i - j * 33); },
                        [&](int i) { return 2 * i; }, attp, a.fail, 3, a.slow[l].ffn_norm, nrmC);
        else
            gather16<1>(rs, 0, ep, [&](int i) { return 0; }, [&](int i) { return 0; }, attp, a.fail, 3, a.slow[l].ffn_norm, nrmC);
        MARK_W();
        __syncthreads();
        ++ep;
        if (sl < 2 && tid < 64) {
            float M = -INFINITY;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq)
                if (attp[qq * 66 + 65] > 0.f) M = fmaxf(M, attp[qq * 66 + 64]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
                const float lq = attp[qq * 66 + 65];
                const float wgt = lq > 0.f ? expf(attp[qq * 66 + 64] - M) : 0.f;
                den = fmaf(wgt, lq, den);
                num = fmaf(wgt, attp[qq * 66 + tid], num);
            }
            store_granule(ga + sl * D + hd * 64 + tid, ep, num / den);
        }
        MARK_W();
        // ---- B2: wo + residual ----
        gather_lin<3>(rs, G_A, D, ep, av, a.fail, 4);
        MARK_W();
        __syncthreads();
        ++ep;
        {
            float o[2][1];
            gemv<WT, D, 1, 2, false>(R.wo, av, D, nullptr, 0.f, lane, o);
            MARK_W();
            if (lane < 2) store_granule(gxB + lane * D + gw, ep, xsA[lane * D + gw] + (lane ? o[1][0] : o[0][0]));
            ld_wo(R, Ln, gw, lane);
        }
        // ---- C: RMSNorm + w1 | w3 + SwiGLU ----
        gather_lin<3>(rs, G_XB, D, ep, xsB, a.fail, 5);
        MARK_W();
        __syncthreads();
        ++ep;
        {
            float o[2][6];
            gemv<WT, D, 6, 2, true>(R.w13, xsB, D, nrmC, 1e-5f, lane, o);
            MARK_W();
            float mine = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (lane == m * 3 + r) mine = silu_f(o[m][r]) * o[m][3 + r];
            if (lane < 6) {
                const int m = lane / 3, r = lane - m * 3;
                store_granule(gh + m * I + 3 * gw + r, ep, mine);
            }
            ld_w2(R, a.slow[l], gw, lane);        // one phase ahead: w1 | w3 are consumed, their registers carry w2 now
        }
        // ---- D: w2 + residual ----
        gather_lin<9>(rs, G_H, I, ep, big, a.fail, 6, Ln.attn_norm, nrmA);
        MARK_W();
        __syncthreads();
        ++ep;
        {
            float o[2][1];
            gemv<WT, I, 1, 2, false>(R.w2, big, I, nullptr, 0.f, lane, o);
            MARK_W();
            if (lane < 2) store_granule(gxA + lane * D + gw, ep, xsB[lane * D + gw] + (lane ? o[1][0] : o[0][0]));
        }
    }

    // ---------------------------------------- fast AR: 8 codebooks x 4 layers on M = 1 row ----------------------------------------
    float* const kvp = a.kv_fast + (long)wg * FAST_KV_WG;       // this workgroup's own fast K / V rows
    WFrag<WT, D> wH[2];                        // codebook head rows gw, gw + 768: resident for the frame
    {
        const int V = a.codebook_size;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = gw + NWV * j;
            if (row > V - 1) row = V - 1;
            wH[j].load(a.fast_out_w, row, lane);
        }
        asm volatile("" ::: "memory");
    }
    for (int cb = 0; cb < NCB; ++cb) {
        const int V = a.codebook_size;
        for (int l = 0; l < AR_FAST_LAYERS; ++l) {
            const bool last = cb == NCB - 1 && l == AR_FAST_LAYERS - 1;
            const ArLayerW& Ln = a.fast[(l + 1) & (AR_FAST_LAYERS - 1)];
            float* const kvl = kvp + (long)l * 7 * 2 * D;
            // ---- FA: RMSNorm + wqkv (position = codebook index);  l == 0: the input is the hidden state (cb == 0) or the embedding
            //      of the code sampled here from the previous codebook's logits ----
            if (l > 0) gather_lin<2>(rs, G_XA, D / 2, ep, xc, a.fail, 7);
            else if (cb > 0) gather_lin<2>(rs, G_LOG, V / 2, ep, lg, a.fail, 8);
            else gather_lin<2>(rs, G_XA + D, D / 2, ep, xc, a.fail, 9);
            MARK_W();
            __syncthreads();
            ++ep;
            {
                const float* xin = xc;
                if (l == 0 && cb > 0) {
                    // nucleus sample (dual_ar_stream.py:1092-1132), redundantly in every compute wave: 16 logits per lane, no barrier
                    float lv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) lv[r] = (lane + 64 * r) < V ? lg[lane + 64 * r] : -INFINITY;
                    const int raw = nucleus_sample<1, 16>(lv, V, lane, s_noise ? s_noise + a.vocab + (long)(cb - 1) * V : nullptr, seed, frame, 1, (cb - 1) * V,
                                                          a.inv_temp, a.top_p, nullptr);
                    int tk = raw;
                    if (use_forced) tk = s_forced[(long)(cb - 1) * a.chunk + a.ci];
                    if (tid == 0) toks[cb - 1] = tk;
                    if (wg == 0 && tid == 0) { s_tok_raw[cb - 1] = raw; s_tok[cb - 1] = tk; }
                    xin = a.fast_emb + (long)tk * D;
                    for (int i = tid; i < D; i += 256) xc[i] = xin[i];       // residual of FB
                }
                if (l == 0 && cb == 0 && wg == 0)
                    for (int i = tid; i < D; i += 256) s_hidden[i] = xc[i];   // hidden = pre-norm state of the content token (forward_generate :340-341)
                float o[1][3];
                gemv<WT, D, 3, 1, true>(R.qkv, xin, D, nrmA, 1e-5f, lane, o);
                MARK_W();
                float mine = o[0][0];
                if (lane == 1) mine = o[0][1];
                if (lane == 2) mine = o[0][2];
                if (lane < 3) store_granule(gqkv + 3 * gw + lane, ep, mine);
                if (!last) ld_qkv(R, Ln, gw, lane);
                ld_w13(R, a.fast[l], gw, lane);
            }
            // ---- FB: attention over <= 8 positions (every workgroup, all heads: wave w takes heads 3w .. 3w + 2), wo + residual ----
            gather_lin<5>(rs, G_QKV, I / 2, ep, big, a.fail, 10, a.fast[l].ffn_norm, nrmC);
            MARK_W();
            __syncthreads();
            ++ep;
            // RoPE of q | k at position cb, in place: 768 (even, odd) pairs over 256 threads
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = 2 * (tid + 256 * j), d = e & 63;
                const float c = ropef[(cb * 32 + (d >> 1)) * 2], sn = ropef[(cb * 32 + (d >> 1)) * 2 + 1];
                const float x0 = big[e], x1 = big[e + 1];
                big[e] = x0 * c - x1 * sn;
                big[e + 1] = x1 * c + x0 * sn;
            }
            __syncthreads();
            {
                const int kg = lane >> 4, kli = lane & 15;
#pragma unroll
                for (int hh = 0; hh < 3; ++hh) {
                    const int hb = (wave * 3 + hh) * 64;
                    const float4 q4 = *reinterpret_cast<const float4*>(big + hb + 4 * kli);
                    float sc2[2];
#pragma unroll
                    for (int rnd = 0; rnd < 2; ++rnd) {
                        const int tk = kg + 4 * rnd;
                        float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (tk < cb) k4 = *reinterpret_cast<const float4*>(kvl + (long)tk * 2 * D + hb + 4 * kli);
                        else if (tk == cb) k4 = *reinterpret_cast<const float4*>(big + D + hb + 4 * kli);
                        const float dot = row16_sum(q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w) * 0.125f;
                        sc2[rnd] = tk <= cb ? dot : -INFINITY;
                    }
                    const float mx = wave_max(fmaxf(sc2[0], sc2[1]));
                    const float e0 = sc2[0] > -INFINITY ? expf(sc2[0] - mx) : 0.f, e1 = sc2[1] > -INFINITY ? expf(sc2[1] - mx) : 0.f;
                    const float inv = 16.f / wave_sum(e0 + e1);                  // every row holds its value 16 times
                    float acc = 0.f;
#pragma unroll
                    for (int tk = 0; tk < NCB; ++tk) {
                        const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (tk >> 2) ? e1 : e0), (tk & 3) * 16));
                        if (tk <= cb) {
                            const float vd = tk < cb ? kvl[(long)(tk < 7 ? tk : 0) * 2 * D + D + hb + lane] : big[2 * D + hb + lane];
                            acc = fmaf(e, vd, acc);
                        }
                    }
                    av[hb + lane] = acc * inv;
                }
                // this position's K (after RoPE) | V for the later codebook steps of the frame: the workgroup's own copy
                if (cb < NCB - 1 && tid < 2 * D / 8) {
                    const float4 x0 = *reinterpret_cast<const float4*>(big + D + 8 * tid), x1 = *reinterpret_cast<const float4*>(big + D + 8 * tid + 4);
                    *reinterpret_cast<float4*>(kvl + (long)cb * 2 * D + 8 * tid) = x0;
                    *reinterpret_cast<float4*>(kvl + (long)cb * 2 * D + 8 * tid + 4) = x1;
                }
            }
            __syncthreads();
            {
                float o[1][1];
                gemv<WT, D, 1, 1, false>(R.wo, av, D, nullptr, 0.f, lane, o);
                MARK_W();
                if (lane == 0) store_granule(gxB + gw, ep, xc[gw] + o[0][0]);
                if (!last) ld_wo(R, Ln, gw, lane);
            }
            // ---- FC ----
            gather_lin<2>(rs, G_XB, D / 2, ep, xo, a.fail, 11);
            MARK_W();
            __syncthreads();
            ++ep;
            {
                float o[1][6];
                gemv<WT, D, 6, 1, true>(R.w13, xo, D, nrmC, 1e-5f, lane, o);
                MARK_W();
                float mine = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (lane == r) mine = silu_f(o[0][r]) * o[0][3 + r];
                if (lane < 3) store_granule(gh + 3 * gw + lane, ep, mine);
                ld_w2(R, a.fast[l], gw, lane);
            }
            // ---- FD ----
            gather_lin<5>(rs, G_H, I / 2, ep, big, a.fail, 12, last ? nullptr : Ln.attn_norm, nrmA);
            MARK_W();
            __syncthreads();
            ++ep;
            {
                float o[1][1];
                gemv<WT, I, 1, 1, false>(R.w2, big, I, nullptr, 0.f, lane, o);
                MARK_W();
                if (lane == 0) store_granule(gxA + gw, ep, xo[gw] + o[0][0]);
            }
        }
        // ---- FH: fast_norm + codebook head (rows gw, gw + 768) ----
        gather_lin<2>(rs, G_XA, D / 2, ep, xc, a.fail, 13);
        MARK_W();
        __syncthreads();
        ++ep;
        {
            float o[1][2];
            gemv<WT, D, 2, 1, true>(wH, xc, D, nrmH, 1e-5f, lane, o);
            MARK_W();
            const int row = gw + NWV * lane;
            if (lane < 2 && row < V) {
                const float mine = lane ? o[0][1] : o[0][0];
                store_granule(glog + row, ep, mine);
                s_fast_logits[(long)cb * V + row] = mine;
            }
        }
    }
    // ---- the last codebook's sample ----
    gather_lin<2>(rs, G_LOG, a.codebook_size / 2, ep, lg, a.fail, 14);
    MARK_W();
    __syncthreads();
    ++ep;
    {
        const int V = a.codebook_size;
        float lv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) lv[r] = (lane + 64 * r) < V ? lg[lane + 64 * r] : -INFINITY;
        const int raw = nucleus_sample<1, 16>(lv, V, lane, s_noise ? s_noise + a.vocab + (long)(NCB - 1) * V : nullptr, seed, frame, 1, (NCB - 1) * V,
                                              a.inv_temp, a.top_p, nullptr);
        int tk = raw;
        if (use_forced) tk = s_forced[(long)(NCB - 1) * a.chunk + a.ci];
        if (tid == 0) toks[NCB - 1] = tk;
        if (wg == 0 && tid == 0) { s_tok_raw[NCB - 1] = raw; s_tok[NCB - 1] = tk; }
        MARK_W();
    }
    // workgroup 0 rewrites the frame header: only after every semantic workgroup has read it
    if (wg == 0 && !a.skip_semantic) gather_lin<1>(rs, G_ACK, NSEM / 2, ep0 + 1, lg, a.fail, 15);
    __syncthreads();

    // ======================================= frame bookkeeping =======================================
    // cached_new_audio_emb = embed(codes) (dual_ar_stream.py:834, 245-255): 4 features per workgroup, codebooks summed in order
    if (tid < D / NWG) {
        const int i = wg * (D / NWG) + tid;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NCB; ++q) acc += a.codebook_emb[((long)toks[q] + (long)q * a.codebook_size) * D + i];
        s_cached_audio_emb[i] = acc;
    }
    if (wg != 0) return;
    if (tid < NCB) {
        s_pred_hist[(long)tid * a.hist_cap + (frame & (a.hist_cap - 1))] = toks[tid];
        s_step_audio[tid * a.chunk + a.ci] = toks[tid];
    }
    if (tid == 0) {
        s_step_content[a.ci] = code;
        *s_nframes = frame + 1;
        *s_last_pos = p0 + 1;
        *s_epoch = ep;
    }
}

}  // namespace

size_t ar_decode2_granule_words() { return (size_t)G_END; }
size_t ar_decode2_kvfast_floats() { return (size_t)NWG * FAST_KV_WG; }
size_t ar_decode2_lds_bytes() { return (size_t)L_END * sizeof(float); }

// the kernel as a host-side function pointer (occupancy queries)
const void* ar_decode2_func(int wt_half, int kv_half) {
    if (wt_half && kv_half) return (const void*)ar_decode2_kernel<__half, __half>;
    if (wt_half) return (const void*)ar_decode2_kernel<__half, float>;
    if (kv_half) return (const void*)ar_decode2_kernel<float, __half>;
    return (const void*)ar_decode2_kernel<float, float>;
}

int launch_ar_decode2(const ArDecodeArgs& a, int wt_half, int kv_half, hipStream_t st) {
    SVA_CHECK(a.vocab <= 8192 && a.vocab <= 32 * NT && (a.vocab & 1) == 0 && (a.codebook_size & 1) == 0 && a.codebook_size <= 2 * NWV && a.codebook_size <= 1024 && (a.hist_cap & (a.hist_cap - 1)) == 0,
              "ar_decode2: unsupported head sizes");
    const size_t smem = ar_decode2_lds_bytes();
    static_assert((size_t)L_END * sizeof(float) <= (size_t)64 * 1024, "LDS layout");
    static_assert(L_END >= 8192 + 256, "semantic sampler scratch");
    const dim3 grid(NWG + (a.skip_semantic ? 0 : NSEM));
    if (wt_half && kv_half) hipLaunchKernelGGL((ar_decode2_kernel<__half, __half>), grid, dim3(NT), smem, st, a);
    else if (wt_half) hipLaunchKernelGGL((ar_decode2_kernel<__half, float>), grid, dim3(NT), smem, st, a);
    else if (kv_half) hipLaunchKernelGGL((ar_decode2_kernel<float, __half>), grid, dim3(NT), smem, st, a);
    else hipLaunchKernelGGL((ar_decode2_kernel<float, float>), grid, dim3(NT), smem, st, a);
    SVA_HIP(hipGetLastError());
    return 0;
}

}  // namespace sva
