// Persistent batch-1 decode kernel of the dual AR: ONE launch per frame instead of ~210 dependent launches.
//
// decode_one_token_ar (modules/dual_ar_stream.py:1168-1219) at batch 1 is a chain of ~200 tiny matrix-vector products
// (12 slow layers on 2 tokens, then 8 x [4 fast layers + codebook head + nucleus sample]); every one needs the whole
// output vector of its predecessor, so as separate kernels each costs a launch boundary plus a cold start of its weight
// stream (5-9 us per kernel measured, 1.5 ms per frame) although the frame only moves 0.5 GB (fp32) / 0.26 GB (fp16).
// Here 96 workgroups (one per CU of the AR stream's partition) stay resident for the whole frame and hand the
// activation vectors to each other in-launch:
//   * every wave owns fixed output rows of every weight matrix and keeps the whole K extent of its rows in registers
//     (lane l holds elements 4l + 256j of an fp32 row / 8l + 512j of an fp16 row: whole-wave 1 KiB loads);
//   * the weight rows of a phase are requested BEFORE the wave waits for that phase's input vector, so the HBM latency
//     of the weight stream overlaps the hand-off of the previous phase's output (the one thing launches cannot do);
//   * hand-off = 8-byte {tag = phase epoch, value} granules written with agent-scope (sc1, write-through) stores and
//     polled with agent-scope loads until every tag matches (cdna_hip_programming.md Guideline 16, form R2: the data is
//     the flag -- no fences, no counters, nothing depends on workgroup placement).  Measured 2.7 us per all-to-all
//     edge at 96 workgroups (tools/micro/ar_edge.hip);  a buffer read in phase p is rewritten no earlier than phase
//     p + 1, when every workgroup has provably finished reading it (it published its phase-p output after the read);
//   * the nucleus sampler of a codebook runs redundantly in every wave (one wave, 16 logits per lane, sort-free
//     threshold search with DPP reductions only), so the sampled token needs no broadcast edge;
//   * fast-AR K/V of the 8 codebook positions go through a small global scratch with agent-scope stores / loads.
// All spins are bounded (a timeout sets *fail and lets the kernel run to its end with garbage instead of hanging).
#include "ar_decode.h"
#include "device_util.h"
#include "sva_common.h"

#include "ar_device.h"

namespace sva {
namespace {

using namespace ardev;
constexpr int GX = 2 * D, GBIG = 2 * I, GATT = AR_WGS * 66, GLOG = 1024, GA = 2 * D;
constexpr int SPIN_LIMIT = 1 << 16;                // polls before a gather gives up (~50 ms); a healthy edge takes a handful

// all 256 threads: wait for the n granules (n even, g 16-byte aligned) of a published vector (tags == ep) and unpack them into LDS.
// Polled two granules at a time with 16-byte sc1 buffer loads, each granule still validated by its own tag: half the load
// instructions per pass of the 8-byte form (tools/micro/ar_edge2.hip: 2.81 -> 2.62 us per all-to-all phase, 3.06 -> 2.76 beside a
// weight stream).  PER = granules per thread of the 8-byte form (kept as the template argument: n <= PER * 256).
typedef int v4i __attribute__((ext_vector_type(4)));
template <int PER>
__device__ __forceinline__ void gather(const u64* g, int n, unsigned ep, float* dst, int* fail, int code) {
    constexpr int PP = (PER + 1) / 2;
    const int tid = threadIdx.x, npairs = n >> 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(g), 0, n * 8, 0x00020000);
    unsigned pending = 0;
#pragma unroll
    for (int k = 0; k < PP; ++k)
        if (tid + k * 256 < npairs) pending |= 1u << k;
    int spins = 0;
    if (*reinterpret_cast<volatile int*>(fail)) pending = 0;       // an earlier gather timed out (e.g. not all 96 workgroups resident): run through
    while (pending) {
        // the buffer-load builtin is an ordinary memory read to the optimiser (unlike the agent-scope atomic loads of the 8-byte form):
        // without a compiler barrier in the loop it is hoisted out and the loop polls a register
        asm volatile("" ::: "memory");
        v4i x[PP];
#pragma unroll
        for (int k = 0; k < PP; ++k)
            if (pending & (1u << k)) x[k] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + k * 256) * 16, 0, 16));     // aux 16 = sc1
#pragma unroll
        for (int k = 0; k < PP; ++k)
            if ((pending & (1u << k)) && (unsigned)x[k].y == ep && (unsigned)x[k].w == ep) {
                *reinterpret_cast<float2*>(dst + 2 * (tid + k * 256)) = make_float2(__int_as_float(x[k].x), __int_as_float(x[k].z));
                pending &= ~(1u << k);
            }
        if (pending && ++spins > SPIN_LIMIT) { *fail = code; break; }
    }
    // Every vector-memory operation this wave issued before the polls -- weight prefetches, and the relaxed write-through stores of
    // fast K / V rows and semantic logits of the PREVIOUS phase -- has completed here: loads return in order, the wait on the last
    // poll is a wait on everything older, made explicit for the stores.  A later reader of those rows is ordered behind this wave's
    // NEXT publish (it gathers that vector), hence behind the completed stores: no tag or fence on the rows themselves is needed.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// SVA_AR_TIMING=1: workgroup 0 records wall_clock64() (100 MHz) at every phase boundary -- [2k] = input gathered, [2k + 1] = output published
#define AR_MARK() do { if (s_dbg && wg == 0 && tid == 0) { s_dbg[nmark] = wall_clock64(); } ++nmark; } while (0)

template <typename WT, typename KVT>
__global__ __launch_bounds__(256, 1) void ar_decode_kernel(const ArDecodeArgs a) {
    // slot of this workgroup row: the per-stream pointers, advanced by y strides (uniform: scalar registers; the argument block itself
    // stays in the kernarg segment -- a modified private copy of it would be indexed through scratch)
    const long y = (long)blockIdx.y + a.slot_base;
    const ArDecodeArgs::SlotStride& t = a.ss;
    const long long* const s_codes = a.codes + y * t.codes;
    float* const s_cached_audio_emb = a.cached_audio_emb + y * t.emb;
    int* const s_last_pos = a.last_pos + y;
    int* const s_nframes = a.nframes + y;
    const unsigned long long* const s_seed = a.seed + y;
    KVT* const s_kv_slow = reinterpret_cast<KVT*>(a.kv_slow) + y * t.kv_slot;
    float* const s_kv_fast = a.kv_fast + y * t.kv_fast;
    unsigned long long* const s_gx = a.gx + y * t.gran;
    unsigned long long* const s_gbig = a.gbig + y * t.gran;
    unsigned long long* const s_gatt = a.gatt + y * t.gran;
    unsigned long long* const s_glog = a.glog + y * t.gran;
    unsigned long long* const s_ga = a.ga + y * t.gran;
    unsigned* const s_epoch = a.epoch + y;
    long long* const s_dbg = y ? nullptr : a.dbg;
    float* const s_slow_logits = a.slow_logits + y * t.slow_logits;
    float* const s_fast_logits = a.fast_logits + y * t.fast_logits;
    float* const s_hidden = a.hidden + y * t.hidden;
    int* const s_sem = a.sem + y;
    int* const s_tok_raw = a.tok_raw + y * t.tok;
    int* const s_tok = a.tok + y * t.tok;
    int* const s_step_audio = a.step_audio + y * t.step_audio;
    int* const s_pred_hist = a.pred_hist + y * t.pred_hist;
    int* const s_step_content = a.step_content + y * t.step_content;
    const float* const s_noise = a.noise ? a.noise + y * t.noise : nullptr;
    const int* const s_forced = a.forced + y * t.forced;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                          // [2][768] residual stream (fast AR: row 0)
    float* big = xs + GX;                     // [2][2304] qkv / SwiGLU output
    float* attp = big + GBIG;                 // [4][66] split-key attention partials of one (row, head)
    float* av = attp + 4 * 68;                // [2][768] attention output
    float* lg = av + GX;                      // [1024] codebook logits
    float* scr = lg + GLOG;                   // [16][68] group partials | sampler scratch | scores
    float* ropef = scr + 16 * 68;             // [8][32][2] RoPE table of the codebook positions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, gw = wg * 4 + wave;
    // this kernel is a latency chain that shares its CUs with the encoder's / vocoder's throughput kernels: its waves go first
    __builtin_amdgcn_s_setprio(3);
    unsigned ep = *s_epoch;
    int nmark = 0;
    const int p0 = *s_last_pos + 1;           // positions of the two new tokens (dual_ar_stream.py:821-824)
    const int frame = *s_nframes;
    const unsigned long long seed = *s_seed;
    const int code = (int)s_codes[a.code_off];
    const int use_forced = *a.use_forced;
    KVT* kv = s_kv_slow;
    const long SH = (long)a.S * 64;           // one head of the cache

    // RoPE factors of this wave's three (even, odd) pairs: the two slow positions in registers, the 8 codebook positions in LDS
    float rc[2][3], rsn[2][3];
    {
        const int n0 = 6 * gw;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
                const int d = (n0 + 2 * pr) & 63;
                rc[m][pr] = a.rope_slow[((long)(p0 + m) * 32 + (d >> 1)) * 2];
                rsn[m][pr] = a.rope_slow[((long)(p0 + m) * 32 + (d >> 1)) * 2 + 1];
            }
        for (int i = tid; i < NCB * 64; i += 256) ropef[i] = a.rope_fast[i];
    }
    // tokens [cached_new_audio_emb, src_cond] (decode_one, :817-837)
    for (int i = tid; i < D; i += 256) {
        xs[i] = s_cached_audio_emb[i];
        xs[D + i] = a.content_emb[(long)code * D + i];
    }
    __syncthreads();

    // ======================================= slow AR: 12 layers on M = 2 rows =======================================
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const ArLayerW& L = a.slow[l];
        KVT* kl = kv + (long)l * a.kv_layer_stride;
        {   // ---- A: RMSNorm + wqkv + RoPE + KV write ----
            WFrag<WT, D> w[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) w[r].load(L.wqkv, 6L * gw + r, lane);
            asm volatile("" ::: "memory");
            if (l > 0) gather<6>(s_gx, GX, ep, xs, a.fail, 1);
            AR_MARK();
            float o[2][6];
            gemv<WT, D, 6, 2, true>(w, xs, D, L.attn_norm, 1e-5f, lane, o);
            const int n0 = 6 * gw, region = gw >> 7;               // 0 q, 1 k, 2 v  (128 waves each)
            if (region < 2) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        const float c = rc[m][pr], sn = rsn[m][pr];
                        const float x0 = o[m][2 * pr], x1 = o[m][2 * pr + 1];
                        o[m][2 * pr] = x0 * c - x1 * sn;
                        o[m][2 * pr + 1] = x1 * c + x0 * sn;
                    }
            }
            AR_MARK();
            ++ep;
            float mine = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (lane == m * 6 + r) mine = o[m][r];
            if (lane < 12) {
                const int m = lane / 6, r = lane - m * 6;
                store_granule(s_gbig + m * I + n0 + r, ep, mine);
                if (region >= 1) {
                    const int nn = n0 + r - D * region, h = nn >> 6, d = nn & 63;
                    st_kv<KVT>(kl + ((long)(region - 1) * H + h) * SH + (long)(p0 + m) * 64 + d, mine);
                }
            }
        }
        {   // ---- B1: attention of (row, head, quarter of the keys) per workgroup ----
            const int r = wg / 48, h = (wg % 48) >> 2, qtr = wg & 3;
            const int Lk = p0 + r + 1;                              // keys 0 .. p_r
            const int seg = qtr * 4 + wave;
            const int lo = (int)((long)seg * Lk / 16), hi = (int)((long)(seg + 1) * Lk / 16);
            const int grp = lane >> 4, li = lane & 15;
            const KVT* kc = kl + (long)h * SH + li * 4;
            const KVT* vc = kc + (long)H * SH;
            const int hc = hi < p0 ? hi : p0;                       // cached keys of this segment: [lo, hc)
            // the cached K / V rows do not depend on this step: request the first 8 keys of every 16-lane group before waiting
            // for q (covers contexts up to 512 positions entirely)
            float4 pkk[8], pvv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int t = lo + grp + 4 * i;
                if (t > hc - 1) t = hc - 1;
                if (t < 0) t = 0;
                pkk[i] = ld_kv4<KVT>(kc + (long)t * 64);
                pvv[i] = ld_kv4<KVT>(vc + (long)t * 64);
            }
            asm volatile("" ::: "memory");
            gather<18>(s_gbig, GBIG, ep, big, a.fail, 2);
            AR_MARK();
            float4 q = *reinterpret_cast<const float4*>(big + r * I + h * 64 + li * 4);
            q.x *= 0.125f; q.y *= 0.125f; q.z *= 0.125f; q.w *= 0.125f;
            float mrun = -INFINITY, lsum = 0.f;
            float4 oacc = make_float4(0.f, 0.f, 0.f, 0.f);
            auto step = [&](const float4& kk, const float4& vv) {
                float s = q.x * kk.x + q.y * kk.y + q.z * kk.z + q.w * kk.w;
                s = row16_sum(s);
                const float mn = fmaxf(mrun, s);
                const float corr = expf(mrun - mn), p = expf(s - mn);
                lsum = lsum * corr + p;
                oacc.x = oacc.x * corr + p * vv.x; oacc.y = oacc.y * corr + p * vv.y;
                oacc.z = oacc.z * corr + p * vv.z; oacc.w = oacc.w * corr + p * vv.w;
                mrun = mn;
            };
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (lo + grp + 4 * i < hc) step(pkk[i], pvv[i]);
            int t = lo + grp + 32;
#pragma unroll 4
            for (; t < hc; t += 4) {
                const float4 kk = ld_kv4<KVT>(kc + (long)t * 64), vv = ld_kv4<KVT>(vc + (long)t * 64);
                step(kk, vv);
            }
            t = lo + grp;
            while (t < hc) t += 4;                                  // first key of this group at or beyond the cached range
            for (; t < hi; t += 4) {                                // the one or two keys written in this launch: from the gathered rows
                const float* cur = big + (t - p0) * I + h * 64 + li * 4;
                step(*reinterpret_cast<const float4*>(cur + D), *reinterpret_cast<const float4*>(cur + 2 * D));
            }
            float* pg = scr + (wave * 4 + grp) * 68;
            *reinterpret_cast<float4*>(pg + li * 4) = oacc;
            if (li == 0) { pg[64] = mrun; pg[65] = lsum; }
            __syncthreads();
            AR_MARK();
            ++ep;
            if (tid < 66) {
                float M = -INFINITY;
#pragma unroll
                for (int g2 = 0; g2 < 16; ++g2)
                    if (scr[g2 * 68 + 65] > 0.f) M = fmaxf(M, scr[g2 * 68 + 64]);
                float val = 0.f, den = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < 16; ++g2) {
                    const float lg2 = scr[g2 * 68 + 65];
                    const float wgt = lg2 > 0.f ? expf(scr[g2 * 68 + 64] - M) : 0.f;
                    den = fmaf(wgt, lg2, den);
                    if (tid < 64) val = fmaf(wgt, scr[g2 * 68 + tid], val);
                }
                store_granule(s_gatt + wg * 66 + tid, ep, tid < 64 ? val : (tid == 64 ? M : den));
            }
        }
        {   // ---- B1m: the first workgroup of every (row, head) merges its four key quarters and publishes that head's output ----
            // (two small edges instead of one 6336-granule gather in every workgroup)
            if ((wg & 3) == 0) {
                gather<2>(s_gatt + wg * 66, 4 * 66, ep, attp, a.fail, 3);
                AR_MARK();
                if (tid < 64) {
                    float M = -INFINITY;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        if (attp[qq * 66 + 65] > 0.f) M = fmaxf(M, attp[qq * 66 + 64]);
                    float num = 0.f, den = 0.f;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const float lq = attp[qq * 66 + 65];
                        const float wgt = lq > 0.f ? expf(attp[qq * 66 + 64] - M) : 0.f;
                        den = fmaf(wgt, lq, den);
                        num = fmaf(wgt, attp[qq * 66 + tid], num);
                    }
                    const int r = wg / 48, h = (wg % 48) >> 2;
                    AR_MARK();
                    store_granule(s_ga + r * D + h * 64 + tid, ep + 1, num / den);
                } else { AR_MARK(); }
            } else { AR_MARK(); AR_MARK(); }
            ++ep;
        }
        {   // ---- B2: wo + residual ----
            WFrag<WT, D> w[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) w[r].load(L.wo, 2L * gw + r, lane);
            asm volatile("" ::: "memory");
            gather<6>(s_ga, GA, ep, av, a.fail, 13);
            AR_MARK();
            float o[2][2];
            gemv<WT, D, 2, 2, false>(w, av, D, nullptr, 0.f, lane, o);
            AR_MARK();
            ++ep;
            float mine = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (lane == m * 2 + r) mine = o[m][r];
            if (lane < 4) {
                const int m = lane >> 1, n = 2 * gw + (lane & 1);
                store_granule(s_gx + m * D + n, ep, xs[m * D + n] + mine);
            }
        }
        {   // ---- C: RMSNorm + w1|w3 + SwiGLU ----
            WFrag<WT, D> w[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) w[r].load(L.w13, 12L * gw + r, lane);
            asm volatile("" ::: "memory");
            gather<6>(s_gx, GX, ep, xs, a.fail, 4);
            AR_MARK();
            float o[2][12];
            gemv<WT, D, 12, 2, true>(w, xs, D, L.ffn_norm, 1e-5f, lane, o);
            AR_MARK();
            ++ep;
            float mine = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (lane == m * 6 + r) mine = silu_f(o[m][r]) * o[m][6 + r];
            if (lane < 12) {
                const int m = lane / 6, r = lane - m * 6;
                store_granule(s_gbig + m * I + 6 * gw + r, ep, mine);
            }
        }
        {   // ---- D: w2 + residual ----
            WFrag<WT, I> w[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) w[r].load(L.w2, 2L * gw + r, lane);
            asm volatile("" ::: "memory");
            gather<18>(s_gbig, GBIG, ep, big, a.fail, 5);
            AR_MARK();
            float o[2][2];
            gemv<WT, I, 2, 2, false>(w, big, I, nullptr, 0.f, lane, o);
            AR_MARK();
            ++ep;
            float mine = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (lane == m * 2 + r) mine = o[m][r];
            if (lane < 4) {
                const int m = lane >> 1, n = 2 * gw + (lane & 1);
                store_granule(s_gx + m * D + n, ep, xs[m * D + n] + mine);
            }
        }
    }
    gather<6>(s_gx, GX, ep, xs, a.fail, 6);
    AR_MARK();
    // hidden = pre-norm state of the content token (forward_generate :340-341): tap + input of the fast AR
    if (wg == 0)
        for (int i = tid; i < D; i += 256) s_hidden[i] = xs[D + i];
    for (int i = tid; i < D; i += 256) xs[i] = xs[D + i];
    __syncthreads();
    if (!a.skip_semantic) {
        // semantic-token logits (dual_ar_stream.py:1181-1186; the sample is discarded by every caller, :833): rows gw + 384 j,
        // written through (agent scope) for the sampler that workgroup 0 runs at the end of the frame
        for (int half = 0; half < 2; ++half) {
            WFrag<WT, D> w[11];
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                int row = gw + AR_WAVES * (half * 11 + j);
                if (row > a.vocab - 1) row = a.vocab - 1;
                w[j].load(a.out_w, row, lane);
            }
            float o[1][11];
            gemv<WT, D, 11, 1, true>(w, xs, D, a.out_norm, 1e-5f, lane, o);
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < 11; ++j)
                if (lane == j) mine = o[0][j];
            const int row = gw + AR_WAVES * (half * 11 + lane);
            if (lane < 11 && row < a.vocab) __hip_atomic_store(s_slow_logits + row, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ======================================= fast AR: 8 codebooks x 4 layers on M = 1 row =======================================
    __shared__ int toks[NCB];                  // the frame's codes (every workgroup samples the same token)
    int tprev = 0;
    WFrag<WT, D> wq0[6];                       // wqkv rows of fast layer 0: requested a phase early (before the sampler of the previous codebook)
#pragma unroll
    for (int r = 0; r < 6; ++r) wq0[r].load(a.fast[0].wqkv, 6L * gw + r, lane);
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb > 0) {
            for (int i = tid; i < D; i += 256) xs[i] = a.fast_emb[(long)tprev * D + i];
            __syncthreads();
        }
        for (int l = 0; l < AR_FAST_LAYERS; ++l) {
            const ArLayerW& L = a.fast[l];
            float* kvg = s_kv_fast + (long)l * NCB * 2 * D;             // [8][k 768 | v 768]
            {   // ---- FA: RMSNorm + wqkv + RoPE (position = codebook index) ----
                WFrag<WT, D> w[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    if (l > 0) w[r].load(L.wqkv, 6L * gw + r, lane);
                    else w[r] = wq0[r];
                }
                asm volatile("" ::: "memory");
                if (l > 0) gather<3>(s_gx, D, ep, xs, a.fail, 7);
                AR_MARK();
                float o[1][6];
                gemv<WT, D, 6, 1, true>(w, xs, D, L.attn_norm, 1e-5f, lane, o);
                const int n0 = 6 * gw, region = gw >> 7;
                if (region < 2) {
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        const int d = (n0 + 2 * pr) & 63;
                        const float c = ropef[(cb * 32 + (d >> 1)) * 2], sn = ropef[(cb * 32 + (d >> 1)) * 2 + 1];
                        const float x0 = o[0][2 * pr], x1 = o[0][2 * pr + 1];
                        o[0][2 * pr] = x0 * c - x1 * sn;
                        o[0][2 * pr + 1] = x1 * c + x0 * sn;
                    }
                }
                AR_MARK();
                ++ep;
                float mine = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (lane == r) mine = o[0][r];
                if (lane < 6) {
                    store_granule(s_gbig + n0 + lane, ep, mine);
                    if (region >= 1)       // K | V of this codebook position for the later positions of this frame
                        __hip_atomic_store(kvg + (long)cb * 2 * D + (n0 + lane - D), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            {   // ---- FB: attention over <= 8 positions (every workgroup computes all heads: wave w takes heads 3w..3w+2), wo + residual ----
                WFrag<WT, D> w[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) w[r].load(L.wo, 2L * gw + r, lane);
                // K | V of the earlier positions (written through at least one whole codebook step ago).  Scores: one key per 16-lane
                // row, 4 dimensions per lane (two rounds cover the 8 positions: 2 row reductions per head instead of 8 wave reductions);
                // P.V: lane = head dimension
                const int kg = lane >> 4, kli = lane & 15;
                u64 pk[3][2][2];
                float pv[3][7];
#pragma unroll
                for (int hh = 0; hh < 3; ++hh) {
                    const int hb = (wave * 3 + hh) * 64;
#pragma unroll
                    for (int rnd = 0; rnd < 2; ++rnd) {
                        const int t = kg + 4 * rnd;
                        if (t < cb) {
                            const u64* src = reinterpret_cast<const u64*>(kvg + (long)t * 2 * D + hb + 4 * kli);
                            pk[hh][rnd][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            pk[hh][rnd][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 7; ++t)
                        if (t < cb) pv[hh][t] = __hip_atomic_load(kvg + (long)t * 2 * D + D + hb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("" ::: "memory");
                gather<9>(s_gbig, I, ep, big, a.fail, 8);
                AR_MARK();
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
                        // probability of position t: held by row t & 3 in round t >> 2
                        const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (t >> 2) ? e1 : e0), (t & 3) * 16));
                        const float vd = t < cb ? (t < 7 ? pv[hh][t < 7 ? t : 0] : 0.f) : big[2 * D + hb + lane];
                        if (t <= cb) acc = fmaf(e, vd, acc);
                    }
                    av[hb + lane] = acc * inv;
                }
                __syncthreads();
                float o[1][2];
                gemv<WT, D, 2, 1, false>(w, av, D, nullptr, 0.f, lane, o);
                AR_MARK();
                ++ep;
                if (lane < 2) {
                    const int n = 2 * gw + lane;
                    store_granule(s_gx + n, ep, xs[n] + (lane == 0 ? o[0][0] : o[0][1]));
                }
            }
            {   // ---- FC ----
                WFrag<WT, D> w[12];
#pragma unroll
                for (int r = 0; r < 12; ++r) w[r].load(L.w13, 12L * gw + r, lane);
                asm volatile("" ::: "memory");
                gather<3>(s_gx, D, ep, xs, a.fail, 9);
                AR_MARK();
                float o[1][12];
                gemv<WT, D, 12, 1, true>(w, xs, D, L.ffn_norm, 1e-5f, lane, o);
                AR_MARK();
                ++ep;
                float mine = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (lane == r) mine = silu_f(o[0][r]) * o[0][6 + r];
                if (lane < 6) store_granule(s_gbig + 6 * gw + lane, ep, mine);
            }
            {   // ---- FD ----
                WFrag<WT, I> w[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) w[r].load(L.w2, 2L * gw + r, lane);
                asm volatile("" ::: "memory");
                gather<9>(s_gbig, I, ep, big, a.fail, 10);
                AR_MARK();
                float o[1][2];
                gemv<WT, I, 2, 1, false>(w, big, I, nullptr, 0.f, lane, o);
                AR_MARK();
                ++ep;
                if (lane < 2) {
                    const int n = 2 * gw + lane;
                    store_granule(s_gx + n, ep, xs[n] + (lane == 0 ? o[0][0] : o[0][1]));
                }
            }
        }
        {   // ---- FH: fast_norm + codebook head (rows gw, gw + 384, gw + 768) ----
            WFrag<WT, D> w[3];
            const int V = a.codebook_size;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int row = gw + AR_WAVES * j;
                if (row > V - 1) row = V - 1;
                w[j].load(a.fast_out_w, row, lane);
            }
            asm volatile("" ::: "memory");
            gather<3>(s_gx, D, ep, xs, a.fail, 11);
            AR_MARK();
            float o[1][3];
            gemv<WT, D, 3, 1, true>(w, xs, D, a.fast_norm, 1e-5f, lane, o);
            AR_MARK();
            ++ep;
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (lane == j) mine = o[0][j];
            const int row = gw + AR_WAVES * lane;
            if (lane < 3 && row < V) {
                store_granule(s_glog + row, ep, mine);
                s_fast_logits[(long)cb * V + row] = mine;
            }
        }
        {   // ---- FS: nucleus sample, redundantly in every workgroup (4 waves x 4 logits per lane) ----
            const int V = a.codebook_size;
            if (cb + 1 < NCB) {
#pragma unroll
                for (int r = 0; r < 6; ++r) wq0[r].load(a.fast[0].wqkv, 6L * gw + r, lane);
                asm volatile("" ::: "memory");
            }
            gather<4>(s_glog, V, ep, lg, a.fail, 12);
            AR_MARK();
            float l[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) l[r] = (tid + 256 * r) < V ? lg[tid + 256 * r] : -INFINITY;
            const int raw = nucleus_sample<4, 4>(l, V, tid, s_noise ? s_noise + a.vocab + (long)cb * V : nullptr, seed, frame, 1, cb * V, a.inv_temp, a.top_p,
                                                 reinterpret_cast<double*>(scr), (s_dbg && wg == 0 && cb == 1) ? s_dbg + 900 : nullptr);
            if (s_dbg && wg == 0 && cb == 1 && tid == 0) s_dbg[906] = wall_clock64();
            int t = raw;
            if (use_forced) t = s_forced[(long)cb * a.chunk + a.ci];
            tprev = t;
            if (tid == 0) toks[cb] = t;
            if (wg == 0 && tid == 0) { s_tok_raw[cb] = raw; s_tok[cb] = t; }
            __syncthreads();          // lg / xs are rewritten by the next codebook step
        }
    }

    // ======================================= frame bookkeeping =======================================
    // cached_new_audio_emb = embed(codes) (dual_ar_stream.py:834, 245-255): 8 features per workgroup, codebooks summed in order
    if (tid < 8) {
        const int i = wg * 8 + tid;
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
        if (a.fail_host) {          // a timeout seen by the end of this launch reaches the host without a synchronising call (sva_step_device_on)
            const int f = *reinterpret_cast<volatile int*>(a.fail);
            if (f) *reinterpret_cast<volatile int*>(a.fail_host) = f;
        }
    }
    if (!a.skip_semantic) {
        float l[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int e = tid + 256 * r;
            l[r] = e < a.vocab ? __hip_atomic_load(s_slow_logits + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
        }
        __syncthreads();
        const int s = nucleus_sample<4, 32>(l, a.vocab, tid, s_noise, seed, frame, 0, 0, a.inv_temp, a.top_p, reinterpret_cast<double*>(scr));
        if (tid == 0) *s_sem = s;
    }
}

constexpr size_t AR_LDS_FLOATS = GX + GBIG + 4 * 68 + GX + GLOG + 16 * 68 + NCB * 64;

}  // namespace

int ar_decode_occupancy(int wt_half, int kv_half, int* blocks_per_cu) {
    const void* f = wt_half ? (kv_half ? (const void*)ar_decode_kernel<__half, __half> : (const void*)ar_decode_kernel<__half, float>)
                            : (kv_half ? (const void*)ar_decode_kernel<float, __half> : (const void*)ar_decode_kernel<float, float>);
    SVA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, f, 256, AR_LDS_FLOATS * sizeof(float)));
    return 0;
}

size_t ar_decode_granule_words() { return (size_t)GX + GBIG + GATT + GLOG + GA; }

int launch_ar_decode(const ArDecodeArgs& a, int wt_half, int kv_half, bool one_per_cu, hipStream_t st, int n_slots) {
    SVA_CHECK(n_slots >= 1 && n_slots <= 8, "ar_decode: 1..8 streams per launch");
    SVA_CHECK(a.vocab <= 22 * AR_WAVES && a.codebook_size <= 3 * AR_WAVES && a.codebook_size <= 1024 && a.codebook_size % 2 == 0 && (a.hist_cap & (a.hist_cap - 1)) == 0,
              "ar_decode: unsupported head sizes");
    // one_per_cu: ask for more than half of a CU's LDS so that the 96 workgroups land on 96 different CUs (the AR stream's own
    // partition: every CU's load bandwidth counts); otherwise the small footprint lets other kernels share the CUs
    const size_t smem_max = (size_t)88 * 1024;
    const size_t smem = one_per_cu ? smem_max : AR_LDS_FLOATS * sizeof(float);
    static_assert(AR_LDS_FLOATS * sizeof(float) <= (size_t)88 * 1024, "LDS layout");
    static DeviceOnce attr;
    if (attr.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)ar_decode_kernel<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
        SVA_HIP(hipFuncSetAttribute((const void*)ar_decode_kernel<__half, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
        SVA_HIP(hipFuncSetAttribute((const void*)ar_decode_kernel<__half, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
        SVA_HIP(hipFuncSetAttribute((const void*)ar_decode_kernel<float, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
        attr.done();
    }
    if (wt_half && kv_half) hipLaunchKernelGGL((ar_decode_kernel<__half, __half>), dim3(AR_WGS, n_slots), dim3(256), smem, st, a);
    else if (wt_half) hipLaunchKernelGGL((ar_decode_kernel<__half, float>), dim3(AR_WGS, n_slots), dim3(256), smem, st, a);
    else if (kv_half) hipLaunchKernelGGL((ar_decode_kernel<float, __half>), dim3(AR_WGS, n_slots), dim3(256), smem, st, a);
    else hipLaunchKernelGGL((ar_decode_kernel<float, float>), dim3(AR_WGS, n_slots), dim3(256), smem, st, a);
    SVA_HIP(hipGetLastError());
    return 0;
}

}  // namespace sva
