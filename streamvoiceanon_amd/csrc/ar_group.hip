// Persistent decode kernel of the dual AR for a GROUP of 2-4 streams: the phase list, the ownership of weight rows and the
// granule hand-offs of ar_decode.hip (decode_one_token_ar, modules/dual_ar_stream.py:1168-1219), with every phase's weight rows
// loaded into registers ONCE and applied to the rows of all NS streams of the group.
//
// Why: ar_decode.hip gives every stream its own 96 workgroups.  Two streams in one launch occupy 192 CUs with workgroups that
// mostly poll, and the encoder / vocoder of the pipelined mode are left with 64 CUs (2 streams: 1.53 ms per step against 1.02 for
// one); three streams are two serial launches.  A frame is 209 dependent hand-offs of ~2.6 us with a few hundred nanoseconds of
// arithmetic each, so a second and third stream riding the SAME hand-offs cost little: the group's 96 workgroups pay the edges
// once, stream the weights once and leave 160 CUs to the other stages.
//
// MEASURED (profiles/r04_group_ab.txt) and NOT the default: the frame of a group costs 0.85 ms + ~0.5 ms per further stream (2 / 3 / 4
// streams: AR stage 1.35 / 1.86 / 2.44 ms; pipelined 1358 / 1498 / 1554 frames/s against 1311 / 1538 / 1688 for the default policy), i.e.
// a further stream costs nearly what it costs alone.  Polling all streams in one sweep and publishing all streams' results back to
// back took 0.16 ms off the two-stream frame; what remains is neither the arithmetic (~0.1 ms per stream) nor the samplers (0.07 ms):
// every workgroup gathers every activation vector of every stream through agent-scope loads -- ~300 MB per stream and frame through
// the fabric (96 workgroups x 1.5-37 KB x 209 phases) -- so the all-gather, not the hand-off latency, is what a stream costs, and
// sharing the hand-offs does not share it.  Kept behind SVA_DEBUG ar_group=1 with a parity test; the default policy is unchanged.
//
// Per stream the arithmetic is EXACTLY that of ar_decode.hip (same gemv over the same lane-owned weight slices, same attention
// partition, same sampler on the same noise), so a stream's codes do not depend on which kernel or which neighbours it ran with.
// Every stream keeps its own granule buffers and epoch; the streams of a group advance through the phases together.
#include "ar_decode.h"
#include "device_util.h"
#include "sva_common.h"

#include "ar_device.h"

namespace sva {
namespace {

using namespace ardev;
constexpr int GX = 2 * D, GBIG = 2 * I, GATT = AR_WGS * 66, GLOG = 1024, GA = 2 * D;
constexpr int SPIN_LIMIT = 1 << 16;
constexpr int PS = GX + GBIG + GX + GLOG;          // LDS floats per stream: xs | big | av | lg

typedef int v4i __attribute__((ext_vector_type(4)));

// all 256 threads: poll the n granules (n even) of the published vectors of ALL NS streams until every tag equals its stream's epoch,
// unpacking into LDS (ar_decode.hip: gather).  The loads of every stream go out in the same sweep: polled one stream after the other
// (the first version of this kernel) each stream cost its own memory round trip per phase -- 0.66 ms per extra stream and frame.
template <int PER, int NS, typename SrcF, typename DstF>
__device__ __forceinline__ void poll_n(SrcF&& src, int n, const unsigned (&ep)[NS], DstF&& dst, int* fail, int code) {
    constexpr int PP = (PER + 1) / 2;
    const int tid = threadIdx.x, npairs = n >> 1;
    __amdgpu_buffer_rsrc_t rs[NS];
    unsigned pending[NS];
    unsigned init = 0;
#pragma unroll
    for (int k = 0; k < PP; ++k)
        if (tid + k * 256 < npairs) init |= 1u << k;
    if (*reinterpret_cast<volatile int*>(fail)) init = 0;          // an earlier wait timed out: run through
    unsigned any = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        rs[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(src(s)), 0, n * 8, 0x00020000);
        pending[s] = init;
        any |= init;
    }
    int spins = 0;
    while (any) {
        asm volatile("" ::: "memory");
        v4i x[NS][PP];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < PP; ++k)
                if (pending[s] & (1u << k)) x[s][k] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs[s], (tid + k * 256) * 16, 0, 16));     // aux 16 = sc1
        any = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float* d = dst(s);
#pragma unroll
            for (int k = 0; k < PP; ++k)
                if ((pending[s] & (1u << k)) && (unsigned)x[s][k].y == ep[s] && (unsigned)x[s][k].w == ep[s]) {
                    *reinterpret_cast<float2*>(d + 2 * (tid + k * 256)) = make_float2(__int_as_float(x[s][k].x), __int_as_float(x[s][k].z));
                    pending[s] &= ~(1u << k);
                }
            any |= pending[s];
        }
        if (any && ++spins > SPIN_LIMIT) { *fail = code; break; }
    }
}
__device__ __forceinline__ void drain_and_sync() {
    // (ar_decode.hip: the explicit drain orders this wave's relaxed write-through stores of the previous phase before its next publish)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// per-stream pointers of one slot (uniform values)
template <typename KVT>
struct Slot {
    const long long* codes;
    float* cached_audio_emb;
    int *last_pos, *nframes;
    KVT* kv_slow;
    float* kv_fast;
    u64 *gx, *gbig, *gatt, *glog, *ga;
    unsigned* epoch;
    float *slow_logits, *fast_logits, *hidden;
    int *sem, *tok_raw, *tok, *step_audio, *pred_hist, *step_content;
    const float* noise;
    const int* forced;
};

template <typename WT, typename KVT, int NS>
__global__ __launch_bounds__(256, 1) void ar_group_kernel(const ArDecodeArgs a) {
    const ArDecodeArgs::SlotStride& t = a.ss;
    Slot<KVT> sp[NS];
    int p0[NS], frame[NS], code[NS];
    unsigned long long seed[NS];
    unsigned ep[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const long y = (long)blockIdx.y * NS + s + a.slot_base;
        Slot<KVT>& q = sp[s];
        q.codes = a.codes + y * t.codes;
        q.cached_audio_emb = a.cached_audio_emb + y * t.emb;
        q.last_pos = a.last_pos + y; q.nframes = a.nframes + y;
        q.kv_slow = reinterpret_cast<KVT*>(a.kv_slow) + y * t.kv_slot;
        q.kv_fast = a.kv_fast + y * t.kv_fast;
        q.gx = a.gx + y * t.gran; q.gbig = a.gbig + y * t.gran; q.gatt = a.gatt + y * t.gran; q.glog = a.glog + y * t.gran; q.ga = a.ga + y * t.gran;
        q.epoch = a.epoch + y;
        q.slow_logits = a.slow_logits + y * t.slow_logits; q.fast_logits = a.fast_logits + y * t.fast_logits; q.hidden = a.hidden + y * t.hidden;
        q.sem = a.sem + y; q.tok_raw = a.tok_raw + y * t.tok; q.tok = a.tok + y * t.tok;
        q.step_audio = a.step_audio + y * t.step_audio; q.pred_hist = a.pred_hist + y * t.pred_hist; q.step_content = a.step_content + y * t.step_content;
        q.noise = a.noise ? a.noise + y * t.noise : nullptr;
        q.forced = a.forced + y * t.forced;
        p0[s] = *q.last_pos + 1;           // positions of the two new tokens (dual_ar_stream.py:821-824)
        frame[s] = *q.nframes;
        seed[s] = a.seed[y];
        code[s] = (int)q.codes[a.code_off];
        ep[s] = *q.epoch;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // per stream s at lds + s * PS: xs [2][768] residual stream (fast AR: row 0) | big [2][2304] qkv / SwiGLU output | av [2][768] attention
    // output | lg [1024] codebook logits; behind them: attp [NS][4][68], scr [NS][16][68], ropef [8][32][2]
    float* const attp = lds + NS * PS;               // [NS][4][68]
    float* const scr = attp + NS * 4 * 68;           // [NS][16][68] (samplers: the first stream's block)
    float* const ropef = scr + NS * 16 * 68;
#define XS(s) (lds + (s) * PS)
#define BIG(s) (lds + (s) * PS + GX)
#define AV(s) (lds + (s) * PS + GX + GBIG)
#define LG(s) (lds + (s) * PS + GX + GBIG + GX)
#define EP_INC() do { _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) ++ep[s_]; } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, gw = wg * 4 + wave;
    __builtin_amdgcn_s_setprio(3);
    const int use_forced = *a.use_forced;
    const long SH = (long)a.S * 64;           // one head of the cache

    // RoPE factors of this wave's three (even, odd) pairs at the two slow positions of every stream; the 8 codebook positions in LDS
    float rc[NS][2][3], rsn[NS][2][3];
    {
        const int n0 = 6 * gw;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    const int d = (n0 + 2 * pr) & 63;
                    rc[s][m][pr] = a.rope_slow[((long)(p0[s] + m) * 32 + (d >> 1)) * 2];
                    rsn[s][m][pr] = a.rope_slow[((long)(p0[s] + m) * 32 + (d >> 1)) * 2 + 1];
                }
        for (int i = tid; i < NCB * 64; i += 256) ropef[i] = a.rope_fast[i];
    }
    // tokens [cached_new_audio_emb, src_cond] (decode_one, :817-837)
#pragma unroll
    for (int s = 0; s < NS; ++s)
        for (int i = tid; i < D; i += 256) {
            XS(s)[i] = sp[s].cached_audio_emb[i];
            XS(s)[D + i] = a.content_emb[(long)code[s] * D + i];
        }
    __syncthreads();

    // ======================================= slow AR: 12 layers on 2 rows per stream =======================================
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const ArLayerW& L = a.slow[l];
        {   // ---- A: RMSNorm + wqkv + RoPE + KV write ----
            WFrag<WT, D> w[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) w[r].load(L.wqkv, 6L * gw + r, lane);
            asm volatile("" ::: "memory");
            if (l > 0) {
                poll_n<6, NS>([&](int s) { return sp[s].gx; }, GX, ep, [&](int s) { return XS(s); }, a.fail, 1);
                drain_and_sync();
            }
            EP_INC();
            const int n0 = 6 * gw, region = gw >> 7;               // 0 q, 1 k, 2 v  (128 waves each)
            // every stream's result first, then all the stores back to back: the consumers sweep all streams of a vector at once, and a
            // stream published a gemv later than the first costs them a whole extra sweep (a memory round trip) per phase
            float mine_[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float o[2][6];
                gemv<WT, D, 6, 2, true>(w, XS(s), D, L.attn_norm, 1e-5f, lane, o);
                if (region < 2) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr) {
                            const float c = rc[s][m][pr], sn = rsn[s][m][pr];
                            const float x0 = o[m][2 * pr], x1 = o[m][2 * pr + 1];
                            o[m][2 * pr] = x0 * c - x1 * sn;
                            o[m][2 * pr + 1] = x1 * c + x0 * sn;
                        }
                }
                float mine = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        if (lane == m * 6 + r) mine = o[m][r];
                mine_[s] = mine;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float mine = mine_[s];
                if (lane < 12) {
                    const int m = lane / 6, r = lane - m * 6;
                    store_granule(sp[s].gbig + m * I + n0 + r, ep[s], mine);
                    if (region >= 1) {
                        const int nn = n0 + r - D * region, h = nn >> 6, d = nn & 63;
                        KVT* kl = sp[s].kv_slow + (long)l * a.kv_layer_stride;
                        st_kv<KVT>(kl + ((long)(region - 1) * H + h) * SH + (long)(p0[s] + m) * 64 + d, mine);
                    }
                }
            }
        }
        {   // ---- B1: attention of (row, head, quarter of the keys) per workgroup, stream after stream ----
            const int r = wg / 48, h = (wg % 48) >> 2, qtr = wg & 3;
            const int seg = qtr * 4 + wave;
            const int grp = lane >> 4, li = lane & 15;
            // the cached K / V rows do not depend on this step: the first 8 keys of every 16-lane group of the FIRST stream are requested
            // before the wait for q; the later streams' while the stream before them is being worked on
            float4 pkk[8], pvv[8];
            auto prefetch = [&](int s) {
                const int Lk = p0[s] + r + 1;
                const int lo = (int)((long)seg * Lk / 16), hi = (int)((long)(seg + 1) * Lk / 16);
                const int hc = hi < p0[s] ? hi : p0[s];
                const KVT* kc = sp[s].kv_slow + (long)l * a.kv_layer_stride + (long)h * SH + li * 4;
                const KVT* vc = kc + (long)H * SH;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int tk = lo + grp + 4 * i;
                    if (tk > hc - 1) tk = hc - 1;
                    if (tk < 0) tk = 0;
                    pkk[i] = ld_kv4<KVT>(kc + (long)tk * 64);
                    pvv[i] = ld_kv4<KVT>(vc + (long)tk * 64);
                }
            };
            prefetch(0);
            asm volatile("" ::: "memory");
            poll_n<18, NS>([&](int s) { return sp[s].gbig; }, GBIG, ep, [&](int s) { return BIG(s); }, a.fail, 2);
            drain_and_sync();
            EP_INC();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int Lk = p0[s] + r + 1;                              // keys 0 .. p_r
                const int lo = (int)((long)seg * Lk / 16), hi = (int)((long)(seg + 1) * Lk / 16);
                const int hc = hi < p0[s] ? hi : p0[s];                    // cached keys of this segment: [lo, hc)
                const KVT* kc = sp[s].kv_slow + (long)l * a.kv_layer_stride + (long)h * SH + li * 4;
                const KVT* vc = kc + (long)H * SH;
                const float* big = BIG(s);
                float4 q = *reinterpret_cast<const float4*>(big + r * I + h * 64 + li * 4);
                q.x *= 0.125f; q.y *= 0.125f; q.z *= 0.125f; q.w *= 0.125f;
                float mrun = -INFINITY, lsum = 0.f;
                float4 oacc = make_float4(0.f, 0.f, 0.f, 0.f);
                auto step = [&](const float4& kk, const float4& vv) {
                    float sc = q.x * kk.x + q.y * kk.y + q.z * kk.z + q.w * kk.w;
                    sc = row16_sum(sc);
                    const float mn = fmaxf(mrun, sc);
                    const float corr = expf(mrun - mn), p = expf(sc - mn);
                    lsum = lsum * corr + p;
                    oacc.x = oacc.x * corr + p * vv.x; oacc.y = oacc.y * corr + p * vv.y;
                    oacc.z = oacc.z * corr + p * vv.z; oacc.w = oacc.w * corr + p * vv.w;
                    mrun = mn;
                };
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (lo + grp + 4 * i < hc) step(pkk[i], pvv[i]);
                int tk = lo + grp + 32;
#pragma unroll 4
                for (; tk < hc; tk += 4) {
                    const float4 kk = ld_kv4<KVT>(kc + (long)tk * 64), vv = ld_kv4<KVT>(vc + (long)tk * 64);
                    step(kk, vv);
                }
                if (s + 1 < NS) prefetch(s + 1);                           // (the registers are free again: in flight under the merge below)
                tk = lo + grp;
                while (tk < hc) tk += 4;                                   // first key of this group at or beyond the cached range
                for (; tk < hi; tk += 4) {                                 // the one or two keys written in this launch: from the gathered rows
                    const float* cur = big + (tk - p0[s]) * I + h * 64 + li * 4;
                    step(*reinterpret_cast<const float4*>(cur + D), *reinterpret_cast<const float4*>(cur + 2 * D));
                }
                float* pg = scr + s * 16 * 68 + (wave * 4 + grp) * 68;
                *reinterpret_cast<float4*>(pg + li * 4) = oacc;
                if (li == 0) { pg[64] = mrun; pg[65] = lsum; }
            }
            __syncthreads();
            if (tid < 66) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float* sc_ = scr + s * 16 * 68;
                    float M = -INFINITY;
#pragma unroll
                    for (int g2 = 0; g2 < 16; ++g2)
                        if (sc_[g2 * 68 + 65] > 0.f) M = fmaxf(M, sc_[g2 * 68 + 64]);
                    float val = 0.f, den = 0.f;
#pragma unroll
                    for (int g2 = 0; g2 < 16; ++g2) {
                        const float lg2 = sc_[g2 * 68 + 65];
                        const float wgt = lg2 > 0.f ? expf(sc_[g2 * 68 + 64] - M) : 0.f;
                        den = fmaf(wgt, lg2, den);
                        if (tid < 64) val = fmaf(wgt, sc_[g2 * 68 + tid], val);
                    }
                    store_granule(sp[s].gatt + wg * 66 + tid, ep[s], tid < 64 ? val : (tid == 64 ? M : den));
                }
            }
            __syncthreads();                                               // scr is rewritten by the next layer
        }
        {   // ---- B1m: the first workgroup of every (row, head) merges its four key quarters and publishes that head's output ----
            if ((wg & 3) == 0) {
                poll_n<2, NS>([&](int s) { return sp[s].gatt + wg * 66; }, 4 * 66, ep, [&](int s) { return attp + s * 4 * 68; }, a.fail, 3);
                drain_and_sync();
                if (tid < 64) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const float* at_ = attp + s * 4 * 68;
                        float M = -INFINITY;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
                            if (at_[qq * 66 + 65] > 0.f) M = fmaxf(M, at_[qq * 66 + 64]);
                        float num = 0.f, den = 0.f;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const float lq = at_[qq * 66 + 65];
                            const float wgt = lq > 0.f ? expf(at_[qq * 66 + 64] - M) : 0.f;
                            den = fmaf(wgt, lq, den);
                            num = fmaf(wgt, at_[qq * 66 + tid], num);
                        }
                        const int r = wg / 48, h = (wg % 48) >> 2;
                        store_granule(sp[s].ga + r * D + h * 64 + tid, ep[s] + 1, num / den);
                    }
                }
                __syncthreads();                                           // attp is rewritten by the next layer
            }
            EP_INC();
        }
        {   // ---- B2: wo + residual ----
            WFrag<WT, D> w[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) w[r].load(L.wo, 2L * gw + r, lane);
            asm volatile("" ::: "memory");
            poll_n<6, NS>([&](int s) { return sp[s].ga; }, GA, ep, [&](int s) { return AV(s); }, a.fail, 13);
            drain_and_sync();
            EP_INC();
            float mine_[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float o[2][2];
                gemv<WT, D, 2, 2, false>(w, AV(s), D, nullptr, 0.f, lane, o);
                float mine = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        if (lane == m * 2 + r) mine = o[m][r];
                mine_[s] = mine;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float mine = mine_[s];
                if (lane < 4) {
                    const int m = lane >> 1, n = 2 * gw + (lane & 1);
                    store_granule(sp[s].gx + m * D + n, ep[s], XS(s)[m * D + n] + mine);
                }
            }
        }
        {   // ---- C: RMSNorm + w1|w3 + SwiGLU ----
            WFrag<WT, D> w[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) w[r].load(L.w13, 12L * gw + r, lane);
            asm volatile("" ::: "memory");
            poll_n<6, NS>([&](int s) { return sp[s].gx; }, GX, ep, [&](int s) { return XS(s); }, a.fail, 4);
            drain_and_sync();
            EP_INC();
            float mine_[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float o[2][12];
                gemv<WT, D, 12, 2, true>(w, XS(s), D, L.ffn_norm, 1e-5f, lane, o);
                float mine = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        if (lane == m * 6 + r) mine = silu_f(o[m][r]) * o[m][6 + r];
                mine_[s] = mine;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float mine = mine_[s];
                if (lane < 12) {
                    const int m = lane / 6, r = lane - m * 6;
                    store_granule(sp[s].gbig + m * I + 6 * gw + r, ep[s], mine);
                }
            }
        }
        {   // ---- D: w2 + residual ----
            WFrag<WT, I> w[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) w[r].load(L.w2, 2L * gw + r, lane);
            asm volatile("" ::: "memory");
            poll_n<18, NS>([&](int s) { return sp[s].gbig; }, GBIG, ep, [&](int s) { return BIG(s); }, a.fail, 5);
            drain_and_sync();
            EP_INC();
            float mine_[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float o[2][2];
                gemv<WT, I, 2, 2, false>(w, BIG(s), I, nullptr, 0.f, lane, o);
                float mine = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        if (lane == m * 2 + r) mine = o[m][r];
                mine_[s] = mine;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float mine = mine_[s];
                if (lane < 4) {
                    const int m = lane >> 1, n = 2 * gw + (lane & 1);
                    store_granule(sp[s].gx + m * D + n, ep[s], XS(s)[m * D + n] + mine);
                }
            }
        }
    }
    poll_n<6, NS>([&](int s) { return sp[s].gx; }, GX, ep, [&](int s) { return XS(s); }, a.fail, 6);
    drain_and_sync();
    // hidden = pre-norm state of the content token (forward_generate :340-341): tap + input of the fast AR
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (wg == 0)
            for (int i = tid; i < D; i += 256) sp[s].hidden[i] = XS(s)[D + i];
        for (int i = tid; i < D; i += 256) XS(s)[i] = XS(s)[D + i];
    }
    __syncthreads();
    if (!a.skip_semantic) {
        // semantic-token logits (dual_ar_stream.py:1181-1186; the sample is discarded by every caller, :833): rows gw + 384 j
        for (int half = 0; half < 2; ++half) {
            WFrag<WT, D> w[11];
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                int row = gw + AR_WAVES * (half * 11 + j);
                if (row > a.vocab - 1) row = a.vocab - 1;
                w[j].load(a.out_w, row, lane);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float o[1][11];
                gemv<WT, D, 11, 1, true>(w, XS(s), D, a.out_norm, 1e-5f, lane, o);
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < 11; ++j)
                    if (lane == j) mine = o[0][j];
                const int row = gw + AR_WAVES * (half * 11 + lane);
                if (lane < 11 && row < a.vocab) __hip_atomic_store(sp[s].slow_logits + row, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ======================================= fast AR: 8 codebooks x 4 layers on 1 row per stream =======================================
    __shared__ int toks[NS][NCB];              // the frame's codes (every workgroup samples the same tokens)
    int tprev[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) tprev[s] = 0;
    WFrag<WT, D> wq0[6];                       // wqkv rows of fast layer 0: requested a phase early (before the samplers of the previous codebook)
#pragma unroll
    for (int r = 0; r < 6; ++r) wq0[r].load(a.fast[0].wqkv, 6L * gw + r, lane);
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb > 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                for (int i = tid; i < D; i += 256) XS(s)[i] = a.fast_emb[(long)tprev[s] * D + i];
            __syncthreads();
        }
        for (int l = 0; l < AR_FAST_LAYERS; ++l) {
            const ArLayerW& L = a.fast[l];
            {   // ---- FA: RMSNorm + wqkv + RoPE (position = codebook index) ----
                WFrag<WT, D> w[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    if (l > 0) w[r].load(L.wqkv, 6L * gw + r, lane);
                    else w[r] = wq0[r];
                }
                asm volatile("" ::: "memory");
                if (l > 0) {
                    poll_n<3, NS>([&](int s) { return sp[s].gx; }, D, ep, [&](int s) { return XS(s); }, a.fail, 7);
                    drain_and_sync();
                }
                EP_INC();
                const int n0 = 6 * gw, region = gw >> 7;
                float mine_[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float o[1][6];
                    gemv<WT, D, 6, 1, true>(w, XS(s), D, L.attn_norm, 1e-5f, lane, o);
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
                    float mine = 0.f;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        if (lane == r) mine = o[0][r];
                    mine_[s] = mine;
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float* kvg = sp[s].kv_fast + (long)l * NCB * 2 * D;             // [8][k 768 | v 768]
                    const float mine = mine_[s];
                    if (lane < 6) {
                        store_granule(sp[s].gbig + n0 + lane, ep[s], mine);
                        if (region >= 1)       // K | V of this codebook position for the later positions of this frame
                            __hip_atomic_store(kvg + (long)cb * 2 * D + (n0 + lane - D), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            {   // ---- FB: attention over <= 8 positions (every workgroup computes all heads: wave w takes heads 3w..3w+2), wo + residual ----
                WFrag<WT, D> w[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) w[r].load(L.wo, 2L * gw + r, lane);
                const int kg = lane >> 4, kli = lane & 15;
                // K | V of the earlier positions (written through at least one whole codebook step ago): the first stream's before the wait
                u64 pk[3][2][2];
                float pv[3][7];
                auto load_hist = [&](int s) {
                    const float* kvg = sp[s].kv_fast + (long)l * NCB * 2 * D;
#pragma unroll
                    for (int hh = 0; hh < 3; ++hh) {
                        const int hb = (wave * 3 + hh) * 64;
#pragma unroll
                        for (int rnd = 0; rnd < 2; ++rnd) {
                            const int tp = kg + 4 * rnd;
                            if (tp < cb) {
                                const u64* src = reinterpret_cast<const u64*>(kvg + (long)tp * 2 * D + hb + 4 * kli);
                                pk[hh][rnd][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                pk[hh][rnd][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
#pragma unroll
                        for (int tp = 0; tp < 7; ++tp)
                            if (tp < cb) pv[hh][tp] = __hip_atomic_load(kvg + (long)tp * 2 * D + D + hb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                };
                load_hist(0);
                asm volatile("" ::: "memory");
                poll_n<9, NS>([&](int s) { return sp[s].gbig; }, I, ep, [&](int s) { return BIG(s); }, a.fail, 8);
                drain_and_sync();
                EP_INC();
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float* big = BIG(s);
                    float* av = AV(s);
#pragma unroll
                    for (int hh = 0; hh < 3; ++hh) {
                        const int hb = (wave * 3 + hh) * 64;
                        const float4 q4 = *reinterpret_cast<const float4*>(big + hb + 4 * kli);
                        float sc2[2];
#pragma unroll
                        for (int rnd = 0; rnd < 2; ++rnd) {
                            const int tp = kg + 4 * rnd;
                            float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (tp < cb) {
                                k4 = make_float4(__uint_as_float((unsigned)pk[hh][rnd][0]), __uint_as_float((unsigned)(pk[hh][rnd][0] >> 32)),
                                                 __uint_as_float((unsigned)pk[hh][rnd][1]), __uint_as_float((unsigned)(pk[hh][rnd][1] >> 32)));
                            } else if (tp == cb) {
                                k4 = *reinterpret_cast<const float4*>(big + D + hb + 4 * kli);
                            }
                            const float dot = row16_sum(q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w) * 0.125f;
                            sc2[rnd] = tp <= cb ? dot : -INFINITY;
                        }
                        const float mx = wave_max(fmaxf(sc2[0], sc2[1]));
                        const float e0 = sc2[0] > -INFINITY ? expf(sc2[0] - mx) : 0.f, e1 = sc2[1] > -INFINITY ? expf(sc2[1] - mx) : 0.f;
                        const float inv = 16.f / wave_sum(e0 + e1);                  // every row holds its value 16 times
                        float acc = 0.f;
#pragma unroll
                        for (int tp = 0; tp < NCB; ++tp) {
                            // probability of position tp: held by row tp & 3 in round tp >> 2
                            const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (tp >> 2) ? e1 : e0), (tp & 3) * 16));
                            const float vd = tp < cb ? (tp < 7 ? pv[hh][tp < 7 ? tp : 0] : 0.f) : big[2 * D + hb + lane];
                            if (tp <= cb) acc = fmaf(e, vd, acc);
                        }
                        av[hb + lane] = acc * inv;
                    }
                    if (s + 1 < NS) load_hist(s + 1);
                }
                __syncthreads();
                float mine_[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float o[1][2];
                    gemv<WT, D, 2, 1, false>(w, AV(s), D, nullptr, 0.f, lane, o);
                    mine_[s] = lane == 0 ? o[0][0] : o[0][1];
                }
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (lane < 2) {
                        const int n = 2 * gw + lane;
                        store_granule(sp[s].gx + n, ep[s], XS(s)[n] + mine_[s]);
                    }
            }
            {   // ---- FC ----
                WFrag<WT, D> w[12];
#pragma unroll
                for (int r = 0; r < 12; ++r) w[r].load(L.w13, 12L * gw + r, lane);
                asm volatile("" ::: "memory");
                poll_n<3, NS>([&](int s) { return sp[s].gx; }, D, ep, [&](int s) { return XS(s); }, a.fail, 9);
                drain_and_sync();
                EP_INC();
                float mine_[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float o[1][12];
                    gemv<WT, D, 12, 1, true>(w, XS(s), D, L.ffn_norm, 1e-5f, lane, o);
                    float mine = 0.f;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        if (lane == r) mine = silu_f(o[0][r]) * o[0][6 + r];
                    mine_[s] = mine;
                }
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (lane < 6) store_granule(sp[s].gbig + 6 * gw + lane, ep[s], mine_[s]);
            }
            {   // ---- FD ----
                WFrag<WT, I> w[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) w[r].load(L.w2, 2L * gw + r, lane);
                asm volatile("" ::: "memory");
                poll_n<9, NS>([&](int s) { return sp[s].gbig; }, I, ep, [&](int s) { return BIG(s); }, a.fail, 10);
                drain_and_sync();
                EP_INC();
                float mine_[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float o[1][2];
                    gemv<WT, I, 2, 1, false>(w, BIG(s), I, nullptr, 0.f, lane, o);
                    mine_[s] = lane == 0 ? o[0][0] : o[0][1];
                }
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (lane < 2) {
                        const int n = 2 * gw + lane;
                        store_granule(sp[s].gx + n, ep[s], XS(s)[n] + mine_[s]);
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
            poll_n<3, NS>([&](int s) { return sp[s].gx; }, D, ep, [&](int s) { return XS(s); }, a.fail, 11);
            drain_and_sync();
            EP_INC();
            float mine_[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float o[1][3];
                gemv<WT, D, 3, 1, true>(w, XS(s), D, a.fast_norm, 1e-5f, lane, o);
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (lane == j) mine = o[0][j];
                mine_[s] = mine;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float mine = mine_[s];
                const int row = gw + AR_WAVES * lane;
                if (lane < 3 && row < V) {
                    store_granule(sp[s].glog + row, ep[s], mine);
                    sp[s].fast_logits[(long)cb * V + row] = mine;
                }
            }
        }
        {   // ---- FS: nucleus samples, redundantly in every workgroup (4 waves x 4 logits per lane, one stream after the other) ----
            const int V = a.codebook_size;
            if (cb + 1 < NCB) {
#pragma unroll
                for (int r = 0; r < 6; ++r) wq0[r].load(a.fast[0].wqkv, 6L * gw + r, lane);
                asm volatile("" ::: "memory");
            }
            poll_n<4, NS>([&](int s) { return sp[s].glog; }, V, ep, [&](int s) { return LG(s); }, a.fail, 12);
            drain_and_sync();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float* lg = LG(s);
                float lv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) lv[r] = (tid + 256 * r) < V ? lg[tid + 256 * r] : -INFINITY;
                const int raw = nucleus_sample<4, 4>(lv, V, tid, sp[s].noise ? sp[s].noise + a.vocab + (long)cb * V : nullptr, seed[s], frame[s], 1, cb * V, a.inv_temp,
                                                     a.top_p, reinterpret_cast<double*>(scr));
                int tk = raw;
                if (use_forced) tk = sp[s].forced[(long)cb * a.chunk + a.ci];
                tprev[s] = tk;
                if (tid == 0) toks[s][cb] = tk;
                if (wg == 0 && tid == 0) { sp[s].tok_raw[cb] = raw; sp[s].tok[cb] = tk; }
                __syncthreads();          // scr (sampler scratch) / lg / xs are rewritten next
            }
        }
    }

    // ======================================= frame bookkeeping =======================================
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // cached_new_audio_emb = embed(codes) (dual_ar_stream.py:834, 245-255): 8 features per workgroup, codebooks summed in order
        if (tid < 8) {
            const int i = wg * 8 + tid;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < NCB; ++q) acc += a.codebook_emb[((long)toks[s][q] + (long)q * a.codebook_size) * D + i];
            sp[s].cached_audio_emb[i] = acc;
        }
    }
    if (wg != 0) return;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (tid < NCB) {
            sp[s].pred_hist[(long)tid * a.hist_cap + (frame[s] & (a.hist_cap - 1))] = toks[s][tid];
            sp[s].step_audio[tid * a.chunk + a.ci] = toks[s][tid];
        }
        if (tid == 0) {
            sp[s].step_content[a.ci] = code[s];
            *sp[s].nframes = frame[s] + 1;
            *sp[s].last_pos = p0[s] + 1;
            *sp[s].epoch = ep[s];
            if (a.fail_host) {
                const int f = *reinterpret_cast<volatile int*>(a.fail);
                if (f) *reinterpret_cast<volatile int*>(a.fail_host) = f;
            }
        }
    }
    if (!a.skip_semantic) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float lsem[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int e = tid + 256 * r;
                lsem[r] = e < a.vocab ? __hip_atomic_load(sp[s].slow_logits + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
            }
            __syncthreads();
            const int sm = nucleus_sample<4, 32>(lsem, a.vocab, tid, sp[s].noise, seed[s], frame[s], 0, 0, a.inv_temp, a.top_p, reinterpret_cast<double*>(scr));
            if (tid == 0) *sp[s].sem = sm;
        }
    }
#undef XS
#undef BIG
#undef AV
#undef LG
#undef EP_INC
}

constexpr size_t group_lds_floats(int ns) { return (size_t)ns * (PS + 4 * 68 + 16 * 68) + NCB * 64; }

template <typename WT, typename KVT, int NS>
int launch_group_t(const ArDecodeArgs& a, hipStream_t st, int n_groups) {
    const size_t smem = group_lds_floats(NS) * sizeof(float);
    static_assert(group_lds_floats(NS) * sizeof(float) <= (size_t)160 * 1024, "LDS of one CU");
    static DeviceOnce attr;
    if (attr.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)ar_group_kernel<WT, KVT, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr.done();
    }
    hipLaunchKernelGGL((ar_group_kernel<WT, KVT, NS>), dim3(AR_WGS, n_groups), dim3(256), smem, st, a);
    SVA_HIP(hipGetLastError());
    return 0;
}

template <typename WT, typename KVT>
int launch_group_ns(const ArDecodeArgs& a, hipStream_t st, int ns, int n_groups) {
    switch (ns) {
        case 2: return launch_group_t<WT, KVT, 2>(a, st, n_groups);
        case 3: return launch_group_t<WT, KVT, 3>(a, st, n_groups);
        case 4: return launch_group_t<WT, KVT, 4>(a, st, n_groups);
    }
    set_error("ar_group: 2..4 streams per group");
    return -1;
}

}  // namespace

// ns streams per group (2..4), n_groups groups: slots a.slot_base .. a.slot_base + ns * n_groups - 1; grid 96 x n_groups, every
// workgroup must be resident (more than 80 KiB of LDS each: one per CU)
int launch_ar_group(const ArDecodeArgs& a, int wt_half, int kv_half, int ns, int n_groups, hipStream_t st) {
    SVA_CHECK(ns >= 2 && ns <= 4 && n_groups >= 1 && n_groups * AR_WGS <= 256, "ar_group: 2..4 streams per group, at most 256 workgroups");
    SVA_CHECK(a.vocab <= 22 * AR_WAVES && a.codebook_size <= 3 * AR_WAVES && a.codebook_size <= 1024 && a.codebook_size % 2 == 0 && (a.hist_cap & (a.hist_cap - 1)) == 0,
              "ar_group: unsupported head sizes");
    SVA_CHECK(wt_half == kv_half, "ar_group: fp32 weights + fp32 KV, or fp16 weights + fp16 KV");
    if (wt_half) return launch_group_ns<__half, __half>(a, st, ns, n_groups);
    return launch_group_ns<float, float>(a, st, ns, n_groups);
}

}  // namespace sva
