// sva engine: per-batch state, stream / graph orchestration and the per-chunk step of the streaming
// voice-conversion hot path (content encoder -> dual AR -> Firefly vocoder) on MI355X.
//
// Host-side control flow mirrors evaluations/infer_arvc.py InferenceWrapper.{prefill_prompt,
// setup_stream_caches, process_one_chunk} (:443-596) and modules/dual_ar_stream.py
// DualARWrapper.{prefill_prompt, prefill_src_condition4delay, decode_one} (:764-837); the
// arithmetic runs in the HIP kernels of gemm.hip / kernels.hip.
#include "engine_internal.h"
#include "engine_kernels.h"
#include <map>
#include <mutex>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace sva {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

}  // namespace sva

using namespace sva;

// joins the AR / vocoder streams of the pipelined mode back into the main stream (no-op when nothing is in flight there)
namespace sva {
static void parse_debug(DebugOptions& o, const char* env) {
    if (!env) return;
    std::string all(env);
    size_t pos = 0;
    while (pos <= all.size()) {
        const size_t end = std::min(all.find(',', pos), all.size());
        const std::string kv = all.substr(pos, end - pos);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) {
            const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
            if (k == "ar_timing") o.ar_timing = atoi(v.c_str());
            else if (k == "pipe_trace") o.pipe_trace = atoi(v.c_str());
            else if (k == "concurrency") o.concurrency = atoi(v.c_str());
            else if (k == "ar_persistent") o.ar_persistent = atoi(v.c_str());
            else if (k == "ar_batch") o.ar_batch = atoi(v.c_str());
            else if (k == "ar_batch_wgs") o.ar_batch_wgs = atoi(v.c_str());
            else if (k == "planes_dbg") o.planes_dbg = atoi(v.c_str());
            else if (k == "reprefill") o.reprefill = atoi(v.c_str());
            else if (k == "planes_dma") o.planes_dma = atoi(v.c_str());
            else if (k == "voc_dma") o.voc_dma = atoi(v.c_str());
            else if (k == "voc_dma_variant") o.voc_dma_variant = atoi(v.c_str());
            else if (k == "planes_lw") o.planes_lw = atoi(v.c_str());
            else if (k == "planes_min_streams") o.planes_min_streams = atoi(v.c_str());
            else if (k == "ar_graph") o.ar_graph = atoi(v.c_str());
            else if (k == "ar_pairs") o.ar_pairs = atoi(v.c_str());
            else if (k == "head_fuse") o.head_fuse = atoi(v.c_str());
            else if (k == "voc_fused_mask") o.voc_fused_mask = atoi(v.c_str());
            else if (k == "autotune") o.autotune = atoi(v.c_str());
            else if (k == "tune_log") o.tune_log = atoi(v.c_str());
            else if (k == "tune_table") o.tune_table = atoi(v.c_str());
            else if (k == "tune_kinds") o.tune_kinds = atoi(v.c_str());
            else if (k == "f16_weights") o.f16_weights = atoi(v.c_str());
            else if (k == "cu_partition") o.cu_partition = atoi(v.c_str());
            else if (k == "cu_ar") o.cu_ar = atoi(v.c_str());
            else if (k == "pipe_skip") o.pipe_skip = atoi(v.c_str());
            else if (k == "tune_dump") o.tune_dump = v;
            else fprintf(stderr, "[sva] debug option '%s' unknown, ignored\n", k.c_str());
        }
        pos = end + 1;
    }
}
static DebugOptions& debug_options_mut() {
    // never destroyed: the tuning dump (an atexit handler registered before this object exists) reads it during process teardown
    static DebugOptions& opt = *new DebugOptions([] { DebugOptions o; parse_debug(o, getenv("SVA_DEBUG")); return o; }());
    return opt;
}
const DebugOptions& debug_options() { return debug_options_mut(); }
}  // namespace sva

namespace { int quiesce(sva_batch* b); }

// Recovery from a persistent-kernel timeout (called by sva_prefill_prompt / sva_streams_begin): clear the device flag, leave the
// persistent kernel for the multi-launch decode (whatever kept its workgroups from being co-resident may still be there), drop the
// graphs that captured its launches, and require fresh prompts: the KV caches and positions of the failed frames are garbage.
static int recover_ar_failure(sva_batch* b) {
    if (!b->ar_failed) return 0;
    SVA_HIP(hipDeviceSynchronize());
    SVA_HIP(hipMemset(b->d_ar_fail, 0, sizeof(int)));
    if (b->h_ar_fail) *b->h_ar_fail = 0;
    b->use_mega = false;
    b->use_abatch = false;
    for (auto& ge : b->pipe_graph_a) if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
    if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
    std::fill(b->prefilled.begin(), b->prefilled.end(), 0);
    b->begun = false;
    b->ar_failed = false;
    return 0;
}

// ============================================================================================
// config defaults
// ============================================================================================
extern "C" const char* sva_last_error(void) { return g_err.c_str(); }


// ============================================================================================
// batch
// ============================================================================================

namespace {
struct StreamSet { hipStream_t main = nullptr, aux0 = nullptr, sa = nullptr, sv = nullptr, aux1 = nullptr; };
std::mutex g_streams_mu;
std::map<int, StreamSet> g_streams;          // per device, process lifetime
// `partitioned`: the AR stream and the encoder / vocoder streams get disjoint compute-unit masks (hipExtStreamCreateWithCUMask;
// mask bit i -> XCD i % 8, so every range is spread over all XCDs).  With few streams the three stage chains are strings of
// short dependent kernels that each leave most of the chip idle, yet when they share CUs the AR chain's kernels queue behind
// the encoder's and vocoder's workgroups.  The gain shrinks with the batch (and wants a smaller AR share as it grows) and turns
// into a loss beyond 32 streams, where the GEMMs want the whole chip.  SVA_DEBUG=cu_partition=0|1[,cu_ar=N] overrides for A/B runs.
int get_streams(int device, bool need_aux1, int n_streams_if_pipelined, StreamSet* out, int ar_cus = 96) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    // variant 1 (one stream): AR 96 CUs | encoder 128 | vocoder 32 (12 / 16 / 4 per XCD) -- 2.11 -> 1.75 ms per step, 1.60 with the
    // encoder split; 96/136/24 measures the same but leaves the vocoder no slack (96/138/22 and 92/140/24 lose the whole gain:
    // the AR GEMV grids are multiples of 96 workgroups, the vocoder is throughput-bound on its share).  Two streams and more: AR
    // `ar_cus` CUs | encoder and vocoder share the rest (the three-way split is worse there); the caller picks ar_cus by batch size.
    bool partitioned = n_streams_if_pipelined >= 1;
    int variant = !partitioned ? 0 : n_streams_if_pipelined == 1 ? 1 : ar_cus;      // (a stream set per split, created on first use)
    int part[6] = {0, 96, 96, 160, 96, 160};
    if (variant == 1) { part[3] = 128; part[4] = 224; part[5] = 32; }
    if (variant > 1) {
        const int n = std::min(std::max(ar_cus & ~7, 32), 224);
        part[1] = n; part[2] = part[4] = n; part[3] = part[5] = 256 - n;
        variant = n;
    }
    if (partitioned) {
        hipDeviceProp_t prop;
        SVA_HIP(hipGetDeviceProperties(&prop, device));
        if (prop.multiProcessorCount != 256) partitioned = false;         // the ranges are sized for the 256 CUs of an MI355X
    }
    if (!partitioned) variant = 0;
    StreamSet& s = g_streams[device * 256 + variant];
    if (!s.main) {
        auto make = [&](hipStream_t* st, int lo, int n) -> int {
            if (!partitioned || n <= 0 || n >= 256) { SVA_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking)); return 0; }
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = lo; i < lo + n && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
            if (hipExtStreamCreateWithCUMask(st, 8, mask) != hipSuccess) {      // (a runtime without CU masking: plain stream, no partition)
                (void)hipGetLastError();
                SVA_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
            }
            return 0;
        };
        SVA_TRY(make(&s.main, part[2], part[3]));
        SVA_TRY(make(&s.aux0, part[2], part[3]));
        SVA_TRY(make(&s.sa, part[0], part[1]));
        SVA_TRY(make(&s.sv, part[4], part[5]));
    }
    if (need_aux1 && !s.aux1) SVA_HIP(hipStreamCreateWithFlags(&s.aux1, hipStreamNonBlocking));     // legacy three-stream vocoder only
    *out = s;
    return 0;
}
}  // namespace

// batch sizes the batched persistent decode kernel (ar_batch.hip) serves: where it beats both the two-streams-per-launch kernel
// (ar_decode.hip, below) and the multi-launch decode (above).  Pipelined (throughput) mode, fp32 AR: 4-24 streams (4: 1681 vs 1632
// frames/s, 5: 2064 vs 1654, 6: 2421 vs 1914, 8: 2949 vs 2508, 16: 4439 vs 4133, 32: 5288 vs 5419); fp16 AR: 3-11 (3: 1624 vs 1487, 4: 2071
// vs 1914, 8: 3440 vs 3197, 12: 4059 vs 4110); with several frames per step (chunk > 1) the decode is the step's longest stage and the
// kernel serves up to 32 streams (32 x chunk 4: 7763 vs 6341 frames/s).  A caller that synchronises every chunk sees the kernel's
// ~2.1-2.3 ms frame against 1.75 ms for two launches of the two-stream kernel: there it starts at 5 streams
// (profiles/r04_small_batch_ab.txt, r04_abatch_sweep in the comments above).  SVA_DEBUG ar_batch=0 never, ar_batch=2 every size it can run
// (a GROUP form of the persistent kernel -- 2-4 streams sharing each phase's weight registers and hand-offs on one set of 96 workgroups --
// was built in round 4, measured slower than this policy at every size and removed in round 5: profiles/r04_group_ab.txt, git history)
// The PIPELINED-mode policy -- which decode serves a batch of B streams (the batched persistent kernel or the multi-launch chain) and how many
// CUs the AR chain's stream owns when it is the chain -- is a GENERATED table: tools/make_policy.py sweeps the candidates on the GPU
// (frames/s of bench.py per stream count x decode x CU partition; log under profiles/) and emits policy_table.inc, so a kernel change
// re-derives every threshold with one command instead of hand-edited constants.  A batch takes the row of the largest
// measured stream count <= its own (same AR precision, same chunk class).
struct PolicyRow { int ar_dtype, chunk_gt1, B, decode, cu_ar; };       // decode: 0 multi-launch chain, 2 batched persistent kernel; cu_ar: 0 = no partition
static const PolicyRow kPolicy[] = {
#include "policy_table.inc"
};
static const PolicyRow* policy_row(int B, int ar_dtype, int chunk) {
    const PolicyRow *below = nullptr, *smallest = nullptr;
    for (const PolicyRow& r : kPolicy) {
        if (r.ar_dtype != ar_dtype || r.chunk_gt1 != (chunk > 1 ? 1 : 0)) continue;
        if (r.B <= B && (!below || r.B > below->B)) below = &r;
        if (!smallest || r.B < smallest->B) smallest = &r;
    }
    return below ? below : smallest;          // (fewer streams than any measured row: the smallest one)
}
// smallest pipelined stream count the batched kernel serves = one more than what the per-stream persistent kernel (ar_decode.hip) takes
static int abatch_lo(int ar_dtype, bool pipelined) { return pipelined ? (ar_dtype == 1 ? 3 : 4) : 5; }
static bool abatch_serves(int B, int ar_dtype, bool pipelined, int chunk) {
    const int mode = debug_options().ar_batch;
    if (mode == 0 || B > AR_BATCH_MAX_STREAMS) return false;
    if (mode == 2) return true;
    if (B < abatch_lo(ar_dtype, pipelined)) return false;
    if (!pipelined) return B <= 32;           // a caller that synchronises every chunk: one launch per frame from 5 streams (profiles/r04_small_batch_ab.txt)
    const PolicyRow* r = policy_row(B, ar_dtype, chunk);
    return r ? r->decode == 2 : B <= 32;
}

static int batch_create_impl(sva_engine* e, const sva_stream_params* p, sva_batch* b);

extern "C" int sva_batch_create(sva_engine* e, const sva_stream_params* p, sva_batch** out) {
    SVA_CHECK(e && p && out && e->finalized, "engine not finalized");
    SVA_HIP(hipSetDevice(e->device));
    (void)hipGetLastError();
    sva_batch* b = new sva_batch();
    for (auto& ev : b->evpool) ev = nullptr;
    for (auto& ev : b->ev) ev = nullptr;
    b->e = e;
    const int rc = batch_create_impl(e, p, b);
    if (rc) {                      // nothing of a half-built batch survives (events, arena chunks, pinned buffers)
        const std::string msg = sva_last_error();
        sva_batch_destroy(b);
        set_error(msg);
        return rc;
    }
    *out = b;
    return 0;
}

static int batch_create_impl(sva_engine* e, const sva_stream_params* p, sva_batch* b) {
    const sva_config& c = e->cfg;
    b->p = *p;
    const int B = b->B = p->n_streams;
    SVA_CHECK(B >= 1 && p->chunk_frames >= 1 && p->delay >= 1 && p->delay <= c.max_delay, "bad stream params (delay 0 is broken upstream too)");
    // a re-prefill rebuilds [speaker prefix | 2 x (truncated prompt + buffer_frames)] + the delay fill in one slot's cache (infer_arvc.py:547-564)
    SVA_CHECK(c.timbre_tokens + 1 + 2 * (p->max_prompt_frames + p->buffer_frames) + 2 * p->delay <= c.max_seq_len,
              "max_prompt_frames + buffer_frames do not fit the KV cache: a re-prefill would overrun the slot");
    SVA_CHECK(p->encode_window_frames % 1 == 0 && p->encode_window_frames >= p->chunk_frames, "bad encode window");
    if (b->p.voc_max_frames < p->chunk_frames) b->p.voc_max_frames = p->chunk_frames;
    // Streams come from a process-wide set per device, created once in a fixed order (main, encoder side stream, AR,
    // vocoder) and shared by every batch: the runtime multiplexes streams onto a few hardware queues (4 by default) in
    // creation order, two streams on one hardware queue serialise (a stream stuck behind another one's event wait blocks
    // its queue mates), and streams created after others were destroyed land on unlucky queues -- measured: the pipelined
    // mode gains 16 % at 64 streams in a fresh process and nothing after one create / destroy cycle.  Batches that are
    // alive at the same time therefore share the streams (in-order, so still correct).
    if (b->p.pipeline) SVA_CHECK(b->voc_grouped, "stage pipelining needs the grouped vocoder launches");
    {
        StreamSet ss;
        // one stream with the persistent AR decode kernel: no CU partition -- that kernel is a single launch of 96 workgroups that
        // mostly wait on hand-offs, so the AR chain no longer queues behind the other stages' workgroups, and the encoder /
        // vocoder GEMMs want the whole chip (measured 1.41 ms per step with the 96 | 128 | 32 split, 1.13 without)
        // (fp16 AR: the batched decode on the f16 pipes overtakes the persistent kernel at 5 streams -- 2191 vs 1645 frames/s, 6: 2562 vs
        // 2064, 4: 1420 vs 1919; fp32: the persistent kernel wins up to 6 -- the round-3 partition A/B scripts, git history)
        int mega_max = debug_options().ar_batch == 0 ? (c.ar_dtype == 1 ? std::min(4, AR_PERSISTENT_MAX_STREAMS) : AR_PERSISTENT_MAX_STREAMS)
                                                     : std::min(abatch_lo(c.ar_dtype, b->p.pipeline != 0) - 1, AR_PERSISTENT_MAX_STREAMS);
        const bool will_mega = B <= mega_max && e->mega_ok && debug_options().ar_persistent != 0 && debug_options().ar_batch != 2;
        b->mega_max = mega_max;
        // multi-launch decode (more than 6 streams): its ~265 small launches per frame are a latency chain that the encoder's and
        // vocoder's chip-filling GEMMs would otherwise queue in front of -- disjoint CU masks (AR n | encoder + vocoder 256 - n)
        // win up to 32 streams, with a smaller AR share as the batch grows (the round-3 partition A/B scripts, git history, round-3 A/B;
        // profiles/r03_partition_ab.txt -- fp32 AR: 8 streams 1873 unpartitioned / 2206 with 96 CUs / 2497 with 128; 12: 2683 / 3422
        // with 96; 24: 3960 / 4312 with 64; 32: 5008 / 5322 with 64; 48: 5725 / 5628.  fp16 AR: 12: 3053 / 3989 with 64; 16: 3682 /
        // 4234; 24: 4415 / 4884; 32: 5337 / 5421).  Round 2 partitioned 7-8 streams only, always 96 | 160.
        // batched persistent decode kernel (ar_batch.hip; the range it serves: batch_create_impl): no CU partition either -- its 72
        // workgroups of 8 waves x 256 registers fill 72 CUs by themselves and leave the other 184 to the encoder and the vocoder
        // (pipelined, fp32 AR, unpartitioned vs the multi-launch decode on its best partition: 8 streams 3066-3169 vs 2538 frames/s,
        // 12: 3761-3795 vs 3463, 16: 4465 vs 4166, 24: 4839 vs 4223, 32: 5288 vs 5419; on 64 / 96 / 128-CU partitions it loses 3-20 %;
        // fp16 AR: 6 streams 2989 vs 2664, 8: 3440 vs 3197, 12: 4059 vs 4110 -- profiles/r04_abatch_sweep.txt)
        const bool will_abatch = !will_mega && abatch_serves(B, c.ar_dtype, b->p.pipeline != 0, p->chunk_frames) && e->mega_ok && debug_options().ar_persistent != 0;
        int ar_cus = 96, part_streams = 0;
        if (b->p.pipeline && !will_mega && !will_abatch && B <= (p->chunk_frames > 1 ? 16 : 32)) {      // (chunk 4: 16 streams +11 %, 32 streams -15 %: the round-3 partition A/B scripts, git history)
            part_streams = B;
            if (B >= 2) ar_cus = c.ar_dtype == 1 ? (B <= 8 ? 96 : 64) : (B <= 8 ? 128 : B <= 20 ? 96 : 64);
        }
        if (b->p.pipeline && !will_mega && !will_abatch && debug_options().ar_batch == 1) {      // the measured table, when it has a row for this batch
            if (const PolicyRow* r = policy_row(B, c.ar_dtype, p->chunk_frames)) {
                if (r->decode == 0) { part_streams = r->cu_ar > 0 ? B : 0; if (r->cu_ar > 0) ar_cus = r->cu_ar; }
            }
        }
        if (debug_options().cu_partition == 0) part_streams = 0;
        else if (debug_options().cu_partition == 1 && b->p.pipeline) { part_streams = B; if (debug_options().cu_ar > 0) ar_cus = debug_options().cu_ar; }
        SVA_TRY(get_streams(e->device, !b->voc_grouped, part_streams, &ss, ar_cus));
        b->ar_cus = part_streams >= 1 ? ar_cus : 0;
        b->ar_partitioned = part_streams >= 1;
        if (b->ar_partitioned) {
            int cus = 0;
            SVA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
            b->enc_cus = std::max(8, cus - ar_cus);
        }
        b->stream = b->main_stream = ss.main;
        b->aux[0] = ss.aux0;
        b->aux[1] = b->voc_grouped ? nullptr : ss.aux1;
        if (b->p.pipeline) { b->sa = ss.sa; b->sv = ss.sv; }
        // split-K scratch per stream, now: a launch inside a stream capture must not allocate (and must not fall back to an
        // unsplit launch, which sums in a different order than the eager launch of the same shape)
        for (hipStream_t s_ : {ss.main, ss.aux0, ss.aux1, ss.sa, ss.sv})
            if (s_) SVA_TRY(conv_gemm_prepare_stream(s_));
    }
    b->out_stream = b->stream;
    for (int i = 0; i < 64; ++i) SVA_HIP(hipEventCreateWithFlags(&b->evpool[i], hipEventDisableTiming));
    if (debug_options().pipe_trace > 0) {
        b->trace_n = debug_options().pipe_trace;
        b->trace_ev.resize((size_t)b->trace_n * 9);
        for (auto& t : b->trace_ev) SVA_HIP(hipEventCreate(&t));
    }
    // a whole-step graph is captured from ONE stream: a capture that forks to the side stream replays at 14 ms per step on this
    // runtime, the single-stream one at 3.8 (eager multi-stream: 3.6)
    if (b->p.use_graph) b->concurrency = false;
    if (!debug_options().concurrency) b->concurrency = false;      // single stream (PMC profiling)
    auto& A = b->allocs;
    const int chunk = p->chunk_frames;
    // control block
    SVA_TRY(dev_alloc(A, &b->d_step, 1));
    SVA_TRY(dev_alloc(A, &b->d_step_x, 1));
    SVA_TRY(dev_alloc(A, &b->d_last_pos, B));
    SVA_TRY(dev_alloc(A, &b->d_nframes, B));
    SVA_TRY(dev_alloc(A, &b->d_ncontent, B));
    SVA_TRY(dev_alloc(A, &b->d_seed, B));
    SVA_TRY(dev_alloc(A, &b->d_use_forced, 1));
    // encoder
    b->We = p->encode_window_frames;
    b->N = b->We * 2048;
    b->T0 = b->N / 512;
    b->T2 = b->T0 / 4;
    SVA_CHECK(b->T2 % 4 == 0 && b->T2 <= 2048, "encode_window_frames must be a multiple of 4 and <= 2048 (the RoPE table of the tokenizer's transformer)");
    const int T0 = b->T0;
    SVA_TRY(dev_alloc(A, &b->ring, (size_t)B * b->N));
    SVA_TRY(dev_alloc(A, &b->d_chunk, (size_t)B * 2048 * chunk));
    SVA_TRY(dev_alloc(A, &b->mag, (size_t)B * T0 * 1088));
    SVA_TRY(alloc_act(A, b->mel, B, 6, T0, c.n_mels));
    for (int i = 0; i < 4; ++i) SVA_TRY(alloc_act(A, b->xs[i], B, 6, T0, c.enc_dims[i]));
    const int Dm = c.tr_dim;
    SVA_TRY(dev_alloc(A, &b->h1, (size_t)B * T0 * Dm));
    SVA_TRY(dev_alloc(A, &b->h2, (size_t)B * T0 * 4 * Dm));
    SVA_TRY(alloc_act(A, b->feat, B, 0, T0, Dm));
    SVA_TRY(alloc_act(A, b->d1, B, 6, T0 / 2, Dm));
    SVA_TRY(alloc_act(A, b->d2, B, 6, T0 / 4, Dm));
    const int T2 = b->T2;
    SVA_TRY(dev_alloc(A, &b->tr_hn, (size_t)B * T2 * Dm));
    SVA_TRY(dev_alloc(A, &b->tr_qkv, (size_t)B * T2 * 3 * Dm));
    SVA_TRY(dev_alloc(A, &b->tr_att, (size_t)B * T2 * Dm));
    SVA_TRY(dev_alloc(A, &b->tr_g, (size_t)B * T2 * c.tr_inter));
    SVA_TRY(dev_alloc(A, &b->tr_z, (size_t)B * T2 * Dm));
    SVA_TRY(dev_alloc(A, &b->tr_x, (size_t)B * T2 * Dm));
    SVA_TRY(alloc_act(A, b->d2c, B, 0, T2, Dm));
    b->Ht = 40;
    b->enc_incremental = T2 > b->Ht + p->chunk_frames;
    {   // streaming state of the exact-incremental encoder
        EncStream& S = b->es;
        const int nm = 4 * chunk;
        std::vector<ShiftDesc> sd;
        auto reg = [&](Act& a, int rows_per_step) {
            if (a.H == 0) return;
            ShiftDesc d;
            d.ptr = a.p; d.bstride = a.bstride; d.H = a.H; d.T = rows_per_step; d.C = a.C; d.pad = 0;
            sd.push_back(d);
        };
        SVA_TRY(dev_alloc(A, &S.mag, (size_t)B * nm * 1088));
        SVA_TRY(alloc_act(A, S.mel, B, 6, nm, c.n_mels));
        reg(S.mel, nm);
        SVA_TRY(alloc_act(A, S.tmp0, B, 0, nm, c.enc_dims[0]));
        S.x.resize(4);
        for (int i = 0; i < 4; ++i) {
            S.x[i].resize(c.enc_depths[i]);
            for (int j = 0; j < c.enc_depths[i]; ++j) {
                SVA_TRY(alloc_act(A, S.x[i][j], B, 6, nm, c.enc_dims[i]));
                reg(S.x[i][j], nm);
            }
            SVA_TRY(alloc_act(A, S.xout[i], B, 0, nm, c.enc_dims[i]));
        }
        SVA_TRY(alloc_act(A, S.feat, B, 0, nm, Dm));
        SVA_TRY(alloc_act(A, S.d1, B, 6, nm / 2, Dm));
        reg(S.d1, nm / 2);
        SVA_TRY(alloc_act(A, S.d1o, B, 0, nm / 2, Dm));
        SVA_TRY(alloc_act(A, S.d2, B, 6, nm / 4, Dm));
        reg(S.d2, nm / 4);
        SVA_TRY(dev_alloc(A, &S.h1, (size_t)B * nm * Dm));
        SVA_TRY(dev_alloc(A, &S.h2, (size_t)B * nm * 4 * Dm));
        S.n_shift = (int)sd.size();
        SVA_TRY(dev_alloc(A, &S.d_shift, sd.size()));
        SVA_HIP(hipMemcpy(S.d_shift, sd.data(), sizeof(ShiftDesc) * sd.size(), hipMemcpyHostToDevice));
        // steady token cache: rows [Ht + c, T2) slide to [Ht, T2 - c) every step
        ShiftDesc dc;
        dc.ptr = b->d2c.p + (long)b->Ht * Dm; dc.bstride = b->d2c.bstride; dc.H = std::max(0, T2 - b->Ht - chunk); dc.T = chunk; dc.C = Dm; dc.pad = 0;
        SVA_TRY(dev_alloc(A, &b->d_shift_d2c, 1));
        SVA_HIP(hipMemcpy(b->d_shift_d2c, &dc, sizeof(ShiftDesc), hipMemcpyHostToDevice));
    }
    if (b->enc_incremental && b->enc_merged) {
        EncMerged& M = b->em;
        M.Hh = 4 * b->Ht; M.nm = 4 * chunk;
        const int Hh = M.Hh, nm = M.nm, R0 = Hh + 6 + nm, R1 = Hh / 2 + 6 + nm / 2, R2 = Hh / 4 + 6 + nm / 4;
        std::vector<ShiftDesc> sd;
        auto reg = [&](Act& a, int head_rows, int new_rows) {      // history rows sit right behind the head rows
            ShiftDesc d;
            d.ptr = a.p + (long)(a.H + head_rows) * a.C; d.bstride = a.bstride; d.H = 6; d.T = new_rows; d.C = a.C; d.pad = 0;
            sd.push_back(d);
        };
        SVA_TRY(dev_alloc(A, &M.mag, (size_t)B * R0 * 1088));
        SVA_TRY(alloc_act(A, M.mel, B, 6, R0, c.n_mels));
        reg(M.mel, Hh, nm);
        SVA_TRY(dev_alloc(A, &M.stem, (size_t)B * R0 * c.enc_dims[0]));
        M.x.resize(4);
        for (int i = 0; i < 4; ++i) {
            M.x[i].resize(c.enc_depths[i]);
            for (int j = 0; j < c.enc_depths[i]; ++j) {
                SVA_TRY(alloc_act(A, M.x[i][j], B, 6, R0, c.enc_dims[i]));
                reg(M.x[i][j], Hh, nm);
            }
            SVA_TRY(alloc_act(A, M.xout[i], B, 0, R0, c.enc_dims[i]));
        }
        SVA_TRY(alloc_act(A, M.feat, B, 0, R0, Dm));
        SVA_TRY(alloc_act(A, M.feat2, B, 0, R0, Dm));
        SVA_TRY(dev_alloc(A, &M.h1b, (size_t)B * R1 * Dm));
        SVA_TRY(dev_alloc(A, &M.h2b, (size_t)B * R1 * 4 * Dm));
        SVA_TRY(alloc_act(A, M.d1, B, 6, R1, Dm));
        reg(M.d1, Hh / 2, nm / 2);
        SVA_TRY(alloc_act(A, M.d1o, B, 0, R1, Dm));
        SVA_TRY(alloc_act(A, M.d2, B, 6, R2, Dm));
        reg(M.d2, Hh / 4, nm / 4);
        SVA_TRY(alloc_act(A, M.tok, B, 0, R2, Dm));
        SVA_TRY(dev_alloc(A, &M.h1, (size_t)B * R0 * Dm));
        SVA_TRY(dev_alloc(A, &M.h2, (size_t)B * R0 * 4 * Dm));
        M.n_shift = (int)sd.size();
        SVA_TRY(dev_alloc(A, &M.d_shift, sd.size()));
        SVA_HIP(hipMemcpy(M.d_shift, sd.data(), sizeof(ShiftDesc) * sd.size(), hipMemcpyHostToDevice));
    }
    SVA_TRY(dev_alloc(A, &b->aatt_part, (size_t)4 * c.ar_heads * 8 * 68));
    SVA_TRY(dev_alloc(A, &b->d_codes_buf[0], (size_t)B * T2));
    SVA_TRY(dev_alloc(A, &b->d_codes_buf[1], (size_t)B * T2));
    b->d_codes = b->d_codes_buf[0];
    SVA_TRY(dev_alloc(A, &b->d_fsq_codes, (size_t)B * c.num_codebooks * T2));
    SVA_TRY(dev_alloc(A, &b->d_u, (size_t)B * T2 * c.bsq_bits));
    // AR
    const int D = c.ar_dim, S = c.max_seq_len, H = c.ar_heads, ncb = c.num_codebooks;
    b->Mmax = std::max(std::max(std::max(2 * B, B * (2 * c.max_delay - 1)), S), std::min(B, 128) * 2 * p->buffer_frames);    // (last: a whole batch re-prefilling at once)
    SVA_TRY(dev_alloc(A, &b->ax, (size_t)b->Mmax * D));
    SVA_TRY(dev_alloc(A, &b->ahn, (size_t)b->Mmax * D));
    SVA_TRY(dev_alloc(A, &b->aqkv, (size_t)b->Mmax * 3 * D));
    SVA_TRY(dev_alloc(A, &b->aatt, (size_t)b->Mmax * D));
    SVA_TRY(dev_alloc(A, &b->ag, (size_t)b->Mmax * c.ar_inter));
    SVA_TRY(dev_alloc(A, &b->xf, (size_t)B * D));
    SVA_TRY(dev_alloc(A, &b->hidden, (size_t)B * D));
    SVA_TRY(dev_alloc(A, &b->slow_logits, (size_t)B * c.ar_vocab));
    SVA_TRY(dev_alloc(A, &b->fast_logits, (size_t)B * ncb * c.codebook_size));
    SVA_TRY(dev_alloc(A, &b->d_slot, b->Mmax));
    SVA_TRY(dev_alloc(A, &b->d_pos, b->Mmax));
    SVA_TRY(dev_alloc(A, &b->d_fast_slot, B));
    SVA_TRY(dev_alloc(A, &b->d_fast_pos, ncb * B));
    {
        std::vector<int> fs(B), fp((size_t)ncb * B);
        for (int i = 0; i < B; ++i) fs[i] = i;
        for (int cb = 0; cb < ncb; ++cb)
            for (int i = 0; i < B; ++i) fp[(size_t)cb * B + i] = cb;
        SVA_HIP(hipMemcpy(b->d_fast_slot, fs.data(), sizeof(int) * B, hipMemcpyHostToDevice));
        SVA_HIP(hipMemcpy(b->d_fast_pos, fp.data(), sizeof(int) * fp.size(), hipMemcpyHostToDevice));
    }
    b->kv_slow_slot = 2L * H * S * 64;
    b->kv_slow_layer = b->kv_slow_slot * B;
    b->kv_fast_slot = 2L * H * ncb * 64;
    b->kv_fast_layer = b->kv_fast_slot * B;
    b->kv_half = c.ar_dtype == 1;              // slow KV cache in fp16 (75.5 MB per stream instead of 151), as the reference's (infer_arvc.py:55-59)
    {
        float* p2;
        if (b->kv_half) { uint16_t* ph; SVA_TRY(dev_alloc(A, &ph, (size_t)c.ar_layers * b->kv_slow_layer)); b->kv_slow = ph; }
        else { float* p1; SVA_TRY(dev_alloc(A, &p1, (size_t)c.ar_layers * b->kv_slow_layer)); b->kv_slow = p1; }
        SVA_TRY(dev_alloc(A, &p2, (size_t)c.ar_fast_layers * b->kv_fast_layer));
        b->kv_fast = p2;
    }
    // persistent batch-1 decode kernel (ar_decode.hip): granule buffers, tag epoch, timeout word, fast K/V scratch
    // up to mega_max_b streams decode in ONE launch of it (96 workgroups per stream, each stream's group talks only to itself); the
    // multi-launch chain of the batched path (~200 dependent launches, 2.4-3.3 ms per frame at 2-8 streams) takes over above that
    b->use_mega = B <= b->mega_max && e->mega_ok && b->fused_decode && debug_options().ar_persistent != 0 && debug_options().ar_batch != 2;
    if (b->use_mega) {
        // every workgroup of a persistent launch must be resident at once: check the launch geometry against the occupancy query and
        // the CUs the AR stream may use, never assume it.  One workgroup per CU is what the kernel is sized for (256 registers, 4 waves);
        // the query's answer beyond 1 is not relied on (it over-reports by one on SGPR-heavy kernels, MI355X_MICROARCH.md).
        int per_cu = 0, cus = 0;
        SVA_TRY(ar_decode_occupancy(c.ar_dtype == 1, b->kv_half, &per_cu));
        SVA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
        const int avail = b->ar_partitioned ? std::min(cus, b->ar_cus > 0 ? b->ar_cus : 96) : cus;       // the AR stream's CU mask (get_streams) is CUs 0 .. ar_cus - 1
        const int wgs = AR_WGS;
        b->mega_per_launch = (avail >= 2 * wgs && !b->ar_partitioned) ? 2 : 1;
        if (per_cu < 1 || avail < wgs) b->use_mega = false;                // fall back to the multi-launch decode
    }
    if (b->use_mega) {
        SVA_TRY(dev_alloc(A, &b->d_gran, ar_decode_granule_words() * B));
        SVA_TRY(dev_alloc(A, &b->d_epoch, B));
        SVA_TRY(dev_alloc(A, &b->d_ar_fail, 1));
        SVA_TRY(dev_alloc(A, &b->kv_fast_mega, (size_t)B * AR_FAST_LAYERS * 8 * 2 * D));
        if (debug_options().ar_timing) SVA_TRY(dev_alloc(A, &b->d_ar_dbg, 1024));
    }
    // batched persistent decode kernel (ar_batch.hip): every stream of the batch in one launch per frame.  Takes the batches the
    // two-streams-per-launch kernel above does not serve; its workgroups must all be resident (same check as above).
    b->use_abatch = !b->use_mega && e->mega_ok && debug_options().ar_persistent != 0 && abatch_serves(B, c.ar_dtype, b->p.pipeline != 0, b->p.chunk_frames) && c.ar_vocab <= 8192 &&
                    c.codebook_size <= 1024;
    if (b->use_abatch) {
        int per_cu = 0, cus = 0;
        SVA_TRY(ar_batch_occupancy(c.ar_dtype == 1, B, &per_cu));
        SVA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
        const int avail = b->ar_partitioned ? std::min(cus, b->ar_cus > 0 ? b->ar_cus : 96) : cus;
        const int cap = std::min(per_cu, 1) * avail;             // 8 waves with up to 256 registers each: one workgroup per CU (the occupancy query's answer beyond 1 is not relied on)
        int G = ar_batch_wanted_workgroups(B);
        if (debug_options().ar_batch_wgs > 0) G = debug_options().ar_batch_wgs;
        G = std::min(G, cap);
        if (G < 48) b->use_abatch = false;
        b->abatch_G = G;
    }
    if (b->use_abatch) {
        static_assert(AR_BATCH_NBUF == 11, "ab_offs");
        const size_t words = ar_batch_granule_words(B, b->ab_offs);
        SVA_TRY(dev_alloc(A, &b->d_ab_gran, words));             // zeroed: tag 0 is never a live epoch (the counter starts at 1)
        SVA_TRY(dev_alloc(A, &b->d_ab_epoch, 2));
        if (!b->d_ar_fail) SVA_TRY(dev_alloc(A, &b->d_ar_fail, 1));
        if (debug_options().ar_timing && !b->d_ar_dbg) SVA_TRY(dev_alloc(A, &b->d_ar_dbg, 1024));
    }
    if (c.mm_mode == 1 || c.voc_dtype == 1) {       // fp16 operand planes: their range check reports here (checked by sva_sync)
        SVA_HIP(hipHostMalloc((void**)&b->h_mm_ovf, sizeof(int), hipHostMallocMapped));
        *b->h_mm_ovf = 0;
        SVA_HIP(hipHostGetDevicePointer((void**)&b->d_mm_ovf, b->h_mm_ovf, 0));
    }
    if (b->d_ar_fail) {     // host-visible mirror of the timeout flag (the stream-ordered API never synchronises, so nothing read d_ar_fail)
        SVA_HIP(hipHostMalloc((void**)&b->h_ar_fail, sizeof(int), hipHostMallocMapped));
        *b->h_ar_fail = 0;
        SVA_HIP(hipHostGetDevicePointer((void**)&b->d_ar_fail_host, b->h_ar_fail, 0));
    }
    SVA_TRY(dev_alloc(A, &b->cached_audio_emb, (size_t)B * D));
    SVA_TRY(dev_alloc(A, &b->cached_ref_emb, (size_t)B * c.max_delay * D));
    SVA_TRY(dev_alloc(A, &b->d_ref_tail, (size_t)B * c.num_codebooks * c.max_delay));
    SVA_TRY(dev_alloc(A, &b->spk, (size_t)(c.timbre_tokens + 1) * D));
    SVA_TRY(dev_alloc(A, &b->d_style, (size_t)B * c.style_dim));
    SVA_TRY(dev_alloc(A, &b->d_timbre, (size_t)B * c.timbre_tokens * c.timbre_dim));
    SVA_TRY(dev_alloc(A, &b->d_sem, B));
    SVA_TRY(dev_alloc(A, &b->d_tok, B * ncb));
    SVA_TRY(dev_alloc(A, &b->d_tok_raw, B * ncb));
    SVA_TRY(dev_alloc(A, &b->d_forced, (size_t)B * ncb * chunk));
    SVA_TRY(dev_alloc(A, &b->d_noise, (size_t)B * chunk * (c.ar_vocab + ncb * c.codebook_size)));
    b->hist_cap = 4096;                      // power of two (ring indexing); >= buffer_frames + delay + chunk
    SVA_CHECK(p->buffer_frames + p->delay + chunk < b->hist_cap, "buffer_frames too large");
    SVA_TRY(dev_alloc(A, &b->d_slot_list, B));
    SVA_TRY(dev_alloc(A, &b->d_content_hist, (size_t)B * b->hist_cap));
    SVA_TRY(dev_alloc(A, &b->d_pred_hist, (size_t)B * ncb * b->hist_cap));
    SVA_TRY(dev_alloc(A, &b->d_step_content, (size_t)B * chunk));
    SVA_TRY(dev_alloc(A, &b->d_step_audio_buf[0], (size_t)B * ncb * chunk));
    SVA_TRY(dev_alloc(A, &b->d_step_audio_buf[1], (size_t)B * ncb * chunk));
    b->d_step_audio = b->d_step_audio_buf[0];
    b->Pmax = S;
    SVA_TRY(dev_alloc(A, &b->d_prompt_cc, b->Pmax));
    SVA_TRY(dev_alloc(A, &b->d_prompt_ac, (size_t)ncb * b->Pmax));
    b->ref_content.resize(B); b->ref_audio.resize(B); b->ref_len.assign(B, 0);
    b->h_last_pos.assign(B, -1); b->h_nframes.assign(B, 0); b->prefilled.assign(B, 0);
    // vocoder
    const int Tv = b->Tv = b->p.voc_max_frames;
    const int V = c.voc_dim;
    if (debug_options().voc_fused_mask >= 0) b->voc_fused_mask = debug_options().voc_fused_mask;
    SVA_TRY(dev_alloc(A, &b->d_voc_frames, 1));
    SVA_HIP(hipMemset(b->d_voc_frames, 0, sizeof(int)));
    SVA_TRY(alloc_act(A, b->zq, B, 0, Tv, V));
    SVA_TRY(alloc_act(A, b->u0, B, 6, 2L * Tv, V));
    SVA_TRY(alloc_act(A, b->v0, B, 0, 2L * Tv, V));
    SVA_TRY(alloc_act(A, b->u1, B, 6, 4L * Tv, V));
    SVA_TRY(alloc_act(A, b->pin, B, e->pre_k - 1, 4L * Tv, V));
    SVA_TRY(dev_alloc(A, &b->vh1, (size_t)B * 4 * Tv * V));
    SVA_TRY(dev_alloc(A, &b->vh2, (size_t)B * 4 * Tv * 4 * V));
    SVA_TRY(alloc_act(A, b->S[0], B, 1, 4L * Tv, V));
    SVA_TRY(register_shift(b, b->u0, 2));
    SVA_TRY(register_shift(b, b->u1, 4));
    SVA_TRY(register_shift(b, b->pin, 4));
    SVA_TRY(register_shift(b, b->S[0], 4));
    long rows = 4L * Tv;
    int rpf = 4;
    int ch = V;
    for (int i = 0; i < 5; ++i) {
        rows *= e->ups_s[i];
        rpf *= e->ups_s[i];
        ch /= 2;
        b->voc_rpf[i] = rpf;
        const bool fused_level = voc_level_is_fused(b, ch);
        // Every level of a batch with enough rows per step: the 18 ResBlock convs on operand planes, three branches per launch (C >= 64: the LDS-DMA
        // planes kernel's conv form; C = 16 / 32: voc_conv_kernel), the activations between them as planes (history included).  A static choice per
        // batch -- the history lives in one form.  From 10 code frames per step over the batch (streams x voc_max_frames): +4.7 / +5.7 / +6.7 / +5.2 /
        // +3.9 % frames/s at 16 / 24 / 32 / 48 / 128 streams, +2 % and a 6 % shorter synchronous step at 10 / 12, even at 8 (profiles/r05_voc_dma_sweep.txt)
        const int voc_pm = c.voc_dtype == 1 ? PLANES_H1 : (c.mm_mode == 1 ? PLANES_H3 : -1);
        bool dma_level = !fused_level && ch % 16 == 0 && voc_pm >= 0 && debug_options().planes_dma != 0 && debug_options().voc_dma != 0 &&
                         (debug_options().voc_dma == 1 || (long)B * Tv >= 10);
        // C = 16 / 32: row-major planes + voc_conv_kernel (the input rows of a tile and the branch's whole weight resident in LDS); wider: K-blocked planes +
        // the LDS-DMA GEMM's conv form
        const bool halo_level = dma_level && ch <= 32 && voc_conv_supported(ch, rpf, voc_pm);
        for (int br = 0; br < 3 && dma_level; ++br)
            for (int j = 0; j < 3; ++j) {
                const ResConv& rcv = e->res[i][br][j];
                if (halo_level) dma_level = dma_level && rcv.q1 && rcv.q2;
                else dma_level = dma_level && ch % 64 == 0 && rcv.c1.Wp && rcv.c2.Wp && rcv.c1.pmode == voc_pm && rcv.c2.pmode == voc_pm;
            }
        if (dma_level) b->voc_pmode = voc_pm;
        b->voc_dma[i] = dma_level ? (halo_level ? 2 : 1) : 0;
        auto alloc_planes = [&](const Act& a, unsigned short** P) -> int {
            const int npl = planes_count(b->voc_pmode);
            const size_t n = (size_t)npl * B * a.bstride;
            SVA_TRY(dev_alloc(A, P, n));
            if (a.H == 0) return 0;
            for (int p = 0; p < npl; ++p) {
                if (halo_level) {       // row-major plane: a [B][rows][C / 2 floats] tensor
                    ShiftDesc d;
                    d.ptr = reinterpret_cast<float*>(*P + (size_t)p * B * a.bstride);
                    d.bstride = a.rows * (a.C / 2); d.H = a.H; d.T = 0; d.C = a.C / 2; d.pad = rpf;
                    b->shift_host.push_back(d);
                    continue;
                }
                // K-blocked: the history rows of every (plane, 32-channel block) shift like a [B][rows][16 floats] tensor of its own
                for (int kb = 0; kb < a.C / 32; ++kb) {
                    ShiftDesc d;
                    d.ptr = reinterpret_cast<float*>(*P + (size_t)p * B * a.bstride + (size_t)kb * B * a.rows * 32);
                    d.bstride = a.rows * 16; d.H = a.H; d.T = 0; d.C = 16; d.pad = rpf;
                    b->shift_host.push_back(d);
                }
            }
            return 0;
        };
        // fused levels keep the receptive field of the whole six-conv chain as input history (their only streaming state)
        SVA_TRY(alloc_act(A, b->X[i], B, fused_level ? (kResK[2] - 1) * 2 * (kResD[0] + kResD[1] + kResD[2]) : (kResK[2] - 1) * kResD[0], rows, ch));
        if (dma_level) SVA_TRY(alloc_planes(b->X[i], &b->XP[i]));        // (the fp32 tensors of such a level are read as residuals only: current rows)
        else SVA_TRY(register_shift(b, b->X[i], rpf));
        for (int br = 0; br < 3; ++br)
            for (int j = 0; j < 3; ++j) {
                SVA_TRY(alloc_act(A, b->tb[i][br][j], B, (kResK[br] - 1) * kResD[j], rows, ch));
                if (dma_level) SVA_TRY(alloc_planes(b->tb[i][br][j], &b->tbP[i][br][j]));
                else SVA_TRY(register_shift(b, b->tb[i][br][j], rpf));
                if (j < 2) {
                    SVA_TRY(alloc_act(A, b->yb[i][br][j], B, (kResK[br] - 1) * kResD[j + 1], rows, ch));
                    if (dma_level) SVA_TRY(alloc_planes(b->yb[i][br][j], &b->ybP[i][br][j]));
                    else SVA_TRY(register_shift(b, b->yb[i][br][j], rpf));
                }
            }
        for (int br = 0; br < 3; ++br) SVA_TRY(alloc_act(A, b->y3[i][br], B, 0, rows, ch));
        SVA_TRY(alloc_act(A, b->S[i + 1], B, i < 4 ? 1 : e->post_k - 1, rows, ch));
        SVA_TRY(register_shift(b, b->S[i + 1], rpf));
    }
    SVA_TRY(dev_alloc(A, &b->d_pcm, (size_t)B * 2048 * Tv));
    SVA_TRY(dev_alloc(A, &b->d_vcodes, (size_t)B * ncb * Tv));
    SVA_TRY(dev_alloc(A, &b->d_shift, b->shift_host.size()));
    SVA_HIP(hipHostMalloc((void**)&b->hp_in, sizeof(float) * (size_t)B * 2048 * chunk));
    SVA_HIP(hipHostMalloc((void**)&b->hp_out, sizeof(float) * (size_t)B * 2048 * chunk));
    for (int i = 0; i < 5; ++i) SVA_HIP(hipEventCreate(&b->ev[i]));
    b->ev_ok = true;
    if (b->use_mega || b->use_abatch) {        // counted BEFORE the synchronisation below: launches another batch enqueued without a chain record are complete behind it
        std::lock_guard<std::mutex> lk(e->mega_mu);
        e->persistent_batches += 1;
        b->counted_persistent = true;
    }
    SVA_HIP(hipDeviceSynchronize());
    return 0;
}

extern "C" void sva_batch_destroy(sva_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->e->device);
    (void)quiesce(b);
    {   // (a later batch may be allocated at this address: the chain's "same batch, no wait" test must not match it)
        std::lock_guard<std::mutex> lk(b->e->mega_mu);
        if (b->e->mega_last == b) b->e->mega_last = nullptr;
        if (b->counted_persistent) { b->e->persistent_batches -= 1; b->counted_persistent = false; }
    }
    if (b->main_stream) (void)hipStreamSynchronize(b->main_stream);
    for (int i = 0; i < 2; ++i) if (b->aux[i]) (void)hipStreamSynchronize(b->aux[i]);
    if (b->sa) (void)hipStreamSynchronize(b->sa);
    if (b->sv) (void)hipStreamSynchronize(b->sv);
    if (b->trace_n && b->trace_steps >= b->trace_n) {       // SVA_PIPE_TRACE: when each chain of the last steps could start / ended (us)
        const long s0 = b->trace_steps - b->trace_n;
        hipEvent_t base = b->trace_ev[(size_t)(s0 % b->trace_n) * 9];
        fprintf(stderr, "[sva pipe trace] step: main[start end] side[start] transformer[start end] ar[start end] vocoder[start end]  (us from the first row)\n");
        for (long q = s0; q < b->trace_steps; ++q) {
            float t[9];
            for (int k = 0; k < 9; ++k) { t[k] = 0.f; (void)hipEventElapsedTime(&t[k], base, b->trace_ev[(size_t)(q % b->trace_n) * 9 + k]); }
            fprintf(stderr, "[sva pipe trace] %ld: main %.0f %.0f | side %.0f | tr %.0f %.0f | ar %.0f %.0f | voc %.0f %.0f\n", q, t[0] * 1e3, t[1] * 1e3,
                    t[2] * 1e3, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3, t[6] * 1e3, t[7] * 1e3, t[8] * 1e3);
        }
    }
    for (auto& t : b->trace_ev) (void)hipEventDestroy(t);
    if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
    for (auto& ge : b->pipe_graph_a) if (ge) (void)hipGraphExecDestroy(ge);
    for (hipGraphExec_t ge : {b->gEm[0], b->gEm[1], b->gEs[0], b->gEs[1], b->gE, b->gE2, b->gT0, b->gT1[0], b->gT1[1], b->gV}) if (ge) (void)hipGraphExecDestroy(ge);
    for (void* p : b->allocs.chunks) (void)hipFree(p);
    if (b->h_ar_fail) (void)hipHostFree(b->h_ar_fail);
    if (b->h_mm_ovf) (void)hipHostFree(b->h_mm_ovf);
    if (b->hp_in) (void)hipHostFree(b->hp_in);
    if (b->hp_out) (void)hipHostFree(b->hp_out);
    for (int i = 0; i < 5; ++i) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
    for (auto& ev : b->prof_ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < 64; ++i) if (b->evpool[i]) (void)hipEventDestroy(b->evpool[i]);
    (void)hipGetLastError();          // the streams belong to the process-wide set and stay
    delete b;
}

// ============================================================================================
// prompt / begin
// ============================================================================================
namespace {
// All engine copies are ordered on the engine's (non-blocking) stream: a null-stream hipMemcpy from pageable
// memory may return before its DMA lands and is NOT ordered against kernels on a non-blocking stream.
int h2d(sva_batch* b, void* dst, const void* src, size_t bytes) {
    SVA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->stream));
    SVA_HIP(hipStreamSynchronize(b->stream));
    return 0;
}
int stage_prompt(sva_batch* b, const std::vector<int64_t>& cc, const std::vector<int32_t>& ac, int R) {
    // cc [R] int64 -> d_prompt_cc int32; ac [8][R] -> d_prompt_ac [8][Pmax]
    const int ncb = b->e->cfg.num_codebooks;
    SVA_CHECK(R <= b->Pmax, "prompt longer than the staging buffer");
    std::vector<int> c32(R);
    for (int i = 0; i < R; ++i) c32[i] = (int)cc[i];
    SVA_HIP(hipMemcpyAsync(b->d_prompt_cc, c32.data(), sizeof(int) * R, hipMemcpyHostToDevice, b->stream));
    for (int q = 0; q < ncb; ++q)
        SVA_HIP(hipMemcpyAsync(b->d_prompt_ac + (long)q * b->Pmax, ac.data() + (long)q * R, sizeof(int) * R, hipMemcpyHostToDevice, b->stream));
    SVA_HIP(hipStreamSynchronize(b->stream));
    return 0;
}
}  // namespace

extern "C" int sva_prefill_prompt(sva_batch* b, int slot, const int64_t* ref_content_codes, const int32_t* ref_audio_codes, int R,
                                  const float* style, const float* timbre, uint64_t noise_seed) {
    SVA_CHECK(b && slot >= 0 && slot < b->B && ref_content_codes && ref_audio_codes && style && timbre, "bad argument");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    SVA_TRY(recover_ar_failure(b));
    const sva_config& c = b->e->cfg;
    const int ncb = c.num_codebooks;
    std::vector<int64_t> cc(ref_content_codes, ref_content_codes + R);
    std::vector<int32_t> ac(ref_audio_codes, ref_audio_codes + (size_t)ncb * R);
    SVA_TRY(h2d(b, b->d_style + (long)slot * c.style_dim, style, sizeof(float) * c.style_dim));
    SVA_TRY(h2d(b, b->d_timbre + (long)slot * c.timbre_tokens * c.timbre_dim, timbre, sizeof(float) * c.timbre_tokens * c.timbre_dim));
    unsigned long long sd = noise_seed;
    SVA_TRY(h2d(b, b->d_seed + slot, &sd, sizeof(sd)));
    // quirk (iv): the KV prefill uses the UNTRUNCATED prompt (infer_arvc.py:484-489) ...
    SVA_TRY(stage_prompt(b, cc, ac, R));
    SVA_TRY(ar_prefill_slot(b, slot, R, true));
    // ... while the stored prompt (re-prefill, vocoder fill) is truncated to max_prompt_frames (:469-470)
    const int Rt = std::min(R, b->p.max_prompt_frames);
    b->ref_content[slot].assign(cc.begin(), cc.begin() + Rt);
    b->ref_audio[slot].resize((size_t)ncb * Rt);
    for (int q = 0; q < ncb; ++q)
        for (int i = 0; i < Rt; ++i) b->ref_audio[slot][(size_t)q * Rt + i] = ac[(size_t)q * R + i];
    b->ref_len[slot] = Rt;
    {   // the last max_delay frames of the stored prompt's audio codes, for the device-side re-prefill (right-aligned; a shorter prompt leaves the front unused)
        const int md = c.max_delay;
        std::vector<int32_t> tail((size_t)ncb * md, 0);
        for (int q = 0; q < ncb; ++q)
            for (int j = 0; j < md; ++j)
                if (Rt - md + j >= 0) tail[(size_t)q * md + j] = ac[(size_t)q * R + (Rt - md + j)];
        SVA_TRY(h2d(b, b->d_ref_tail + (long)slot * ncb * md, tail.data(), sizeof(int32_t) * tail.size()));
    }
    b->prefilled[slot] = 1;
    return 0;
}

extern "C" int sva_vocode_reset(sva_batch* b) {
    SVA_CHECK(b, "null batch");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    auto zero = [&](Act& a) -> int {
        SVA_HIP(hipMemsetAsync(a.p, 0, sizeof(float) * (size_t)b->B * a.bstride, b->stream));
        return 0;
    };
    SVA_TRY(zero(b->u0)); SVA_TRY(zero(b->u1)); SVA_TRY(zero(b->pin));
    SVA_HIP(hipMemsetAsync(b->d_voc_frames, 0, sizeof(int), b->stream));
    for (int i = 0; i < 6; ++i) SVA_TRY(zero(b->S[i]));
    auto zero_planes = [&](const Act& a, unsigned short* P) -> int {
        if (P) SVA_HIP(hipMemsetAsync(P, 0, sizeof(unsigned short) * (size_t)planes_count(b->voc_pmode) * b->B * a.bstride, b->stream));
        return 0;
    };
    for (int i = 0; i < 5; ++i) {
        SVA_TRY(zero(b->X[i]));
        SVA_TRY(zero_planes(b->X[i], b->XP[i]));
        for (int br = 0; br < 3; ++br)
            for (int j = 0; j < 3; ++j) {
                SVA_TRY(zero(b->tb[i][br][j]));
                SVA_TRY(zero_planes(b->tb[i][br][j], b->tbP[i][br][j]));
                if (j < 2) {
                    SVA_TRY(zero(b->yb[i][br][j]));
                    SVA_TRY(zero_planes(b->yb[i][br][j], b->ybP[i][br][j]));
                }
            }
    }
    SVA_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

namespace {
int upload_vcodes(sva_batch* b, const int32_t* codes, int T) {
    // host codes [B][8][T] -> d_vcodes [B][8][Tv]
    const int ncb = b->e->cfg.num_codebooks;
    SVA_HIP(hipMemcpy2DAsync(b->d_vcodes, sizeof(int) * b->Tv, codes, sizeof(int) * T, sizeof(int) * T, (size_t)b->B * ncb,
                             hipMemcpyHostToDevice, b->stream));
    return 0;
}
int download_pcm(sva_batch* b, int T, float* pcm_out) {
    SVA_HIP(hipMemcpy2DAsync(pcm_out, sizeof(float) * 2048 * T, b->d_pcm, sizeof(float) * 2048 * b->Tv, sizeof(float) * 2048 * T, b->B,
                             hipMemcpyDeviceToHost, b->stream));
    SVA_HIP(hipStreamSynchronize(b->stream));
    return 0;
}
}  // namespace

extern "C" int sva_vocode_stream(sva_batch* b, const int32_t* codes, int T, float* pcm_out) {
    SVA_CHECK(b && codes && pcm_out, "null argument");
    SVA_CHECK(T >= 1 && T <= b->Tv, "vocode: T out of range (1 .. voc_max_frames)");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    SVA_TRY(upload_vcodes(b, codes, T));
    SVA_TRY(vocode(b, T, true));
    return download_pcm(b, T, pcm_out);
}

extern "C" int sva_vocode_window(sva_batch* b, const int32_t* codes, int T, float* pcm_out) {
    SVA_CHECK(b && codes && pcm_out, "null argument");
    SVA_CHECK(T >= 1 && T <= b->Tv, "vocode: T out of range (1 .. voc_max_frames)");
    SVA_TRY(sva_vocode_reset(b));
    SVA_TRY(sva_vocode_stream(b, codes, T, pcm_out));
    return sva_vocode_reset(b);
}

// firefly.quantizer.decode (modules/vqgan/modules/fsq.py:112-116) as a seam of its own: codes int32[B][8][T] -> z float[B][4T][512]
// (channel-last; the reference tensor is [B, 512, 4T]), window semantics (zero history) like sva_vocode_window
extern "C" int sva_quantizer_decode(sva_batch* b, const int32_t* codes, int T, float* z_out) {
    SVA_CHECK(b && codes && z_out, "null argument");
    SVA_CHECK(T >= 1 && T <= b->Tv, "quantizer_decode: T out of range (1 .. voc_max_frames)");
    SVA_TRY(sva_vocode_reset(b));
    SVA_TRY(upload_vcodes(b, codes, T));
    SVA_TRY(vocode(b, T, false, 1));
    const int V = b->e->cfg.voc_dim;
    SVA_HIP(hipMemcpy2DAsync(z_out, sizeof(float) * 4 * T * V, b->pin.p + (long)b->pin.H * V, sizeof(float) * b->pin.bstride, sizeof(float) * 4 * T * V, b->B,
                             hipMemcpyDeviceToHost, b->stream));
    SVA_HIP(hipStreamSynchronize(b->stream));
    return sva_vocode_reset(b);
}

// firefly.head (HiFiGANGenerator.forward, firefly.py:280-293) as a seam of its own: z float[B][4T][512] -> pcm float[B][2048 T]
extern "C" int sva_vocoder_head(sva_batch* b, const float* z, int T, float* pcm_out) {
    SVA_CHECK(b && z && pcm_out, "null argument");
    SVA_CHECK(T >= 1 && T <= b->Tv, "vocoder_head: T out of range (1 .. voc_max_frames)");
    SVA_TRY(sva_vocode_reset(b));
    const int V = b->e->cfg.voc_dim;
    SVA_HIP(hipMemcpy2DAsync(b->pin.p + (long)b->pin.H * V, sizeof(float) * b->pin.bstride, z, sizeof(float) * 4 * T * V, sizeof(float) * 4 * T * V, b->B,
                             hipMemcpyHostToDevice, b->stream));
    SVA_TRY(vocode(b, T, false, 2));
    SVA_TRY(download_pcm(b, T, pcm_out));
    return sva_vocode_reset(b);
}

extern "C" int sva_streams_begin(sva_batch* b) {
    SVA_CHECK(b, "null batch");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    SVA_TRY(recover_ar_failure(b));
    const int B = b->B, ncb = b->e->cfg.num_codebooks, c = b->p.chunk_frames;
    for (int i = 0; i < B; ++i) SVA_CHECK(b->prefilled[i], "every slot needs sva_prefill_prompt before sva_streams_begin");
    // setup_stream_caches (infer_arvc.py:443-460)
    SVA_HIP(hipMemsetAsync(b->ring, 0, sizeof(float) * (size_t)B * b->N, b->stream));
    SVA_HIP(hipMemsetAsync(b->d_step, 0, sizeof(int), b->stream));
    SVA_HIP(hipMemsetAsync(b->d_nframes, 0, sizeof(int) * B, b->stream));
    SVA_HIP(hipMemsetAsync(b->d_ncontent, 0, sizeof(int) * B, b->stream));
    b->h_step = 0; b->h_ncontent = 0; b->delay_filled = false;
    std::fill(b->h_nframes.begin(), b->h_nframes.end(), 0);
    if (b->enc_incremental) {
        // The reference's window starts as all-zero audio (:451): initialise the per-layer histories and the steady
        // token cache with the network's response to silence by streaming zero frames until every history row and
        // every cached token is that steady response (receptive field 117 mel frames; T2 tokens in the cache).
        auto zero = [&](Act& a) -> int {
            SVA_HIP(hipMemsetAsync(a.p, 0, sizeof(float) * (size_t)B * a.bstride, b->stream));
            return 0;
        };
        EncStream& S = b->es;
        SVA_TRY(zero(b->d2c));
        const int warm = 48 / c + 2;             // > receptive field (117 mel frames = 30 tokens) -> histories are steady
        if (b->enc_merged) {
            EncMerged& M = b->em;
            SVA_TRY(zero(M.mel)); SVA_TRY(zero(M.d1)); SVA_TRY(zero(M.d2));
            for (auto& st_ : M.x) for (auto& a : st_) SVA_TRY(zero(a));
            sva_engine* e = b->e;
            std::vector<ShiftDesc> sd(M.n_shift);
            SVA_HIP(hipMemcpy(sd.data(), M.d_shift, sizeof(ShiftDesc) * M.n_shift, hipMemcpyDeviceToHost));
            const auto key = std::make_pair(B, c);
            auto it = e->silence.find(key);
            if (it == e->silence.end()) {
                for (int i = 0; i < warm; ++i) SVA_TRY(enc_frontend_merged(b, nullptr, 0, 0));     // (the ring is all zeros here)
                sva_engine::SilenceState ss;
                for (const ShiftDesc& d : sd) {          // keep the newest history row of item 0 per buffer, and the token row
                    float* r = nullptr;
                    SVA_TRY(dev_alloc(e->allocs, &r, (size_t)d.C, false));
                    SVA_HIP(hipMemcpyAsync(r, d.ptr + (long)(d.H - 1) * d.C, sizeof(float) * d.C, hipMemcpyDeviceToDevice, b->stream));
                    ss.rows.push_back(r);
                }
                SVA_TRY(dev_alloc(e->allocs, &ss.tok, (size_t)b->e->cfg.tr_dim, false));
                SVA_HIP(hipMemcpyAsync(ss.tok, b->d2c.p + (long)(b->T2 - 1) * b->e->cfg.tr_dim, sizeof(float) * b->e->cfg.tr_dim, hipMemcpyDeviceToDevice, b->stream));
                e->silence[key] = ss;
            } else {
                const sva_engine::SilenceState& ss = it->second;
                for (size_t i = 0; i < sd.size(); ++i) {
                    const ShiftDesc& d = sd[i];
                    hipLaunchKernelGGL(fill_rows_kernel, dim3(d.H, B), dim3(128), 0, b->stream, d.ptr, d.bstride, d.C, ss.rows[i]);
                }
                hipLaunchKernelGGL(fill_rows_kernel, dim3(1, B), dim3(128), 0, b->stream, b->d2c.p + (long)(b->T2 - 1) * b->e->cfg.tr_dim, b->d2c.bstride,
                                   b->e->cfg.tr_dim, ss.tok);
                SVA_HIP(hipGetLastError());
            }
        } else {
            SVA_TRY(zero(S.mel)); SVA_TRY(zero(S.d1)); SVA_TRY(zero(S.d2));
            for (auto& st_ : S.x) for (auto& a : st_) SVA_TRY(zero(a));
            for (int i = 0; i < warm; ++i) SVA_TRY(enc_frontend_stream(b, nullptr, 0, 0));
        }
        // every cached token of a silent window is that same steady response
        hipLaunchKernelGGL(broadcast_row_kernel, dim3(b->T2, B), dim3(256), 0, b->stream, b->d2c.p, b->d2c.bstride, b->T2 - 1, 0, b->T2,
                           b->e->cfg.tr_dim);
        SVA_HIP(hipStreamSynchronize(b->stream));
    }
    // prime the streaming vocoder with the tail of the (truncated) prompt: the reference left-fills its
    // 64-frame vocoder window with the prompt's last frames (:567-571, quirk ix), and the newest frame only
    // depends on the newest 16 code frames, so running the last (window-1) prompt frames through the
    // ring-buffer vocoder reproduces the windowed output.
    SVA_TRY(sva_vocode_reset(b));
    int P = b->p.decode_window_frames - 1;
    for (int i = 0; i < B; ++i) P = std::min(P, b->ref_len[i]);
    P = (P / c) * c;
    SVA_CHECK(P >= 16 || P == b->p.decode_window_frames - 1, "prompt shorter than the vocoder receptive field (16 frames)");
    std::vector<int32_t> codes((size_t)B * ncb * c);
    std::vector<float> sink((size_t)B * 2048 * c);
    for (int f = 0; f < P; f += c) {
        for (int i = 0; i < B; ++i) {
            const int R = b->ref_len[i];
            for (int q = 0; q < ncb; ++q)
                for (int k = 0; k < c; ++k) codes[((size_t)i * ncb + q) * c + k] = b->ref_audio[i][(size_t)q * R + (R - P + f + k)];
        }
        SVA_TRY(sva_vocode_stream(b, codes.data(), c, sink.data()));
    }
    b->begun = true;
    b->steady_eager_steps = 0;     // (a captured graph stays valid: every pointer / constant in it is unchanged)
    return 0;
}

// ============================================================================================
// the per-chunk step
// ============================================================================================
namespace {

int check_mm_overflow(sva_batch* b) {
    if (b->h_mm_ovf && *reinterpret_cast<volatile int*>(b->h_mm_ovf) != 0) {
        *b->h_mm_ovf = 0;
        SVA_CHECK(false, "a batch-scale GEMM on fp16 operand planes produced a non-finite output: an activation or weight is outside the fp16 range "
                         "(as it would be for the reference under torch.autocast(fp16)); create the engine with sva_config.mm_mode = 0 (and voc_dtype = 0) "
                         "for the range-safe bf16 kernels -- the results since the last synchronisation are invalid");
    }
    return 0;
}

// every due slot in one pass over the layers, no host synchronisation (see build_reprefill_kernel)
int reprefill_slots(sva_batch* b, const std::vector<int>& due) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int ncb = c.num_codebooks, d = b->p.delay, bf = b->p.buffer_frames, D = c.ar_dim, nspk = c.timbre_tokens + 1;
    hipStream_t st = b->stream;
    for (size_t lo = 0; lo < due.size(); lo += 128) {
        ReprefillArgs a;
        a.n = (int)std::min<size_t>(128, due.size() - lo);
        a.row_off[0] = 0;
        for (int i = 0; i < a.n; ++i) {
            const int slot = due[lo + i];
            const int nf = b->h_nframes[slot], ncon = b->h_ncontent;
            const int na = std::min(bf, nf);
            SVA_CHECK(na == std::max(0, (ncon - d) - std::max(0, ncon - bf - d)) && na >= d, "re-prefill: content/audio history length mismatch");
            a.slot[i] = slot; a.Rt[i] = b->ref_len[slot]; a.nf[i] = nf; a.na[i] = na;
            a.row_off[i + 1] = a.row_off[i] + 2 * na;
            // BEFORE anything is enqueued: rows past the slot's cache would land in the next slot's
            SVA_CHECK(nspk + 2 * (a.Rt[i] + na) + 2 * d <= c.max_seq_len, "re-prefill: prompt too long for the KV cache");
        }
        const int M = a.row_off[a.n];
        SVA_CHECK(M <= b->Mmax, "re-prefill: more rows than the AR scratch holds");
        hipLaunchKernelGGL(build_reprefill_kernel, dim3(M), dim3(256), 0, st, a, e->content_emb, e->codebook_emb, b->d_content_hist, b->d_pred_hist,
                           b->hist_cap, b->h_ncontent, b->d_ref_tail, c.max_delay, d, ncb, c.codebook_size, D, nspk, b->ax, b->d_slot, b->d_pos);
        SVA_TRY(ar_layers_pass(b, e->ar_layers, M, b->d_slot, b->d_pos, e->rope_ar, (float*)b->kv_slow, b->kv_slow_layer, b->kv_slow_slot,
                               c.max_seq_len, b->ax));
        hipLaunchKernelGGL(finish_reprefill_kernel, dim3(a.n * d), dim3(256), 0, st, a, e->codebook_emb, b->d_pred_hist, b->hist_cap, b->d_ref_tail,
                           c.max_delay, d, ncb, c.codebook_size, D, nspk, b->cached_ref_emb, b->d_last_pos);
        SVA_HIP(hipGetLastError());
        for (int i = 0; i < a.n; ++i) b->h_last_pos[a.slot[i]] = nspk + 2 * (a.Rt[i] + a.na[i]) - 1;
    }
    return 0;
}

int reprefill_slot(sva_batch* b, int slot) {
    // infer_arvc.py:547-564: prompt <- [ref (truncated), last buffer_frames predicted frames] /
    // [ref content, src content[-buffer-d:-d]]; then delay fill with src content[-d:]
    const sva_config& c = b->e->cfg;
    const int ncb = c.num_codebooks, d = b->p.delay, bf = b->p.buffer_frames;
    const int R = b->ref_len[slot];
    const int nf = b->h_nframes[slot], ncon = b->h_ncontent;
    const int na = std::min(bf, nf);                   // pred_codes[..., -bf:]
    const int c_hi = ncon - d, c_lo = std::max(0, ncon - bf - d);
    const int nc = std::max(0, c_hi - c_lo);
    SVA_CHECK(na == nc, "re-prefill: content/audio history length mismatch");
    std::vector<int> hc(nc), ha((size_t)ncb * na);
    SVA_HIP(hipStreamSynchronize(b->stream));
    {   // the histories are rings of hist_cap entries: fetch the slot's rows and index them on the host (rare path)
        const int cap = b->hist_cap;
        std::vector<int> ring(cap);
        if (nc) {
            SVA_HIP(hipMemcpy(ring.data(), b->d_content_hist + (long)slot * cap, sizeof(int) * cap, hipMemcpyDeviceToHost));
            for (int i = 0; i < nc; ++i) hc[i] = ring[(c_lo + i) & (cap - 1)];
        }
        for (int q = 0; q < ncb && na; ++q) {
            SVA_HIP(hipMemcpy(ring.data(), b->d_pred_hist + ((long)slot * ncb + q) * cap, sizeof(int) * cap, hipMemcpyDeviceToHost));
            for (int i = 0; i < na; ++i) ha[(size_t)q * na + i] = ring[(nf - na + i) & (cap - 1)];
        }
    }
    std::vector<int64_t> cc(b->ref_content[slot]);
    for (int i = 0; i < nc; ++i) cc.push_back(hc[i]);
    const int Rn = R + na;
    std::vector<int32_t> ac((size_t)ncb * Rn);
    for (int q = 0; q < ncb; ++q) {
        for (int i = 0; i < R; ++i) ac[(size_t)q * Rn + i] = b->ref_audio[slot][(size_t)q * R + i];
        for (int i = 0; i < na; ++i) ac[(size_t)q * Rn + R + i] = ha[(size_t)q * na + i];
    }
    SVA_TRY(stage_prompt(b, cc, ac, Rn));
    SVA_TRY(ar_prefill_slot(b, slot, Rn));
    return 0;
}

// device work of one steady-state chunk: E0..E8, c AR frames, streaming vocoder, history shift.  Only launches
// on b->stream with device-resident control state, so the sequence can be captured in a hipGraph.
int steady_launches(sva_batch* b, bool timing_events) {
    const sva_config& c = b->e->cfg;
    const int B = b->B, chunk = b->p.chunk_frames, n = 2048 * chunk, ncb = c.num_codebooks;
    hipStream_t st = b->stream;
    if (timing_events) SVA_HIP(hipEventRecord(b->ev[0], st));
    SVA_TRY(launch_ring_write(b->ring, b->d_step, B, b->N, b->step_src ? b->step_src : b->d_chunk, n, st));
    SVA_TRY(b->enc_incremental ? encode_incremental(b, b->d_step, n, 1) : encode(b, b->d_step, n, 1));
    hipLaunchKernelGGL(append_content_kernel, dim3((B + 63) / 64), dim3(64), 0, st, b->d_codes, b->T2, chunk, b->d_content_hist, b->hist_cap,
                       b->d_ncontent, b->d_step_content, B, b->d_step);
    if (timing_events) SVA_HIP(hipEventRecord(b->ev[1], st));
    for (int ci = 0; ci < chunk; ++ci) SVA_TRY(ar_decode_frame(b, ci));          // :534-538
    if (timing_events) SVA_HIP(hipEventRecord(b->ev[2], st));
    b->voc_codes = b->d_step_audio; b->voc_codes_bstride = (long)ncb * chunk; b->voc_codes_gstride = chunk;      // read in place
    const int vrc = vocode(b, chunk, true);
    b->voc_codes = nullptr;
    if (vrc) return vrc;
    if (timing_events) SVA_HIP(hipEventRecord(b->ev[3], st));
    return 0;
}

// Replays `body` (launches on stream st only, all arguments fixed device pointers / constants) as a hipGraph captured on first
// use; falls back to eager launches for good if the runtime cannot capture.
template <class F>
int stage_graph(sva_batch* b, hipGraphExec_t* slot, hipStream_t st, F&& body) {
    if (const int skip = debug_options().pipe_skip) {      // timing diagnostic only (tools/pipe_skip.sh): the chain's work is left out, its events stay
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "[sva] SVA_DEBUG=pipe_skip=%d: chains of the pipelined step are LEFT OUT -- timing diagnostic, outputs are garbage\n", skip); }
        const int bit = (slot == &b->gEm[0] || slot == &b->gEm[1]) ? 1 : slot == &b->gV ? 8 : 2;
        if (skip & bit) return 0;
    }
    if (!b->stage_graphs) return body();
    if (!*slot) {
        hipGraph_t graph = nullptr;
        SVA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = body();
        const hipError_t ce = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess || hipGraphInstantiate(slot, graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            *slot = nullptr;
            b->stage_graphs = false;
            if (graph) (void)hipGraphDestroy(graph);
            return body();
        }
        (void)hipGraphDestroy(graph);
    }
    SVA_HIP(hipGraphLaunch(*slot, st));
    return 0;
}

// Pipelined steady step (sva_step_device with p.pipeline): the three stages of chunk-step n go to three streams,
//   main: E(n)   ->   sa: A(n)   ->   sv: V(n)
// chained by events, so that E(n+1), A(n) and V(n-1) overlap on the GPU.  Hazards between consecutive steps:
//   * d_codes (E writes, A reads to the end of its frame): double-buffered, swapped per step;
//   * d_step_audio (A writes at the end of a frame, V reads it first thing): double-buffered like d_codes, A(n) waits for
//     V(n-2)'s read;
//   * content history (E appends, a re-prefill on sa reads): E(n+1) waits for the re-prefill of step n.
// Every other buffer is private to its stage.
int steady_pipelined(sva_batch* b) {
    const sva_config& c = b->e->cfg;
    const int B = b->B, chunk = b->p.chunk_frames, n = 2048 * chunk, ncb = c.num_codebooks;
    hipStream_t se = b->main_stream, sa = b->sa, sv = b->sv;
    // the stage-boundary events of sva_get_timings mean nothing under overlap and every event record is a barrier packet on its queue (~8 us of
    // the AR chain's period each: profiles/r06_ar_chain_events.txt) -- only the pipe trace keeps them
    const bool stage_ev = b->trace_n > 0;
    if (!b->pipe_dirty) {            // entering the pipelined regime: the side streams must see all serial work so far
        hipEvent_t ev = next_event(b);
        SVA_HIP(hipEventRecord(ev, se));
        SVA_HIP(hipStreamWaitEvent(sa, ev, 0));
        SVA_HIP(hipStreamWaitEvent(sv, ev, 0));
        SVA_HIP(hipMemcpyAsync(b->d_step_x, b->d_step, sizeof(int), hipMemcpyDeviceToDevice, se));      // the side chain's chunk counter
        b->pipe_evVc[0] = b->pipe_evVc[1] = nullptr; b->pipe_evR = nullptr; b->pipe_evA[0] = b->pipe_evA[1] = nullptr; b->pipe_evD2C = nullptr;
        b->pipe_evFeat[0] = b->pipe_evFeat[1] = nullptr;
    }
    const int par = b->pipe_parity;
    b->d_codes = b->d_codes_buf[par];
    b->d_step_audio = b->d_step_audio_buf[par];      // the frame's audio codes: A(n) writes them last, V(n) reads them first
    b->pipe_parity ^= 1;
    hipEvent_t evE = nullptr;
    // debug trace: k = 0 main start, 1 main end, 2 side start, 3 transformer start, 4 transformer end, 5 AR start, 6 AR end,
    // 7 vocoder start, 8 vocoder end (each recorded behind the waits of its chain, i.e. when the chain could really start)
    auto mark = [&](int k, hipStream_t st_) -> int {
        if (b->trace_n) SVA_HIP(hipEventRecord(b->trace_ev[(size_t)(b->trace_steps % b->trace_n) * 9 + k], st_));
        return 0;
    };
    if (b->pipe_split_e && b->enc_incremental && b->concurrency && b->aux[0]) {
        // The encoder is three chains -- head pass (main), streaming pass (side stream sx) and the 8-layer transformer + BSQ,
        // which needs both -- and as one in-order stage its latency (1.7 ms) was the period of the whole pipeline.  The
        // transformer therefore moves to the side stream behind the streaming pass:
        //   main: slide d2c, head pass(n)                         -> evHead
        //   sx:   streaming pass(n), [evHead] transformer + BSQ(n) -> evE
        // so head pass(n+1) overlaps transformer(n) and the encoder's period is the side stream's chain.  Hazards: the token
        // cache d2c (written by the slide, the head rows and the tail rows; read by the transformer in its FIRST layer only --
        // QKV operand and the residual of the output projection): main(n+1) waits for an event the transformer records
        // after that layer, and sx is in order; the chunk counter (ring position) is read by the first kernel of each
        // front-end chain, so each chain advances its own copy when it is done (the append no longer does).
        hipStream_t sx = b->aux[0];
        if (b->pipe_evD2C && !b->enc_merged) SVA_HIP(hipStreamWaitEvent(se, b->pipe_evD2C, 0));
        b->stream = se;
        if (stage_ev) SVA_HIP(hipEventRecord(b->ev[0], se));
        SVA_TRY(mark(0, se));
        if (b->enc_merged) {
            // ONE front-end chain on the main stream (the new frames ride in the head-pass launches); the side stream only carries
            // the transformer, which overlaps the next step's front-end.  Both chains replay as hipGraphs (stage_graph).
            if (b->enc_cut) {
                // Balanced cut: the backbone (STFT .. final LayerNorm) is the main chain, the quantizer's downsampler joins the
                // transformer on the side chain -- both near the AR frame time, and the token cache d2c is then written and read on
                // ONE in-order stream.  The hand-over buffer feat[] alternates; main(n+2) waits until side(n) has read it.
                if (b->pipe_evFeat[par]) SVA_HIP(hipStreamWaitEvent(se, b->pipe_evFeat[par], 0));
                SVA_TRY(stage_graph(b, &b->gEm[par], se, [&]() -> int {
                    b->stream = se;
                    SVA_TRY(launch_ring_write(b->ring, b->d_step, B, b->N, b->step_src ? b->step_src : b->d_chunk, n, se));
                    b->step_bump = b->d_step;              // the history shift that ends the chain also advances the chunk counter
                    const int frc = enc_frontend_merged(b, b->d_step, n, 1, 3, par);
                    b->step_bump = nullptr;
                    return frc;
                }));
                SVA_TRY(mark(1, se));
                SVA_TRY(mark(2, sx));
                SVA_TRY(stream_fork(b, se, sx));
                int src_ = stage_graph(b, &b->gEs[par], sx, [&]() -> int {
                    b->stream = sx;
                    SVA_TRY(launch_shift_history(b->d_shift_d2c, 1, B, sx, 16));               // steady tokens slide down by c
                    return enc_frontend_merged(b, nullptr, n, 1, 4, par);
                });
                b->stream = se;
                if (src_) return src_;
                b->pipe_evFeat[par] = next_event(b);
                SVA_HIP(hipEventRecord(b->pipe_evFeat[par], sx));
            } else {
            // the token cache d2c is only touched by the last three kernels of the front-end (slide, head + new tokens in, history
            // shift): only they wait for transformer(n-1)'s first layer -- the front-end of step n itself starts at once
            SVA_TRY(stage_graph(b, &b->gE, se, [&]() -> int {
                b->stream = se;
                SVA_TRY(launch_ring_write(b->ring, b->d_step, B, b->N, b->step_src ? b->step_src : b->d_chunk, n, se));
                return enc_frontend_merged(b, b->d_step, n, 1, 1);
            }));
            if (b->pipe_evD2C) SVA_HIP(hipStreamWaitEvent(se, b->pipe_evD2C, 0));
            SVA_TRY(stage_graph(b, &b->gE2, se, [&]() -> int {
                b->stream = se;
                SVA_TRY(launch_shift_history(b->d_shift_d2c, 1, B, se, 16));                   // steady tokens slide down by c
                SVA_TRY(enc_frontend_merged(b, b->d_step, n, 1, 2));
                return launch_add_i32(b->d_step, 1, se);
            }));
            SVA_TRY(mark(1, se));
            SVA_TRY(mark(2, sx));
            SVA_TRY(stream_fork(b, se, sx));                                              // transformer(n) needs the front-end of step n
            }
            if (b->pipe_evA[par]) SVA_HIP(hipStreamWaitEvent(sx, b->pipe_evA[par], 0));  // back-pressure: BSQ overwrites the codes A(n-2) read
            if (b->pipe_evR) { SVA_HIP(hipStreamWaitEvent(sx, b->pipe_evR, 0)); b->pipe_evR = nullptr; }
            SVA_TRY(mark(3, sx));
            b->tr_l0_event = nullptr;
            int trc = stage_graph(b, &b->gT0, sx, [&]() -> int { b->stream = sx; return enc_transformer(b, b->d2c, chunk, 1); });
            b->stream = se;
            if (trc) return trc;
            b->pipe_evD2C = next_event(b);                                                // the token cache is not read past the first layer
            SVA_HIP(hipEventRecord(b->pipe_evD2C, sx));
            trc = stage_graph(b, &b->gT1[par], sx, [&]() -> int {
                b->stream = sx;
                SVA_TRY(enc_transformer(b, b->d2c, chunk, 2));
                hipLaunchKernelGGL(append_content_kernel, dim3((B + 63) / 64), dim3(64), 0, sx, b->d_codes, b->T2, chunk, b->d_content_hist, b->hist_cap,
                                   b->d_ncontent, b->d_step_content, B, (int*)nullptr);
                SVA_HIP(hipGetLastError());
                return 0;
            });
            b->stream = se;
            if (trc) return trc;
            if (stage_ev) SVA_HIP(hipEventRecord(b->ev[1], sx));
            SVA_TRY(mark(4, sx));
        } else {
        SVA_TRY(launch_ring_write(b->ring, b->d_step, B, b->N, b->step_src ? b->step_src : b->d_chunk, n, se));
        SVA_TRY(launch_shift_history(b->d_shift_d2c, 1, B, se, 16));                       // steady tokens slide down by c
        int erc = 0;
        if (b->stream_cut > 0) {          // the first stages of the streaming pass run on main (its chunk counter), the rest on sx
            SVA_TRY(enc_frontend_stream(b, b->d_step, n, 1, 1));
            SVA_TRY(stream_fork(b, se, sx));
            b->stream = sx;
            SVA_TRY(mark(2, sx));
            erc = enc_frontend_stream(b, b->d_step, n, 1, 2);
            if (!erc) erc = launch_add_i32(b->d_step_x, 1, sx);                       // (kept in step although this layout does not read it)
        } else {
            SVA_TRY(stream_fork(b, se, sx));
            b->stream = sx;
            SVA_TRY(mark(2, sx));
            erc = enc_frontend_stream(b, b->d_step_x, n, 1);                          // c newest tokens -> d2c tail
            if (!erc) erc = launch_add_i32(b->d_step_x, 1, sx);
        }
        b->stream = se;
        if (erc) return erc;
        SVA_TRY(enc_frontend_window(b, b->d_step, n, 1, 4 * b->Ht, nullptr, &b->d2c));     // head pass -> d2c rows [0, Ht)
        SVA_TRY(launch_add_i32(b->d_step, 1, se));
        SVA_TRY(mark(1, se));
        SVA_TRY(stream_fork(b, se, sx));                                              // transformer(n) needs head pass(n)
        // back-pressure: this step's BSQ overwrites the code buffer that A(n-2) read, so E may lead A by at most two steps
        if (b->pipe_evA[par]) SVA_HIP(hipStreamWaitEvent(sx, b->pipe_evA[par], 0));
        if (b->pipe_evR) { SVA_HIP(hipStreamWaitEvent(sx, b->pipe_evR, 0)); b->pipe_evR = nullptr; }
        b->stream = sx;
        SVA_TRY(mark(3, sx));
        b->pipe_evD2C = next_event(b);
        b->tr_l0_event = b->pipe_evD2C;
        const int trc = enc_transformer(b, b->d2c, chunk);
        b->tr_l0_event = nullptr;
        b->stream = se;
        if (trc) return trc;
        hipLaunchKernelGGL(append_content_kernel, dim3((B + 63) / 64), dim3(64), 0, sx, b->d_codes, b->T2, chunk, b->d_content_hist, b->hist_cap,
                           b->d_ncontent, b->d_step_content, B, (int*)nullptr);
        if (stage_ev) SVA_HIP(hipEventRecord(b->ev[1], sx));
        SVA_TRY(mark(4, sx));
        }
        if (b->pipe_evVc[par]) SVA_HIP(hipStreamWaitEvent(sx, b->pipe_evVc[par], 0));     // V(n-2) has read the audio-code buffer A(n) will overwrite (long since)
        evE = next_event(b);
        SVA_HIP(hipEventRecord(evE, sx));
    } else {
        // back-pressure: this step's encoder overwrites the code buffer that A(n-2) read, so E may lead A by at most two steps
        if (b->pipe_evA[par]) SVA_HIP(hipStreamWaitEvent(se, b->pipe_evA[par], 0));
        if (b->pipe_evR) { SVA_HIP(hipStreamWaitEvent(se, b->pipe_evR, 0)); b->pipe_evR = nullptr; }
        b->stream = se;
        if (stage_ev) SVA_HIP(hipEventRecord(b->ev[0], se));
        SVA_TRY(launch_ring_write(b->ring, b->d_step, B, b->N, b->step_src ? b->step_src : b->d_chunk, n, se));
        SVA_TRY(b->enc_incremental ? encode_incremental(b, b->d_step, n, 1) : encode(b, b->d_step, n, 1));
        hipLaunchKernelGGL(append_content_kernel, dim3((B + 63) / 64), dim3(64), 0, se, b->d_codes, b->T2, chunk, b->d_content_hist, b->hist_cap,
                           b->d_ncontent, b->d_step_content, B, b->d_step);
        if (stage_ev) SVA_HIP(hipEventRecord(b->ev[1], se));
        if (b->pipe_evVc[par]) SVA_HIP(hipStreamWaitEvent(se, b->pipe_evVc[par], 0));
        evE = next_event(b);
        SVA_HIP(hipEventRecord(evE, se));
    }
    // A(n)
    SVA_HIP(hipStreamWaitEvent(sa, evE, 0));       // (evE also carries "V(n-2) has read the code buffer": the encoder chain waited for that event above --
                                                   //  one barrier packet fewer on the AR queue, whose chain is the period of a one-stream pipeline)
    b->stream = sa;
    SVA_TRY(mark(5, sa));
    int rc = 0;
    if (debug_options().pipe_skip & 4) {
    } else if (b->pipe_graph_mode && (debug_options().ar_graph >= 0 ? debug_options().ar_graph != 0 : true)) {
        // The AR stage is ~210 launches on ONE stream whose arguments never change (positions, frame counters, noise keys
        // and teacher-forcing flags live in device memory; the code buffer alternates, hence one graph per parity): replaying
        // it as a hipGraph takes those launches off the enqueueing thread, whose launch rate is otherwise as tight a bound
        // on a single-stream step as the GPU (1.0-2.1 ms per step depending on the host core, vs 1.75 ms of GPU time).
        if (!b->pipe_graph_a[par]) {
            hipGraph_t graph = nullptr;
            SVA_HIP(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
            for (int ci = 0; ci < chunk && !rc; ++ci) rc = ar_decode_frame(b, ci);
            const hipError_t ce = hipStreamEndCapture(sa, &graph);
            if (rc) { b->stream = se; return rc; }
            if (ce != hipSuccess || hipGraphInstantiate(&b->pipe_graph_a[par], graph, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();                   // capture unavailable: enqueue this and every later AR stage kernel by kernel
                b->pipe_graph_a[par] = nullptr;
                b->pipe_graph_mode = 0;
            }
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (b->pipe_graph_a[par]) {
            // the captured stage's persistent launches carry no cross-batch ordering of their own: chain the replay as a whole
            PersistentChain chain(b, sa, (b->use_mega || b->use_abatch) && !b->edits_on);
            rc = chain.rc;
            if (!rc) {
                SVA_HIP(hipGraphLaunch(b->pipe_graph_a[par], sa));
                rc = chain.finish();
            }
        } else for (int ci = 0; ci < chunk && !rc; ++ci) rc = ar_decode_frame(b, ci);
    } else {
        for (int ci = 0; ci < chunk && !rc; ++ci) rc = ar_decode_frame(b, ci);
    }
    if (rc) { b->stream = se; return rc; }
    if (stage_ev) SVA_HIP(hipEventRecord(b->ev[2], sa));
    SVA_TRY(mark(6, sa));
    hipEvent_t evA = next_event(b);
    SVA_HIP(hipEventRecord(evA, sa));
    b->pipe_evA[par] = evA;
    // V(n)
    SVA_HIP(hipStreamWaitEvent(sv, evA, 0));
    b->stream = sv;
    SVA_TRY(mark(7, sv));
    b->voc_codes = b->d_step_audio; b->voc_codes_bstride = (long)ncb * chunk; b->voc_codes_gstride = chunk;      // read in place ...
    b->pipe_evVc[par] = next_event(b);
    b->voc_codes_event = b->pipe_evVc[par];       // ... and A(n+1) may overwrite them once the FSQ decode has run
    if (b->stage_graphs && !b->pcm_dst) {
        rc = vocode(b, chunk, true, 3);           // FSQ decode (reads this parity's code buffer) + the release event, eagerly
        b->voc_codes_event = nullptr;
        if (!rc) rc = stage_graph(b, &b->gV, sv, [&]() -> int { b->stream = sv; return vocode(b, chunk, true, 4); });
    } else {
        rc = vocode(b, chunk, true);
    }
    b->voc_codes = nullptr; b->voc_codes_event = nullptr;
    b->stream = se;
    if (rc) return rc;
    if (stage_ev) SVA_HIP(hipEventRecord(b->ev[3], sv));
    SVA_TRY(mark(8, sv));
    b->trace_steps += 1;
    b->pipe_dirty = true;
    b->out_stream = sv;
    return 0;
}

int quiesce(sva_batch* b) {
    if (!b || !b->pipe_dirty) return 0;
    hipStream_t se = b->main_stream;
    hipEvent_t e1 = next_event(b), e2 = next_event(b);
    SVA_HIP(hipEventRecord(e1, b->sa));
    SVA_HIP(hipEventRecord(e2, b->sv));
    SVA_HIP(hipStreamWaitEvent(se, e1, 0));
    SVA_HIP(hipStreamWaitEvent(se, e2, 0));
    if (b->aux[0]) {                       // (carries the encoder's transformer in the pipelined regime)
        hipEvent_t e3 = next_event(b);
        SVA_HIP(hipEventRecord(e3, b->aux[0]));
        SVA_HIP(hipStreamWaitEvent(se, e3, 0));
    }
    b->pipe_dirty = false;
    b->pipe_evVc[0] = b->pipe_evVc[1] = nullptr; b->pipe_evR = nullptr; b->pipe_evA[0] = b->pipe_evA[1] = nullptr; b->pipe_evD2C = nullptr;
        b->pipe_evFeat[0] = b->pipe_evFeat[1] = nullptr;
    b->out_stream = se;
    return 0;
}

int step_body(sva_batch* b) {
    sva_engine* e = b->e;
    (void)e;
    if (b->h_ar_fail && *reinterpret_cast<volatile int*>(b->h_ar_fail) != 0) b->ar_failed = true;      // (a launch of an earlier step saw a wait time out)
    SVA_CHECK(!b->ar_failed, "this batch's persistent AR decode kernel timed out earlier: call sva_streams_begin to restart its streams");
    // callers of the stream-ordered API may never call sva_sync: the fp16-planes overflow flag is polled here too
    SVA_TRY(check_mm_overflow(b));
    const int B = b->B, chunk = b->p.chunk_frames, d = b->p.delay, n = 2048 * chunk;
    hipStream_t st = b->stream;
    const bool steady = b->delay_filled && (b->h_ncontent + chunk >= d);
    if (steady) {
        const bool graph_ok = b->p.use_graph && b->noise_on_device && !b->forced_now && !b->prof_on;
        const bool pipe_ok = b->p.pipeline && b->sa && b->allow_pipe && b->noise_on_device && !b->forced_now && !b->prof_on && !b->p.use_graph &&
                             b->voc_grouped && b->steady_eager_steps >= 2;
        if (!pipe_ok) SVA_TRY(quiesce(b));
        if (pipe_ok) {
            SVA_TRY(steady_pipelined(b));
            b->graph_step = false;
        } else if (graph_ok && b->steady_eager_steps >= 2) {
            if (!b->graph_ready) {
                // capture once: every launch argument is a fixed device pointer or a constant; per-step variation
                // (ring position, KV positions, frame counters, sampler noise keys) lives in device memory
                hipGraph_t graph = nullptr;
                SVA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                int rc = steady_launches(b, false);
                hipError_t ce = hipStreamEndCapture(st, &graph);
                if (rc) return rc;
                SVA_HIP(ce);
                SVA_HIP(hipGraphInstantiate(&b->graph_exec, graph, nullptr, nullptr, 0));
                SVA_HIP(hipGraphDestroy(graph));
                b->graph_ready = true;
            }
            SVA_HIP(hipEventRecord(b->ev[0], st));
            {
                PersistentChain chain(b, st, (b->use_mega || b->use_abatch) && !b->edits_on);       // (as for the captured AR stage of the pipelined mode)
                SVA_TRY(chain.rc);
                SVA_HIP(hipGraphLaunch(b->graph_exec, st));
                SVA_TRY(chain.finish());
            }
            SVA_HIP(hipEventRecord(b->ev[3], st));
            b->graph_step = true;
        } else {
            SVA_TRY(steady_launches(b, true));
            b->steady_eager_steps += 1;
            b->graph_step = false;
        }
        b->h_step += 1;
        b->h_ncontent += chunk;
        for (int i = 0; i < B; ++i) { b->h_last_pos[i] += 2 * chunk; b->h_nframes[i] += chunk; }
        // re-prefill when current_pos // 2 >= max_seq_frames (:547-564).  Positions are deterministic, so the host
        // mirror decides without a device round trip; doing it after the vocoder instead of before (as the reference
        // does) changes nothing: the vocoder consumes the codes just decoded, the re-prefill only rewrites KV state.
        std::vector<int> redo;
        if (b->pipe_dirty) b->stream = b->sa;  // pipelined: the KV rewrite belongs to the AR stream (it already waited for E of this step)
        int rrc = 0;
        for (int i = 0; i < B; ++i)
            if (b->h_last_pos[i] / 2 >= b->p.max_seq_frames) redo.push_back(i);
        if (!redo.empty()) {
            // one pass for all due slots against the cached prompt prefix (reprefill_slots); SVA_DEBUG reprefill=0: the round-3 path, one
            // whole-prompt prefill per slot behind a host synchronisation (A/B, parity of the two)
            // (a stream that has decoded fewer than `delay` frames -- a prompt about as long as max_seq_frames -- keeps the general form)
            bool one_pass = debug_options().reprefill != 0;
            for (int s_ : redo) one_pass = one_pass && std::min(b->p.buffer_frames, b->h_nframes[s_]) >= b->p.delay;
            if (one_pass) rrc = reprefill_slots(b, redo);
            else for (size_t i = 0; i < redo.size() && !rrc; ++i) rrc = reprefill_slot(b, redo[i]);
        }
        if (!rrc) rrc = ar_delay_fill(b, redo);       // prefill_src_condition4delay(src_content_codes[-d:]) for those slots only
        if (b->pipe_dirty) {
            b->stream = b->main_stream;
            if (!rrc && !redo.empty()) {       // E of the next step must not append to the content history before this has read it
                b->pipe_evR = next_event(b);
                SVA_HIP(hipEventRecord(b->pipe_evR, b->sa));
            }
        }
        return rrc;
    }
    b->graph_step = false;
    SVA_TRY(quiesce(b));
    SVA_HIP(hipEventRecord(b->ev[0], st));
    // E0: shift window / append chunk (:495-496), then E1..E8
    SVA_TRY(launch_ring_write(b->ring, b->d_step, B, b->N, b->step_src ? b->step_src : b->d_chunk, n, st));
    SVA_TRY(b->enc_incremental ? encode_incremental(b, b->d_step, n, 1) : encode(b, b->d_step, n, 1));
    hipLaunchKernelGGL(append_content_kernel, dim3((B + 63) / 64), dim3(64), 0, st, b->d_codes, b->T2, chunk, b->d_content_hist, b->hist_cap,
                       b->d_ncontent, b->d_step_content, B, b->d_step);
    b->h_step += 1;
    b->h_ncontent += chunk;
    SVA_HIP(hipEventRecord(b->ev[1], st));
    if (b->h_ncontent >= d && !b->delay_filled) {
        SVA_TRY(ar_delay_fill(b));                      // :521-525 -> zeros
        b->delay_filled = true;
    }                                                   // else :519-520 -> zeros
    SVA_HIP(hipEventRecord(b->ev[2], st));
    SVA_HIP(hipMemsetAsync(b->d_pcm, 0, sizeof(float) * (size_t)B * 2048 * b->Tv, st));
    SVA_HIP(hipEventRecord(b->ev[3], st));
    return 0;
}

}  // namespace

static int check_ar_fail(sva_batch* b);

extern "C" int sva_step(sva_batch* b, const float* pcm_in, float* pcm_out, const float* noise, const int32_t* forced_codes) {
    SVA_CHECK(b && pcm_in && pcm_out && b->begun, "sva_step: bad argument or sva_streams_begin not called");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    const sva_config& c = b->e->cfg;
    const int B = b->B, chunk = b->p.chunk_frames, n = 2048 * chunk, ncb = c.num_codebooks;
    hipStream_t st = b->stream;
    memcpy(b->hp_in, pcm_in, sizeof(float) * (size_t)B * n);
    SVA_HIP(hipMemcpyAsync(b->d_chunk, b->hp_in, sizeof(float) * (size_t)B * n, hipMemcpyHostToDevice, st));
    b->noise_on_device = (noise == nullptr);
    if (noise)
        SVA_HIP(hipMemcpyAsync(b->d_noise, noise, sizeof(float) * (size_t)B * chunk * (c.ar_vocab + ncb * c.codebook_size), hipMemcpyHostToDevice, st));
    b->forced_now = forced_codes != nullptr;
    const int uf = forced_codes ? 1 : 0;
    SVA_HIP(hipMemcpyAsync(b->d_use_forced, &uf, sizeof(int), hipMemcpyHostToDevice, st));
    b->h_use_forced = uf;
    if (forced_codes) SVA_HIP(hipMemcpyAsync(b->d_forced, forced_codes, sizeof(int) * (size_t)B * ncb * chunk, hipMemcpyHostToDevice, st));
    SVA_HIP(hipStreamSynchronize(st));
    b->gemm_flops = 0; b->gemm_launches = 0; b->gemm_bytes = 0;
    SVA_TRY(step_body(b));
    SVA_HIP(hipMemcpy2DAsync(b->hp_out, sizeof(float) * n, b->d_pcm, sizeof(float) * 2048 * b->Tv, sizeof(float) * n, B, hipMemcpyDeviceToHost, st));
    SVA_HIP(hipStreamSynchronize(st));
    SVA_TRY(check_ar_fail(b));
    memcpy(pcm_out, b->hp_out, sizeof(float) * (size_t)B * n);
    float t;
    if (!b->graph_step)
        for (int i = 0; i < 3; ++i) {
            if (hipEventElapsedTime(&t, b->ev[i], b->ev[i + 1]) == hipSuccess) b->last_ms[i] = t;
        }
    if (hipEventElapsedTime(&t, b->ev[0], b->ev[3]) == hipSuccess) b->last_ms[3] = t;
    return 0;
}

extern "C" int sva_step_device(sva_batch* b, const float* d_pcm_in, float* d_pcm_out) {
    SVA_CHECK(b && d_pcm_in && d_pcm_out && b->begun, "sva_step_device: bad argument");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    const int B = b->B, n = 2048 * b->p.chunk_frames;
    hipStream_t st = b->main_stream;
    b->stream = st;
    // no staging: the ring write reads the caller's chunk, the vocoder's last kernel writes the caller's PCM buffer and
    // the FSQ decode reads the step's codes in place (each staging copy was a kernel of its own in a chain whose length is
    // what bounds the step)
    const bool direct = !b->p.use_graph && !(b->p.pipeline && b->stage_graphs);       // a captured graph bakes its pointers: it keeps the fixed staging buffers
    if (direct) b->step_src = d_pcm_in;
    else SVA_HIP(hipMemcpyAsync(b->d_chunk, d_pcm_in, sizeof(float) * (size_t)B * n, hipMemcpyDeviceToDevice, st));
    b->noise_on_device = true;
    b->forced_now = false;
    if (b->h_use_forced != 0) { SVA_HIP(hipMemsetAsync(b->d_use_forced, 0, sizeof(int), st)); b->h_use_forced = 0; }
    b->gemm_flops = 0; b->gemm_launches = 0; b->gemm_bytes = 0;
    b->allow_pipe = true;
    b->out_stream = st;
    b->pcm_dst = direct ? d_pcm_out : nullptr; b->pcm_dst_bstride = n; b->pcm_direct_done = false;
    const int rc = step_body(b);
    b->allow_pipe = false;
    b->step_src = nullptr; b->pcm_dst = nullptr;
    if (rc) return rc;
    if (!b->pcm_direct_done)     // delay warm-up steps run no vocoder: their zeros come from d_pcm
        SVA_HIP(hipMemcpy2DAsync(d_pcm_out, sizeof(float) * n, b->d_pcm, sizeof(float) * 2048 * b->Tv, sizeof(float) * n, B, hipMemcpyDeviceToDevice,
                                 b->out_stream));
    return 0;
}

// Stream-ordered variants: the reference's process_one_chunk runs on the caller's current stream, so whatever filled the input
// buffer is ordered before it and whatever reads the output after it (evaluations/infer_arvc.py:495-508 on torch's stream).  The
// engine keeps streams of its own; these entry points tie them to the caller's stream with events instead of host synchronisation.
extern "C" int sva_step_device_on(sva_batch* b, const float* d_pcm_in, float* d_pcm_out, void* caller_stream, int join_output) {
    SVA_CHECK(b && b->begun, "sva_step_device_on: bad argument");
    SVA_HIP(hipSetDevice(b->e->device));
    hipStream_t cs = (hipStream_t)caller_stream;
    {   // everything already enqueued on the caller's stream happens before the engine's first access to d_pcm_in
        hipEvent_t ev = next_event(b);
        SVA_HIP(hipEventRecord(ev, cs));
        SVA_HIP(hipStreamWaitEvent(b->main_stream, ev, 0));
    }
    SVA_TRY(sva_step_device(b, d_pcm_in, d_pcm_out));
    if (join_output) return sva_join_stream(b, caller_stream);
    return 0;
}

extern "C" int sva_join_stream(sva_batch* b, void* caller_stream) {
    SVA_CHECK(b, "null batch");
    SVA_HIP(hipSetDevice(b->e->device));
    hipStream_t cs = (hipStream_t)caller_stream;
    // the caller's stream waits (on the device, not the host) for every chain of the engine: the stage streams of a pipelined batch,
    // the encoder's side stream and the main stream; the pipelined regime itself is left untouched
    hipStream_t all[4] = {b->main_stream, b->p.pipeline ? b->sa : nullptr, b->p.pipeline ? b->sv : nullptr, b->aux[0]};
    for (hipStream_t st_ : all) {
        if (!st_) continue;
        hipEvent_t ev = next_event(b);
        SVA_HIP(hipEventRecord(ev, st_));
        SVA_HIP(hipStreamWaitEvent(cs, ev, 0));
    }
    if (b->out_stream && b->out_stream != b->main_stream) {
        hipEvent_t ev = next_event(b);
        SVA_HIP(hipEventRecord(ev, b->out_stream));
        SVA_HIP(hipStreamWaitEvent(cs, ev, 0));
    }
    return 0;
}

extern "C" int sva_stream_chunks(sva_batch* b, const float* pcm_in, float* pcm_out, int n_chunks) {
    SVA_CHECK(b && pcm_in && pcm_out && b->begun && n_chunks >= 1, "sva_stream_chunks: bad argument");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    const int B = b->B, n = 2048 * b->p.chunk_frames;
    const size_t per = (size_t)B * n;
    float *d_in = nullptr, *d_out = nullptr;
    SVA_HIP(hipMalloc((void**)&d_in, sizeof(float) * per * n_chunks));
    SVA_HIP(hipMalloc((void**)&d_out, sizeof(float) * per * n_chunks));
    hipStream_t st = b->main_stream;
    // [B][n_chunks*n] on the host <-> [n_chunks][B][n] on the device (one contiguous block per step)
    int rc = 0;
    for (int k = 0; k < n_chunks && !rc; ++k)
        if (hipMemcpy2DAsync(d_in + per * k, sizeof(float) * n, pcm_in + (size_t)k * n, sizeof(float) * n * n_chunks, sizeof(float) * n, B,
                             hipMemcpyHostToDevice, st) != hipSuccess) { set_error("sva_stream_chunks: upload failed"); rc = -2; }
    for (int k = 0; k < n_chunks && !rc; ++k) rc = sva_step_device(b, d_in + per * k, d_out + per * k);
    if (!rc) rc = sva_sync(b);
    for (int k = 0; k < n_chunks && !rc; ++k)
        if (hipMemcpy2DAsync(pcm_out + (size_t)k * n, sizeof(float) * n * n_chunks, d_out + per * k, sizeof(float) * n, sizeof(float) * n, B,
                             hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("sva_stream_chunks: download failed"); rc = -2; }
    if (hipStreamSynchronize(st) != hipSuccess && !rc) { set_error("sva_stream_chunks: sync failed"); rc = -2; }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

// the persistent decode kernel gives up on a hand-off that does not arrive (bounded spin) and records where: surface it at the
// next synchronisation instead of returning garbage codes
static int check_ar_fail(sva_batch* b) {
    if ((!b->use_mega && !b->use_abatch) || !b->d_ar_fail) return 0;
    int f = 0;
    SVA_HIP(hipMemcpy(&f, b->d_ar_fail, sizeof(int), hipMemcpyDeviceToHost));
    if (f != 0) b->ar_failed = true;
    SVA_CHECK(f == 0, "persistent AR decode kernel timed out waiting for a workgroup hand-off (code " + std::to_string(f) +
                      "): its workgroups were not all resident (GPU shared with another process / a long kernel?).  The frames since the last "
                      "synchronisation are invalid; sva_streams_begin restarts the streams on the multi-launch decode");
    return 0;
}

// test / tool hook: change debug options of this process after start-up ("key=value,..." as in SVA_DEBUG; batches created afterwards
// see them -- the GEMM dispatch table and the autotune switch are read once, before the first launch)
extern "C" int sva_debug_configure(const char* kv) {
    SVA_CHECK(kv, "null argument");
    parse_debug(sva::debug_options_mut(), kv);
    return 0;
}

// test hook: make the batch look as if its persistent decode kernel had timed out (sets the device-side fail word)
extern "C" int sva_test_force_ar_timeout(sva_batch* b) {
    SVA_CHECK(b && (b->use_mega || b->use_abatch) && b->d_ar_fail, "sva_test_force_ar_timeout: the batch does not use the persistent decode kernel");
    SVA_HIP(hipSetDevice(b->e->device));
    SVA_TRY(quiesce(b));
    SVA_HIP(hipStreamSynchronize(b->stream));
    const int code = 99;
    SVA_HIP(hipMemcpy(b->d_ar_fail, &code, sizeof(int), hipMemcpyHostToDevice));
    return 0;
}
extern "C" int sva_batch_uses_persistent_decode(sva_batch* b) { return b && !b->edits_on ? (b->use_mega ? 1 : b->use_abatch ? 2 : 0) : 0; }

// previous_tokens / repetition_penalty / suppress_tokens of decode_one_token_ar (modules/dual_ar_stream.py:1099-1117, 1175-1213)
extern "C" int sva_set_sampler_edits(sva_batch* b, const int32_t* previous_tokens, int W, float repetition_penalty, const int32_t* suppress_tokens,
                                     int n_suppress) {
    SVA_CHECK(b, "sva_set_sampler_edits: null batch");
    SVA_CHECK(W >= 0 && W <= EDIT_CAP && n_suppress >= 0 && n_suppress <= EDIT_CAP, "sva_set_sampler_edits: at most 4096 previous tokens per head / suppressed tokens");
    SVA_CHECK((W == 0 || previous_tokens) && (n_suppress == 0 || suppress_tokens), "sva_set_sampler_edits: null list");
    SVA_CHECK(repetition_penalty > 0.f, "sva_set_sampler_edits: repetition_penalty must be positive");
    SVA_HIP(hipSetDevice(b->e->device));
    SVA_TRY(quiesce(b));
    SVA_HIP(hipStreamSynchronize(b->stream));
    const int heads = 1 + b->e->cfg.num_codebooks;
    const bool on = W > 0 || n_suppress > 0;
    if (on && !b->d_edit_prev) {
        SVA_TRY(dev_alloc(b->allocs, &b->d_edit_prev, (size_t)heads * EDIT_CAP));
        SVA_TRY(dev_alloc(b->allocs, &b->d_edit_suppress, (size_t)EDIT_CAP));
        SVA_TRY(dev_alloc(b->allocs, &b->d_edit_params, (size_t)heads * 4));
    }
    if (on) {
        for (int h = 0; h < heads && W > 0; ++h)
            SVA_HIP(hipMemcpy(b->d_edit_prev + (size_t)h * EDIT_CAP, previous_tokens + (size_t)h * W, sizeof(int) * (size_t)W, hipMemcpyHostToDevice));
        if (n_suppress) SVA_HIP(hipMemcpy(b->d_edit_suppress, suppress_tokens, sizeof(int) * (size_t)n_suppress, hipMemcpyHostToDevice));
        std::vector<int> prm((size_t)heads * 4, 0);
        int pbits;
        memcpy(&pbits, &repetition_penalty, 4);
        for (int h = 0; h < heads; ++h) { prm[h * 4] = W; prm[h * 4 + 1] = h == 0 ? n_suppress : 0; prm[h * 4 + 2] = pbits; }     // the list goes to the token head only (:1189)
        SVA_HIP(hipMemcpy(b->d_edit_params, prm.data(), sizeof(int) * prm.size(), hipMemcpyHostToDevice));
    }
    if (on != b->edits_on) {
        // the AR stage's captured launches change (edit kernels in / out, persistent kernel out / in): drop its graphs
        for (auto& ge : b->pipe_graph_a) if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
        if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
        b->edits_on = on;
    }
    return 0;
}

extern "C" int sva_sync(sva_batch* b) {
    SVA_CHECK(b, "null batch");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    SVA_HIP(hipStreamSynchronize(b->stream));
    SVA_TRY(check_ar_fail(b));
    SVA_TRY(conv_gemm_check_errors());
    SVA_TRY(check_mm_overflow(b));
    float t;
    if (!b->graph_step)
        for (int i = 0; i < 3; ++i)
            if (hipEventElapsedTime(&t, b->ev[i], b->ev[i + 1]) == hipSuccess) b->last_ms[i] = t;
    if (hipEventElapsedTime(&t, b->ev[0], b->ev[3]) == hipSuccess) b->last_ms[3] = t;
    return 0;
}

// ---- AR seams: DualARWrapper.prefill_src_condition4delay / decode_one driven with caller-supplied content codes ----
__global__ void put_codes_kernel(const long long* __restrict__ src, int n_per, long long* __restrict__ codes, int T2, int* __restrict__ content_hist,
                                 int hist_cap, int* __restrict__ ncontent, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = ncontent[b];
    for (int i = 0; i < n_per; ++i) {
        const long long c = src[(long)b * n_per + i];
        codes[(long)b * T2 + T2 - n_per + i] = c;
        content_hist[(long)b * hist_cap + ((n + i) & (hist_cap - 1))] = (int)c;
    }
    ncontent[b] = n + n_per;
}

extern "C" int sva_ar_delay_fill(sva_batch* b, const int64_t* codes) {
    SVA_CHECK(b && codes && b->begun, "sva_ar_delay_fill: bad argument");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    const int B = b->B, d = b->p.delay;
    long long* tmp = (long long*)b->d_noise;       // scratch (>= B*d*8 bytes)
    SVA_TRY(h2d(b, tmp, codes, sizeof(int64_t) * (size_t)B * d));
    hipLaunchKernelGGL(put_codes_kernel, dim3((B + 63) / 64), dim3(64), 0, b->stream, tmp, d, b->d_codes, b->T2, b->d_content_hist, b->hist_cap,
                       b->d_ncontent, B);
    b->h_ncontent += d;
    SVA_TRY(ar_delay_fill(b));
    b->delay_filled = true;
    SVA_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int sva_ar_decode_one(sva_batch* b, const int64_t* code, const float* noise, const int32_t* forced, int32_t* codes_out, int32_t* pos_out) {
    SVA_CHECK(b && code && codes_out && b->begun && b->delay_filled, "sva_ar_decode_one: bad argument / delay not filled");
    SVA_CHECK(b->p.chunk_frames == 1, "sva_ar_decode_one needs chunk_frames == 1");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    const sva_config& c = b->e->cfg;
    const int B = b->B, ncb = c.num_codebooks;
    hipStream_t st = b->stream;
    long long* tmp = (long long*)b->d_tok_raw;     // scratch: B*8 ints >= B int64
    SVA_TRY(h2d(b, tmp, code, sizeof(int64_t) * (size_t)B));
    hipLaunchKernelGGL(put_codes_kernel, dim3((B + 63) / 64), dim3(64), 0, st, tmp, 1, b->d_codes, b->T2, b->d_content_hist, b->hist_cap,
                       b->d_ncontent, B);
    b->h_ncontent += 1;
    b->noise_on_device = (noise == nullptr);
    if (noise) SVA_TRY(h2d(b, b->d_noise, noise, sizeof(float) * (size_t)B * (c.ar_vocab + ncb * c.codebook_size)));
    const int uf = forced ? 1 : 0;
    SVA_TRY(h2d(b, b->d_use_forced, &uf, sizeof(int)));
    b->h_use_forced = uf;
    if (forced) SVA_TRY(h2d(b, b->d_forced, forced, sizeof(int) * (size_t)B * ncb));
    SVA_TRY(ar_decode_frame(b, 0));
    for (int i = 0; i < B; ++i) { b->h_last_pos[i] += 2; b->h_nframes[i] += 1; }
    SVA_HIP(hipMemcpyAsync(codes_out, b->d_step_audio, sizeof(int) * (size_t)B * ncb, hipMemcpyDeviceToHost, st));
    if (pos_out) SVA_HIP(hipMemcpyAsync(pos_out, b->d_last_pos, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, st));
    SVA_HIP(hipStreamSynchronize(st));
    return 0;
}

// ---- offline ARVCWrapper.generate (modules/arvc_wrapper.py:82-98 + DualARWrapper.generate, dual_ar_stream.py:698-762) ----
__global__ void put_row_kernel(const float* __restrict__ table, long long code, int D, float* __restrict__ dst) {
    for (int i = threadIdx.x; i < D; i += blockDim.x) dst[i] = table[code * D + i];
}
__global__ void prepare_offline_step_kernel(const float* __restrict__ cached_audio_emb, const float* __restrict__ content_emb,
                                            const long long* __restrict__ rem, int step, const int* __restrict__ last_pos, int D,
                                            float* __restrict__ x, int* __restrict__ slot, int* __restrict__ pos) {
    const long long code = rem[step];
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        x[i] = cached_audio_emb[i];
        x[D + i] = content_emb[code * D + i];
    }
    if (threadIdx.x == 0) { slot[0] = 0; slot[1] = 0; pos[0] = last_pos[0] + 1; pos[1] = last_pos[0] + 2; }
}

extern "C" int sva_generate(sva_batch* b, const int64_t* ref_cc, const int32_t* ref_ac, int R, const int64_t* src_cc, int S,
                            const float* style, const float* timbre, uint64_t noise_seed, const float* noise, int32_t* codes_out) {
    SVA_CHECK(b && ref_cc && ref_ac && src_cc && style && timbre && codes_out, "sva_generate: null argument");
    SVA_CHECK(b->B == 1 && b->p.chunk_frames == 1, "sva_generate: the offline path is batch 1 / chunk 1 like the reference");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int D = c.ar_dim, d = b->p.delay, ncb = c.num_codebooks, nspk = c.timbre_tokens + 1;
    hipStream_t st = b->stream;
    SVA_CHECK(S > d && S <= b->hist_cap, "sva_generate: source length must exceed the delay");
    const int Rp = R + d, M = nspk + 2 * Rp + 1;
    SVA_CHECK(M + 2 * (S - 1) <= c.max_seq_len && M <= b->Mmax, "sva_generate: prompt + source exceed the 2048-position KV cache (the reference breaks there too)");
    SVA_TRY(h2d(b, b->d_style, style, sizeof(float) * c.style_dim));
    SVA_TRY(h2d(b, b->d_timbre, timbre, sizeof(float) * c.timbre_tokens * c.timbre_dim));
    unsigned long long sd = noise_seed;
    SVA_TRY(h2d(b, b->d_seed, &sd, sizeof(sd)));
    SVA_HIP(hipMemsetAsync(b->d_nframes, 0, sizeof(int), st));
    b->h_nframes[0] = 0;
    // prefill_cond = [ref_cond, src_cond[:d]] interleaved with [wait4start[:d], embed(ref_audio)]  (:711-715)
    std::vector<int64_t> cc(ref_cc, ref_cc + R);
    for (int i = 0; i < d; ++i) cc.push_back(src_cc[i]);
    std::vector<int32_t> ac((size_t)ncb * Rp, 0);
    for (int q = 0; q < ncb; ++q)
        for (int i = 0; i < R; ++i) ac[(size_t)q * Rp + i] = ref_ac[(size_t)q * R + i];
    SVA_TRY(stage_prompt(b, cc, ac, Rp));
    // remaining_cond = [src_cond[d:], wait4end[:d]]  (:716), wait4end_j addressed as vocab + j in the extended table
    std::vector<int64_t> rem(S);
    for (int i = 0; i < S; ++i) rem[i] = i < S - d ? src_cc[d + i] : (int64_t)c.ar_vocab + (i - (S - d));
    struct DevGuard { long long* p = nullptr; ~DevGuard() { if (p) (void)hipFree(p); } } remq_guard;      // freed on every return path
    SVA_HIP(hipMalloc((void**)&remq_guard.p, sizeof(long long) * (size_t)S));
    long long* d_remq = remq_guard.p;
    SVA_HIP(hipMemcpyAsync(d_remq, rem.data(), sizeof(long long) * (size_t)S, hipMemcpyHostToDevice, st));
    SVA_HIP(hipStreamSynchronize(st));
    // speaker prefix + prompt rows, then remaining_cond[0] as the last row
    SVA_TRY(gemm_call(b, b->d_timbre, (long)c.timbre_tokens * c.timbre_dim, 0, c.timbre_dim, 1, c.timbre_tokens, 1, 1, 1, c.timbre_dim, e->context_in,
                      b->spk, (long)nspk * D, 0, D));
    SVA_TRY(gemm_call(b, b->d_style, c.style_dim, 0, c.style_dim, 1, 1, 1, 1, 1, c.style_dim, e->style_in, b->spk + (long)c.timbre_tokens * D, D, 0, D));
    hipLaunchKernelGGL(build_prompt_kernel, dim3(M - 1), dim3(256), 0, st, b->spk, nspk, e->content_emb, e->codebook_emb, e->wait4start,
                       b->d_prompt_cc, b->d_prompt_ac, b->Pmax, Rp, d, ncb, c.codebook_size, D, b->ax);
    hipLaunchKernelGGL(put_row_kernel, dim3(1), dim3(256), 0, st, e->content_emb, (long long)rem[0], D, b->ax + (long)(M - 1) * D);
    std::vector<int> hs(M, 0), hp(M);
    for (int i = 0; i < M; ++i) hp[i] = i;
    SVA_HIP(hipMemcpyAsync(b->d_slot, hs.data(), sizeof(int) * M, hipMemcpyHostToDevice, st));
    SVA_HIP(hipMemcpyAsync(b->d_pos, hp.data(), sizeof(int) * M, hipMemcpyHostToDevice, st));
    SVA_HIP(hipStreamSynchronize(st));
    SVA_TRY(ar_layers_pass(b, e->ar_layers, M, b->d_slot, b->d_pos, e->rope_ar, (float*)b->kv_slow, b->kv_slow_layer, b->kv_slow_slot,
                           c.max_seq_len, b->ax, 0, 0));
    const int lp = M - 1;
    SVA_TRY(h2d(b, b->d_last_pos, &lp, sizeof(int)));
    b->h_last_pos[0] = lp;
    const int nstride = c.ar_vocab + ncb * c.codebook_size;
    const int zero = 0;
    SVA_TRY(h2d(b, b->d_use_forced, &zero, sizeof(int)));
    b->h_use_forced = 0;
    b->noise_on_device = (noise == nullptr);
    for (int i = 0; i < S; ++i) {
        if (noise) SVA_TRY(h2d(b, b->d_noise, noise + (size_t)i * nstride, sizeof(float) * nstride));
        if (i == 0) {
            // the prefill's decode ignores the caller's sampling_kwargs: generate() calls decode_one_token_ar without them
            // (dual_ar_stream.py:722), i.e. with temperature = top_p = 0.7
            const float t_user = b->p.temperature, p_user = b->p.top_p;
            b->p.temperature = 0.7f; b->p.top_p = 0.7f;
            b->edits_skip = true;                     // ... and without previous_tokens / suppress_tokens
            const int rc0 = ar_frame_tail(b, 0, 0, (long)(M - 1) * D, d_remq, S, 0, 0);
            b->edits_skip = false;
            b->p.temperature = t_user; b->p.top_p = p_user;
            SVA_TRY(rc0);
        } else if (b->use_mega && !b->edits_on) {
            // decode steps of the offline loop (dual_ar_stream.py:735-760) through the persistent kernel: the same two tokens
            // [embed(previous codes), cond_i] at the next two positions, one launch per frame instead of ~200
            SVA_TRY(ar_decode_frame_mega(b, 0, d_remq, i));
            b->h_last_pos[0] += 2;
        } else {
            hipLaunchKernelGGL(prepare_offline_step_kernel, dim3(1), dim3(256), 0, st, b->cached_audio_emb, e->content_emb, d_remq, i, b->d_last_pos, D,
                               b->ax, b->d_slot, b->d_pos);
            SVA_TRY(ar_layers_pass(b, e->ar_layers, 2, b->d_slot, b->d_pos, e->rope_ar, (float*)b->kv_slow, b->kv_slow_layer, b->kv_slow_slot,
                                   c.max_seq_len, b->ax));
            SVA_TRY(ar_frame_tail(b, 0, (long)2 * D, (long)D, d_remq, S, i, 2));
            b->h_last_pos[0] += 2;
        }
        b->h_nframes[0] += 1;
    }
    SVA_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < ncb; ++q)
        SVA_HIP(hipMemcpy(codes_out + (size_t)q * S, b->d_pred_hist + (size_t)q * b->hist_cap, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sva_encode_window(sva_batch* b, const float* audio, int64_t* codes_out, float* u_out) {
    SVA_CHECK(b && audio && codes_out, "null argument");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    hipStream_t st = b->stream;
    SVA_HIP(hipMemcpyAsync(b->ring, audio, sizeof(float) * (size_t)b->B * b->N, hipMemcpyHostToDevice, st));
    b->gemm_flops = 0; b->gemm_launches = 0; b->gemm_bytes = 0;
    SVA_HIP(hipEventRecord(b->ev[0], st));
    SVA_TRY(encode(b, nullptr, 0, 0));
    SVA_HIP(hipEventRecord(b->ev[1], st));
    SVA_HIP(hipMemcpyAsync(codes_out, b->d_codes, sizeof(int64_t) * (size_t)b->B * b->T2, hipMemcpyDeviceToHost, st));
    if (u_out) SVA_HIP(hipMemcpyAsync(u_out, b->d_u, sizeof(float) * (size_t)b->B * b->T2 * b->e->cfg.bsq_bits, hipMemcpyDeviceToHost, st));
    SVA_HIP(hipStreamSynchronize(st));
    float t;
    if (hipEventElapsedTime(&t, b->ev[0], b->ev[1]) == hipSuccess) b->last_ms[0] = t;
    return 0;
}

// wav2target_fn (evaluations/infer_arvc.py:168-171) -> FireflyArchitecture.encode (firefly.py:560-574) on full windows:
// log-mel -> voc.backbone -> quantizer.downsample -> grouped FSQ indices.  Same front-end kernels as the content encoder
// with the vocoder's weight set; the window buffers are shared, so this does not interleave with a running stream.
extern "C" int sva_firefly_encode(sva_batch* b, const float* audio, int32_t* codes_out) {
    SVA_CHECK(b && audio && codes_out, "null argument");
    sva_engine* e = b->e;
    SVA_CHECK(e->vocf.loaded, "firefly.encode weights (voc.backbone.*, voc.quantizer.downsample.*, ...project_in) were not loaded");
    SVA_HIP(hipSetDevice(e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    hipStream_t st = b->stream;
    const sva_config& c = e->cfg;
    SVA_HIP(hipMemcpyAsync(b->ring, audio, sizeof(float) * (size_t)b->B * b->N, hipMemcpyHostToDevice, st));
    b->gemm_flops = 0; b->gemm_launches = 0; b->gemm_bytes = 0;
    SVA_HIP(hipEventRecord(b->ev[0], st));
    SVA_TRY(enc_frontend_window(b, nullptr, 0, 0, b->T0, &e->vocf));
    SVA_TRY(launch_fsq_encode(b->d2.p, b->d2.bstride, (long)b->d2.H * c.voc_dim, c.voc_dim, b->B, b->T2, c.num_codebooks,
                              c.voc_dim / c.num_codebooks, e->fsq_in_W, e->fsq_in_b, b->d_fsq_codes, (long)c.num_codebooks * b->T2,
                              b->T2, st));
    SVA_HIP(hipEventRecord(b->ev[1], st));
    SVA_HIP(hipMemcpyAsync(codes_out, b->d_fsq_codes, sizeof(int32_t) * (size_t)b->B * c.num_codebooks * b->T2, hipMemcpyDeviceToHost, st));
    SVA_HIP(hipStreamSynchronize(st));
    float t;
    if (hipEventElapsedTime(&t, b->ev[0], b->ev[1]) == hipSuccess) b->last_ms[0] = t;
    return 0;
}

extern "C" long sva_get_tap(sva_batch* b, const char* what, void* out, long out_bytes) {
    if (!b || !what || !out) { set_error("null argument"); return -1; }
    hipSetDevice(b->e->device);
    if (quiesce(b)) return -2;            // pipelined mode: join the stage streams first
    const sva_config& c = b->e->cfg;
    const std::string w(what);
    const void* src = nullptr;
    long bytes = 0;
    const int B = b->B, chunk = b->p.chunk_frames;
    if (w == "content_codes") { src = b->d_step_content; bytes = sizeof(int) * (long)B * chunk; }
    else if (w == "audio_codes") { src = b->d_step_audio; bytes = sizeof(int) * (long)B * c.num_codebooks * chunk; }
    else if (w == "sampled_codes") { src = b->d_tok_raw; bytes = sizeof(int) * (long)B * c.num_codebooks; }
    else if (w == "slow_logits") { src = b->slow_logits; bytes = sizeof(float) * (long)B * c.ar_vocab; }
    else if (w == "fast_logits") { src = b->fast_logits; bytes = sizeof(float) * (long)B * c.num_codebooks * c.codebook_size; }
    else if (w == "hidden") { src = b->hidden; bytes = sizeof(float) * (long)B * c.ar_dim; }
    else if (w == "semantic") { src = b->d_sem; bytes = sizeof(int) * (long)B; }
    else if (w == "last_pos") { src = b->d_last_pos; bytes = sizeof(int) * (long)B; }
    else if (w == "window_codes") { src = b->d_codes; bytes = sizeof(long long) * (long)B * b->T2; }
    else if (w == "u") { src = b->d_u; bytes = sizeof(float) * (long)B * b->T2 * c.bsq_bits; }
    else if (w == "mel") { src = b->mel.p; bytes = sizeof(float) * (long)B * b->mel.bstride; }
    else if (w == "feat") { src = b->feat.p; bytes = sizeof(float) * (long)B * b->feat.bstride; }
    else if (w == "mag") { src = b->mag; bytes = sizeof(float) * (long)B * b->T0 * 1088; }
    else if (w == "z") { src = b->tr_z; bytes = sizeof(float) * (long)B * b->T2 * c.tr_dim; }
    else if (w == "ar_fail") { src = b->d_ar_fail; bytes = b->d_ar_fail ? (long)sizeof(int) : 0; if (!src) { set_error("ar_fail: the persistent decode kernel is not in use"); return -1; } }
    else if (w == "ar_timing") { src = b->d_ar_dbg; bytes = b->d_ar_dbg ? 1024L * (long)sizeof(long long) : 0; if (!src) { set_error("ar_timing: set SVA_AR_TIMING=1"); return -1; } }
    else if (w == "voc_z") { src = b->pin.p; bytes = sizeof(float) * (long)B * b->pin.bstride; }
    else { set_error("unknown tap " + w); return -1; }
    if (out_bytes < bytes) { set_error("tap buffer too small"); return -1; }
    if (hipStreamSynchronize(b->stream) != hipSuccess || hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("tap copy failed");
        return -2;
    }
    return bytes;
}

extern "C" int sva_get_timings(sva_batch* b, float ms[4]) {
    SVA_CHECK(b && ms, "null argument");
    for (int i = 0; i < 4; ++i) ms[i] = b->last_ms[i];
    return 0;
}
extern "C" int sva_profile_gemm(sva_batch* b, int enable) {
    SVA_CHECK(b, "null batch");
    b->prof_on = enable != 0;
    b->prof_n = 0;
    if (enable) b->prof_shapes.clear();
    // per-launch durations are only meaningful when launches do not overlap: single stream while profiling
    static bool saved = true;
    if (enable) { saved = b->concurrency; b->concurrency = false; }
    else b->concurrency = saved;
    return 0;
}
extern "C" int sva_get_gemm_profile(sva_batch* b, double* total_ms, long* launches) {
    SVA_CHECK(b && total_ms && launches, "null argument");
    SVA_HIP(hipSetDevice(b->e->device));
    (void)hipGetLastError();
    SVA_TRY(quiesce(b));
    SVA_HIP(hipStreamSynchronize(b->stream));
    double tot = 0;
    for (int i = 0; i + 1 < b->prof_n; i += 2) {
        float t = 0;
        SVA_HIP(hipEventElapsedTime(&t, b->prof_ev[i], b->prof_ev[i + 1]));
        tot += t;
    }
    *total_ms = tot;
    *launches = b->prof_n / 2;
    return 0;
}
// per-launch table of the profiled steps: rows of (M, N, K, taps, mode, microseconds) as doubles; returns #rows
extern "C" long sva_get_gemm_profile_table(sva_batch* b, double* out, long max_rows) {
    if (!b || !out) return -1;
    hipSetDevice(b->e->device);
    hipStreamSynchronize(b->stream);
    long n = std::min<long>(max_rows, (long)b->prof_shapes.size());
    n = std::min<long>(n, b->prof_n / 2);
    for (long i = 0; i < n; ++i) {
        float t = 0;
        hipEventElapsedTime(&t, b->prof_ev[2 * i], b->prof_ev[2 * i + 1]);
        const auto& sh = b->prof_shapes[i];
        out[i * 6 + 0] = sh[0]; out[i * 6 + 1] = sh[1]; out[i * 6 + 2] = sh[2]; out[i * 6 + 3] = sh[3]; out[i * 6 + 4] = sh[4];
        out[i * 6 + 5] = t * 1e3;
    }
    return n;
}
extern "C" int sva_stream_codes(sva_batch* b, int slot, int n, int32_t* codes_out, long* n_frames_total) {
    SVA_CHECK(b && slot >= 0 && slot < b->B, "bad slot");
    const int nf = b->h_nframes[slot], cap = b->hist_cap, ncb = b->e->cfg.num_codebooks;
    if (n_frames_total) *n_frames_total = nf;
    SVA_CHECK(n >= 0 && n <= nf && n <= cap, "more frames requested than the stream has decoded (or than the 4096-frame ring holds)");
    if (!n) return 0;
    SVA_CHECK(codes_out, "null argument");
    SVA_HIP(hipSetDevice(b->e->device));
    SVA_TRY(quiesce(b));
    SVA_HIP(hipStreamSynchronize(b->stream));
    std::vector<int> ring(cap);
    for (int q = 0; q < ncb; ++q) {
        SVA_HIP(hipMemcpy(ring.data(), b->d_pred_hist + ((long)slot * ncb + q) * cap, sizeof(int) * cap, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) codes_out[(size_t)q * n + i] = ring[(nf - n + i) & (cap - 1)];
    }
    return 0;
}
extern "C" int sva_get_gemm_bytes(sva_batch* b, double* bytes) {
    SVA_CHECK(b && bytes, "null argument");
    *bytes = b->gemm_bytes;
    return 0;
}
extern "C" int sva_get_gemm_stats(sva_batch* b, double* flops, long* launches) {
    SVA_CHECK(b && flops && launches, "null argument");
    *flops = b->gemm_flops;
    *launches = b->gemm_launches;
    return 0;
}
