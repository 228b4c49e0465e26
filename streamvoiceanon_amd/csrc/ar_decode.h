// Persistent small-batch decode kernel of the dual AR (ar_decode.hip): one launch = one frame (of every stream of the batch) of
// decode_one_token_ar (modules/dual_ar_stream.py:1168-1219) -- 12 slow layers on the two new tokens, the semantic head,
// 8 x (4 fast layers + codebook head + nucleus sampler) and the frame bookkeeping.
#pragma once
#include <hip/hip_runtime.h>

namespace sva {

constexpr int AR_WGS = 96;            // workgroups of the persistent kernel (= CUs of the AR stream's partition)
constexpr int AR_WAVES = AR_WGS * 4;
constexpr int AR_SLOW_LAYERS = 12, AR_FAST_LAYERS = 4;
constexpr int AR_PERSISTENT_MAX_STREAMS = 6;      // batches up to this size (fp16 AR: up to 4, engine.hip) decode with the persistent kernel (two streams per launch); above it
                                                  // the batched MFMA chain is faster (profiles/r02_streams_curve.txt)
// weights of one layer in the decode kernel's layout: row-major [N][K] (fp32 or fp16 by the kernel's template argument);
// w13: wave w of the kernel owns rows [12w, 12w + 12) = w1 rows 6w..6w+5 followed by w3 rows 6w..6w+5
struct ArLayerW {
    const void *wqkv, *wo, *w13, *w2;
    const float *attn_norm, *ffn_norm;
};

struct ArDecodeArgs {
    ArLayerW slow[AR_SLOW_LAYERS];
    ArLayerW fast[AR_FAST_LAYERS];
    const void* out_w;            // [vocab][768] semantic head
    const float* out_norm;
    const void* fast_out_w;       // [codebook_size][768]
    const float* fast_norm;
    const float *content_emb, *codebook_emb, *fast_emb, *rope_slow, *rope_fast;
    // stream state (slot 0 of a one-stream batch)
    const long long* codes;       // content code of this frame = codes[code_off]
    int code_off;
    float* cached_audio_emb;      // [768] in: embedding of the previous frame's codes; out: of this frame's
    int* last_pos;                // [1] last written slow KV position (advanced by 2)
    int* nframes;                 // [1] decoded frames (RNG counter, history index; advanced by 1)
    const unsigned long long* seed;
    void* kv_slow;                // [layer][2][H][S][64] (slot 0), float or __half
    long kv_layer_stride;         // elements
    int S;
    float* kv_fast;               // [4][8][2][768] scratch, rewritten every frame
    unsigned long long *gx, *gbig, *gatt, *glog, *ga;     // granule buffers: [2][768], [2][2304], [96][66], [1024], [2][768]
    unsigned* epoch;              // [1] running phase counter (tags), persists across launches
    int* fail;                    // [1] set to a phase code if a gather timed out (never in a healthy run)
    int* fail_host;               // null, or a host-mapped mirror of *fail: written at the end of a launch that saw the flag set (callers that never synchronise)
    long long* dbg;               // null, or [1024] phase timestamps of workgroup 0 (SVA_AR_TIMING=1)
    // taps / outputs
    float *slow_logits, *fast_logits, *hidden;
    int *sem, *tok_raw, *tok, *step_audio, *pred_hist, *step_content;
    int hist_cap, chunk, ci;
    const float* noise;           // [vocab + 8 * codebook_size] Exp(1) draws of this frame, or null = counter RNG
    const int* forced;            // [8][chunk] teacher-forced codes (used when *use_forced)
    const int* use_forced;
    float inv_temp, top_p;
    int skip_semantic;
    int vocab, codebook_size;
    // several streams in one launch: gridDim.y = streams, workgroup row y works on slot y = every per-stream pointer above advanced by
    // y times its stride (elements of the pointer's type).  Each slot has its own granule buffers and epoch: the 96 workgroups of a
    // slot only ever talk to each other.
    int slot_base;                // first slot of this launch (workgroup row y works on slot slot_base + y)
    struct SlotStride {
        long codes, emb, kv_slot, kv_fast, gran, slow_logits, fast_logits, hidden, tok, step_audio, pred_hist, step_content, noise, forced;
    } ss;
};

// wt_half / kv_half: element types of the weights / the slow KV cache (0 = fp32, 1 = fp16)
// one_per_cu: pad the LDS request so that no two of the 96 workgroups share a CU
// n_slots: streams decoded by this launch (grid 96 x n_slots; all 96 n_slots workgroups must be co-resident)
int launch_ar_decode(const ArDecodeArgs& a, int wt_half, int kv_half, bool one_per_cu, hipStream_t st, int n_slots = 1);
// blocks of the kernel the runtime says fit one CU at the launch's LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor)
int ar_decode_occupancy(int wt_half, int kv_half, int* blocks_per_cu);
size_t ar_decode_granule_words();     // u64 words of a stream's granule block (gx | gbig | gatt | glog | ga, in this order)

}  // namespace sva
