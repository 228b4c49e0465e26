// Persistent BATCHED decode kernel of the dual AR (ar_batch.hip): one launch = one frame of decode_one_token_ar
// (modules/dual_ar_stream.py:1168-1219) for EVERY stream of a batch -- 12 slow layers on the 2 B new tokens, the semantic head,
// 8 x (4 fast layers on B rows + codebook head + nucleus sampler) and the frame bookkeeping -- instead of the ~265 dependent
// launches of the multi-launch chain (engine.hip: ar_layers_pass + ar_frame_tail).
#pragma once
#include "ar_decode.h"

namespace sva {

constexpr int AR_BATCH_MAX_STREAMS = 128;

struct ArBatchArgs {
    // weights: wqkv / wo / w2 row-major [N][K]; w13 = (gate tile | up tile) interleaved in 16-row groups (Packer::w13); fp32 or fp16 by
    // the kernel's template argument
    ArLayerW slow[AR_SLOW_LAYERS];
    ArLayerW fast[AR_FAST_LAYERS];
    const void* out_w;            // [vocab][768] semantic head
    const float* out_norm;
    const void* fast_out_w;       // [codebook_size][768]
    const float* fast_norm;
    const float *codebook_emb, *fast_emb, *rope_slow, *rope_fast;
    int B;                        // streams (slots 0 .. B - 1)
    int G;                        // workgroups of the launch (all co-resident)
    // stream state (slot-indexed)
    float* cached_audio_emb;      // [B][768] out: embedding of this frame's codes
    int* last_pos;                // [B] last written slow KV position (advanced by 2)
    int* nframes;                 // [B] decoded frames (RNG counter, history index; advanced by 1)
    const unsigned long long* seed;
    void* kv_slow;                // [layer][slot][K|V][H][S][64], float or __half
    long kv_layer_stride, kv_slot_stride;     // elements
    int S;
    const float* xs_in;           // [2B][768] fp32: the frame's 2 B input tokens (ar_prepare_step_kernel)
    // activations that cross workgroups inside the launch: 8-byte {tag = epoch of the producing phase, fp32 value} granules -- see
    // ar_batch.hip.  Offsets of the arrays in one allocation: ar_batch_granule_words (same order)
    unsigned long long *gxs, *gqkv, *gatt, *gg;       // slow: [2B][768], [2B][2304], [2B][768], [2B][2304]
    unsigned long long *gxf, *gqkvf, *gattf, *ggf;    // fast: [B][768], [B][2304], [B][768], [B][2304]
    unsigned long long* gkvf;                         // [4][B][8][k 768 | v 768] fast-AR K / V of the frame's codebook positions
    unsigned long long *glog, *gsem;                  // [B][1024] codebook logits of the current codebook, [B][8192] semantic logits
    unsigned* epoch;              // [1] running phase counter, persists across launches
    unsigned* done;               // [1] exit counter of the launch (the last workgroup out advances *epoch)
    int* fail;                    // [1] set to a phase code if a wait timed out (never in a healthy run)
    int* fail_host;               // null, or a host-mapped mirror of *fail: written at the end of a launch that saw the flag set (callers that never synchronise)
    long long* dbg;               // null, or [1024] phase timestamps of workgroup 0 (SVA_DEBUG=ar_timing=1)
    // taps / outputs
    float *slow_logits, *fast_logits, *hidden;
    int *sem, *tok_raw, *tok, *step_audio, *pred_hist;
    int hist_cap, chunk, ci;
    const float* noise;           // slot s: noise + s * noise_ld: [vocab + 8 * codebook_size] Exp(1) draws of this frame; null = counter RNG
    long noise_ld;
    const int* forced;            // [B][8][chunk] teacher-forced codes (used when *use_forced)
    const int* use_forced;
    float inv_temp, top_p;
    int skip_semantic;
    int vocab, codebook_size;
};

// row-tile heights (x 16 rows) of the slow (2 B rows) and fast (B rows) linear phases for a batch size
void ar_batch_tiles(int B, int* mts, int* mtf);
// granules (8 bytes each) of the hand-off block for B streams and the offsets of its arrays (in the order of ArBatchArgs::g*)
constexpr int AR_BATCH_NBUF = 11;
size_t ar_batch_granule_words(int B, size_t offs[AR_BATCH_NBUF]);
// workgroups the kernel wants for B streams (one 16-column tile of the widest phase each)
int ar_batch_wanted_workgroups(int B);
int ar_batch_occupancy(int wt_half, int B, int* blocks_per_cu);
// wt_half: weights and slow KV cache in fp16 (ar_dtype = 1) or fp32
int launch_ar_batch(const ArBatchArgs& a, int wt_half, hipStream_t st);

}  // namespace sva
