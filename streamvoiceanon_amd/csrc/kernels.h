// Launchers of the non-GEMM kernels (kernels.hip).  All activations are fp32 channel-last.
#pragma once
#include "sva_common.h"

namespace sva {

// ---- content encoder -------------------------------------------------------------------
// E0/E1: causal STFT magnitude of the sliding audio window kept as a ring.
//   ring  [B, N] (N = window samples); oldest sample at ((*step + add) * n_chunk) % N
//   (step == nullptr -> 0);  frames m0 .. m0+nfr-1 of the window -> mag[b*mag_bstride + i*ldm], bins 0..1024
//   written, pad columns zeroed
int launch_stft_mag_ring(const float* ring, const int* step, int n_chunk, int add, int B, int N, const float2* twiddle,
                         const float* hann, float* mag, int ldm, long mag_bstride, int m0, int nfr, hipStream_t st);
//   two frame ranges in one launch: frames [m0, m0+nfr) -> rows [0, nfr), frames [m0b, m0b+nfrb) -> rows [row_b0, row_b0+nfrb)
int launch_stft_mag_ring2(const float* ring, const int* step, int n_chunk, int add, int B, int N, const float2* twiddle,
                          const float* hann, float* mag, int ldm, long mag_bstride, int m0, int nfr, int m0b, int nfrb, int row_b0, hipStream_t st);

// depthwise causal k=7 conv + LayerNorm(eps) over channels (ConvNeXtBlock prologue).
//   x element (b, r, c): x[b*x_bstride + x_off + r*C + c]; output row t reads rows t..t+6.
//   wT [7][C] (tap-major), out element (b, t, c) at out[b*o_bstride + t*C + c].
//   outp != null: the consumer is a planes GEMM (gemm_planes.hip) -- fp16 hi (+ lo when op_planes == 2) parts instead of the fp32 rows, plane
//   p at outp + p * op_pstride, element (b, t, c) at the fp32 tensor's index (op_rows == 0) or K-blocked over op_rows = B * T rows (planes_split.h)
int launch_dwconv7_ln(const float* x, long x_bstride, long x_off, int B, int T, int C, const float* wT,
                      const float* bias, const float* ln_w, const float* ln_b, float eps, float* out, long o_bstride,
                      hipStream_t st, unsigned short* outp = nullptr, long op_pstride = 0, int op_planes = 0, long op_rows = 0);

// row LayerNorm / RMSNorm with strided in/out (rows = B*T).
int launch_layernorm_rows(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int C,
                          const float* w, const float* b, float eps, float* out, long o_bstride, long o_off,
                          int ldo, hipStream_t st, int skip_lo = 0, int skip_hi = 0);      // rows t in [skip_lo, skip_hi) of each item are left alone
int launch_rmsnorm_rows(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int C, const float* w,
                        float eps, float* out, long o_bstride, long o_off, int ldo, hipStream_t st,
                        unsigned short* outp = nullptr, long op_pstride = 0, int op_planes = 0, long op_rows = 0);     // planes output as launch_dwconv7_ln

// causal self-attention of the BSQ pre-transformer: qkv [B, T, 3*D] -> out [B, T, D]; RoPE
// (adjacent pairs, bf16-rounded table rope[T][hd/2][2]) applied to q and k on load.
//   outp != null (T <= 128 only, enc_attention_can_write_planes): planes output as launch_dwconv7_ln
bool enc_attention_can_write_planes(int T);
int launch_enc_attention(const float* qkv, const float* rope, int B, int T, int H, int hd, float* out, int row0,
                         hipStream_t st, unsigned short* outp = nullptr, long op_pstride = 0, int op_planes = 0, long op_rows = 0);

// BSQ: u = W z + b (nbits x C), index = sum_d (u_d > 0) << (nbits-1-d); optional L2-normalised u out.
int launch_bsq(const float* z, long z_bstride, long z_off, int ldz, int B, int T, int C, const float* norm_w /*fused RMSNorm or null*/,
               float eps, float* zn_out /*normalised rows, same layout as z, or null*/, const float* W, const float* bias, int nbits,
               long long* idx_out, int idx_bstride, int idx_off, float* u_out, hipStream_t st);

// ---- dual AR ------------------------------------------------------------------------------
// RoPE on q,k of qkv rows [M, 3*D] (in place) + KV-cache write at (slot[m], pos[m]).
//   cache layout per layer: [slot][2][H][S][hd]; KV = float or __half
template <typename KV>
int launch_rope_kvwrite(float* qkv, int M, int H, int hd, const int* slot, const int* pos, const float* rope,
                        KV* cache, long slot_stride, int S, hipStream_t st);
// RoPE + KV write + attention over <= 8 cached positions in one launch (fast AR, batched decode; fp32 cache)
int launch_ar_fast_attention(const float* qkv, int M, int H, const int* slot, const int* pos, const float* rope, float* cache,
                             long slot_stride, int S, float* out, hipStream_t st);
// attention of M query rows against their slot's cache, keys 0..pos[m] inclusive.
template <typename KV>
int launch_ar_attention(const float* qkv, int M, int H, int hd, const int* slot, const int* pos, const KV* cache,
                        long slot_stride, int S, float* out, hipStream_t st, float* partial = nullptr, int splits = 1);
// rows (2 s, 2 s + 1) = positions (p, p + 1) of one slot for every s: one workgroup per (head, stream) reads the slot's K / V rows once for both
template <typename KV>
int launch_ar_attention_pairs(const float* qkv, int M, int H, int hd, const int* slot, const int* pos, const KV* cache, long slot_stride, int S, float* out,
                              hipStream_t st);
// The same attention for M rows at CONSECUTIVE positions pos0 .. pos0 + M - 1 of ONE slot (prefill / re-prefill / offline generate):
// flash-style MFMA kernel, one 16-row query tile per workgroup, its four waves splitting the key blocks.
template <typename KV>
int launch_ar_prefill_attention(const float* qkv, int M, int H, int hd, int slot0, int pos0, const KV* cache, long slot_stride, int S, float* out,
                                hipStream_t st);
// partial != null: split-key decode -- `splits` workgroups per (head, row) write unnormalised partials
// partial[((m*H + h)*splits + s)*68 + {0..63: P.V, 64: max score, 65: exp-sum}] that gemv mode 4 merges

// nucleus + temperature + Exp(1)-argmax sampler (modules/dual_ar_stream.py:1092-1132), one
// workgroup per row.  noise: [rows, ldn] or nullptr -> on-device counter RNG keyed by
// (seed[row], frame[row], kind) at element offset noise_elem_off.
int launch_sampler(const float* logits, int rows, int V, int ldl, const float* noise, int ldn,
                   const unsigned long long* seed, const int* frame, int kind, int noise_elem_off,
                   float temperature, float top_p, int* tok_out, int tok_stride, hipStream_t st);

// logits_to_probs' edits ahead of the nucleus cut (modules/dual_ar_stream.py:1107-1117), in place on rows [rows][ldl]:
//   repetition penalty on the tokens of prev[0 .. W) (negative entries skipped): l < 0 ? l * penalty : l / penalty, every listed token
//   once (the reference gathers the ORIGINAL scores and scatters them back, so duplicates do not compound);
//   then logits[suppress[i]] = -inf.  params = {W, n_suppress, penalty bits} in device memory (stable kernel arguments under graphs).
int launch_logit_edits(float* logits, int rows, int V, int ldl, const int* prev, int prev_cap, const int* suppress, const int* params, hipStream_t st);

// out[r, :] = table[idx[r*idx_stride] + idx_offset, :]
int launch_gather_rows(const float* table, const int* idx, int idx_stride, int idx_offset, int rows, int D,
                       float* out, int ldo, hipStream_t st);
// audio embedding: out[r, :] = sum_i table[codes[r*code_stride + i*cb_stride] + i*codebook_size, :]
int launch_audio_embed(const float* table, const int* codes, int code_stride, int cb_stride, int rows, int ncb,
                       int codebook_size, int D, float* out, int ldo, hipStream_t st);

// ---- vocoder ------------------------------------------------------------------------------
// FSQ index -> latent: codes element (b,g,t) at codes[b*c_bstride + g*c_gstride + t] ->
// out (b, t, g*gdim + j) at out[b*o_bstride + o_off + t*ldo + g*gdim + j]
int launch_fsq_decode(const int* codes, long c_bstride, long c_gstride, int B, int T, int G, int gdim,
                      const float* Wout /*[G][gdim][4]*/, const float* bout /*[G][gdim]*/, float* out,
                      long o_bstride, long o_off, int ldo, hipStream_t st);
// latent -> FSQ index (firefly.encode): x (b, t, g*gdim + j) at x[b*x_bstride + x_off + t*ldx + ...] -> codes (b, g, t)
int launch_fsq_encode(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int G, int gdim,
                      const float* Win /*[G][4][gdim]*/, const float* bin /*[G][4]*/, int* codes, long c_bstride,
                      long c_gstride, hipStream_t st);
// conv_post (C -> 1, k taps) on silu(x) + tanh: x (b, r, c) at x[b*x_bstride + x_off + r*C + c]; row t reads t..t+k-1
int launch_conv_post_tanh(const float* x, long x_bstride, long x_off, int B, int T, int C, int k,
                          const float* w /*[k][C]*/, const float* bias, float* pcm, long p_bstride, long p_off,
                          hipStream_t st);

struct ShiftDesc {      // one history buffer: rows [T, T+H) move to [0, H) after a step
    float* ptr;
    long bstride;
    int H, T, C;
    int pad;
};
int launch_shift_history(const ShiftDesc* descs_dev, int n_desc, int B, hipStream_t st, int col_slices = 1, int* counter = nullptr, int counter_add = 0);

// small helpers
int launch_fill_i32(int* p, int n, int v, hipStream_t st);
int launch_add_i32(int* p, int v, hipStream_t st);
int launch_ring_write(float* ring, int* step, int B, int N, const float* chunk, int n, hipStream_t st);

}  // namespace sva

namespace sva {
// ---- decode GEMV (M <= 4 rows): one wave per pair of output columns, W rows streamed fully coalesced, the x rows
// held in registers; optional fused RMSNorm prologue and SwiGLU / RoPE+KV-write epilogues (kernels.hip) ----
struct Gemv {
    const float* X = nullptr; int ldx = 0; int M = 0;
    const float* W = nullptr; int N = 0, K = 0;
    const float* norm_w = nullptr; float eps = 1e-5f;     // RMSNorm(x) * norm_w fused on load
    const float* bias = nullptr;
    const float* res = nullptr; int ldr = 0;
    float* Y = nullptr; int ldy = 0;
    int mode = 0;                                         // 0 plain, 1 SwiGLU (w13 interleave), 2 q/k RoPE + KV write,
                                                          // 3 X = decode attention of the qkv rows over <= 8 cached keys (fast AR)
                                                          // 4 X = merge of split-key attention partials, S = #splits (slow AR)
    const int* slot = nullptr; const int* pos = nullptr; const float* rope = nullptr;
    float* kv = nullptr; long kv_slot_stride = 0; int S = 0, H = 0;
};
int launch_gemv(const Gemv& g, hipStream_t st);

// sampler for vocabularies <= 1024 without a sort (rank by counting), fused with teacher forcing and the embedding
// gather of the sampled token:  tok_raw[row*tok_stride] = sample;  tok[...] = forced ? forced : sample;
// emb_out[row, :] = emb_table[tok, :]  (emb_table may be null)
int launch_sampler_variant(int variant, const float* logits, int rows, int V, int ldl, const float* noise, int ldn,
                           const unsigned long long* seed, const int* frame, float temperature, float top_p, int* tok_out, hipStream_t st);
int launch_sampler_small(const float* logits, int rows, int V, int ldl, const float* noise, int ldn,
                         const unsigned long long* seed, const int* frame, int kind, int noise_elem_off, float temperature,
                         float top_p, int* tok_raw, int* tok, int tok_stride, const int* forced, int forced_stride,
                         const int* use_forced, const float* emb_table, int D, float* emb_out, int ldo, hipStream_t st);
}  // namespace sva
