/*
 * sva.h -- C ABI of the MI355X-native StreamVoiceAnon streaming voice-conversion engine.
 *
 * The reference (Plachtaa/StreamVoiceAnon) is pure Python and has no FFI of its own; this
 * header is the boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to replace
 * the four per-chunk seams of evaluations/infer_arvc.py `process_one_chunk` (:492-596):
 *     speech_tokenizer.encode  (:506-508)   -> sva_encode_window / inside sva_step
 *     model.decode_one         (:535-537)   -> inside sva_step (sva_prefill_prompt for :484-489)
 *     firefly.quantizer.decode (:175)       -> sva_vocode_window / inside sva_step
 *     firefly.head             (:175)       -> sva_vocode_window / inside sva_step
 *
 * Conventions: every function returns 0 on success and a negative code on failure (message via
 * sva_last_error()); nothing throws across the ABI.  The caller owns all I/O buffers; the engine
 * owns weights and per-stream state.  Handles are not re-entrant (one host thread per handle).
 * All host pointers are plain C arrays; no torch / C++ types appear in any signature.
 */
#ifndef SVA_H
#define SVA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sva_engine sva_engine;
typedef struct sva_batch sva_batch;

/* Model dimensions.  sva_config_default() fills the values of the reference's Hydra YAMLs
 * (configs/hydra_arcs/{vc/firefly_arvc_bsq_8192_delay0_8, speech_tokenizers/causal-encoder-lfq-8192,
 * vocoders/firefly_gan_vq}.yaml). */
typedef struct sva_config {
    int n_mels;            /* 160 */
    int enc_depths[4];     /* 3,3,9,3 */
    int enc_dims[4];       /* 128,256,384,512 */
    int tr_layers;         /* 8   BSQ pre_module transformer */
    int tr_heads;          /* 8 */
    int tr_dim;            /* 512 */
    int tr_inter;          /* 1536 */
    int bsq_bits;          /* 13 */
    int ar_dim;            /* 768 */
    int ar_heads;          /* 12 */
    int ar_layers;         /* 12 */
    int ar_fast_layers;    /* 4 */
    int ar_inter;          /* 2304 */
    int ar_vocab;          /* 8192 */
    int codebook_size;     /* 1000 */
    int num_codebooks;     /* 8 */
    int max_delay;         /* 8 */
    int max_seq_len;       /* 2048 */
    int timbre_dim;        /* 128 */
    int timbre_tokens;     /* 32 */
    int style_dim;         /* 192 */
    int voc_dim;           /* 512 */
    int ar_dtype;          /* 0: fp32 AR weights + fp32 KV (parity mode); 1: fp16 AR weights (streamed as fp16 by the batch-1 decode kernel;
                            * the batched / prefill GEMMs use the same fp16-rounded values) + fp16 slow KV cache, as the reference
                            * decodes under torch.autocast(fp16) with fp16 caches (evaluations/infer_arvc.py:55-59, 483, 493) */
    int mm_mode;           /* fp32-grade batch-scale GEMMs (batches of 10+ streams) of the encoder / vocoder: 1 (default) = two pre-split fp16 planes per
                            * operand, three part products (csrc/gemm_planes.hip): half the matrix work of 0, for operands inside the fp16
                            * range -- which torch.autocast(fp16), infer_arvc.py:493, demands of the reference too; 0 = three bf16 parts split
                            * inside the K loop, six products (csrc/gemm_split.hip, the round-3 kernel): any fp32 range.  (2 -- three pre-split
                            * bf16 planes -- was an A/B arm of round 4, slower than 0, and is refused since round 5.)  NOTE: mode 1, the default,
                            * has fp16's RANGE: an operand beyond +-65504 raises an error at the next sva_step* / sva_sync naming mm_mode = 0 */
    int voc_dtype;         /* 0: vocoder (firefly.decode) GEMMs in the mm_mode grade; 1: fp16 operands, fp32 accumulate -- the reference's
                            * own precision for this stage (torch.autocast(fp16) around code2wav_fn, infer_arvc.py:493, 571-590) */
} sva_config;

/* evaluations/infer_arvc.py setup_stream_caches (:443-460) + stream_infer defaults (:598-613) */
typedef struct sva_stream_params {
    int n_streams;             /* B concurrent streams (the reference is hard-wired to 1) */
    int encode_window_frames;  /* 128 */
    int decode_window_frames;  /* 64  (used only to size vocoder priming = window - 1 frames) */
    int chunk_frames;          /* decode_chunk_frames, 1 */
    int delay;                 /* 2 */
    int max_seq_frames;        /* 768 */
    int buffer_frames;         /* 32 */
    int max_prompt_frames;     /* 256 */
    float temperature;         /* 0.7  (modules/dual_ar_stream.py:1103) */
    float top_p;               /* 0.7  (:1104) */
    int voc_max_frames;        /* largest T accepted by sva_vocode_window / one streaming call (>= chunk) */
    int use_graph;             /* capture the steady-state step in a hipGraph */
    int skip_semantic;         /* skip the semantic-token head whose sample every caller discards (:833) */
    int pipeline;              /* sva_step_device only: run encoder / AR / vocoder of consecutive chunk-steps on three streams
                                * (E(n+1) || A(n) || V(n-1)); same results, higher throughput for simulated streaming; a caller that
                                * synchronises per chunk sees the unpipelined latency */
} sva_stream_params;

const char* sva_last_error(void);
int sva_config_default(sva_config* cfg);
int sva_stream_params_default(sva_stream_params* p);

/* ---- engine: weights ------------------------------------------------------------------ */
int sva_engine_create(const sva_config* cfg, int device, sva_engine** out);
/* name = reference state-dict key prefixed by its network: "arvc." (ARVCWrapper), "tok." (speech
 * tokenizer), "voc." (Firefly vocoder); data = fp32 host array of the given shape.  Unknown names
 * are ignored (strict=False, evaluations/infer_arvc.py:79-93,162).  Weight-norm pairs
 * (...parametrizations.weight.original0/1) are folded as g*v/||v|| (firefly.py:295-301). */
int sva_engine_load_weight(sva_engine* e, const char* name, int ndim, const int64_t* shape, const float* data);
/* pack to the compute layout and upload; after this the engine is immutable */
int sva_engine_finalize(sva_engine* e);
void sva_engine_destroy(sva_engine* e);

/* ---- batch of streams ------------------------------------------------------------------- */
int sva_batch_create(sva_engine* e, const sva_stream_params* p, sva_batch** out);
void sva_batch_destroy(sva_batch* b);

/* ARVCWrapper.prefill_prompt (modules/arvc_wrapper.py:100-112) + InferenceWrapper.prefill_prompt
 * bookkeeping (infer_arvc.py:463-489) for one stream slot: content codes int64[R], audio codes
 * int32[8][R] (row-major), style float[192], timbre float[32][128].  noise_seed keys the
 * on-device sampler noise of this utterance.  Call for every slot, then sva_streams_begin(). */
int sva_prefill_prompt(sva_batch* b, int slot, const int64_t* ref_content_codes, const int32_t* ref_audio_codes,
                       int R, const float* style, const float* timbre, uint64_t noise_seed);
/* zero the audio windows / histories and prime the streaming vocoder with the prompt tail */
int sva_streams_begin(sva_batch* b);

/* InferenceWrapper.process_one_chunk (infer_arvc.py:492-596) for all streams in lock step.
 *   pcm_in  host float[B][2048*chunk]      pcm_out host float[B][2048*chunk]
 *   noise   host float[B][chunk][vocab + 8*codebook_size] Exp(1) draws in the reference's RNG
 *           order (slow head first, then the 8 codebooks), or NULL = on-device counter RNG
 *   forced_codes host int32[B][8][chunk] teacher-forces the AR (parity tests), or NULL */
int sva_step(sva_batch* b, const float* pcm_in, float* pcm_out, const float* noise, const int32_t* forced_codes);
/* same with device pointers (no host copies); asynchronous: d_pcm_in must stay valid and d_pcm_out must not be read until
 * sva_sync() (or a later synchronous call on the handle) returns.  The engine runs on streams of its own and does NOT synchronise
 * with the caller's: whatever produced d_pcm_in (a kernel or copy on another stream) must have COMPLETED before this call.  With sva_stream_params.pipeline the stages of consecutive
 * calls overlap on three streams. */
int sva_step_device(sva_batch* b, const float* d_pcm_in, float* d_pcm_out);
/* Stream-ordered form -- what a torch caller wants: process_one_chunk in the reference runs on the caller's current stream
 * (evaluations/infer_arvc.py:495-508), so the producer of the input is ordered before it and the consumer of the output after it.
 * caller_stream is a hipStream_t (NULL = the legacy default stream; pass torch.cuda.current_stream().cuda_stream).  The engine's first
 * access to d_pcm_in waits (event, on the device) for everything enqueued on caller_stream so far; with join_output != 0 the
 * caller's stream then waits for the step's output, i.e. full same-stream semantics (and no overlap between consecutive steps);
 * with join_output == 0 consecutive steps keep overlapping, d_pcm_in must not be rewritten and d_pcm_out not read before a later
 * sva_join_stream(b, stream) (device-side wait for all engine work enqueued so far) or sva_sync(). */
int sva_step_device_on(sva_batch* b, const float* d_pcm_in, float* d_pcm_out, void* caller_stream, int join_output);
int sva_join_stream(sva_batch* b, void* caller_stream);
int sva_sync(sva_batch* b);
/* stream_infer's chunk loop (evaluations/infer_arvc.py:650-675) in one call: n_chunks consecutive sva_step_device steps over
 * host arrays pcm_in / pcm_out float[B][n_chunks * 2048 * chunk] (staged through device memory, stages pipelined when
 * sva_stream_params.pipeline is set), synchronised on return.  Same results as n_chunks calls of sva_step with noise = NULL. */
int sva_stream_chunks(sva_batch* b, const float* pcm_in, float* pcm_out, int n_chunks);

/* ---- seam-level entry points (parity tests, drop-in for the module calls) ------------------ */
/* speech_tokenizer.encode (firefly_encoder.py:553-566) on full windows: audio host float[B][W*2048]
 * -> codes int64[B][W] (+ optional L2-normalised pre-sign u float[B][W][13]) */
int sva_encode_window(sva_batch* b, const float* audio, int64_t* codes_out, float* u_out);
/* wav2target_fn (infer_arvc.py:168-171) -> FireflyArchitecture.encode (modules/vqgan/modules/firefly.py:560-574) of the
 * PROMPT on full windows: audio host float[B][W*2048] -> acoustic codes int32[B][8][W] (the ref_audio_codes operand of
 * sva_prefill_prompt).  Needs the optional voc.backbone.* / voc.quantizer.downsample.* / ...residual_fsq...project_in
 * tensors; fails with an error if the engine was finalized without them. */
int sva_firefly_encode(sva_batch* b, const float* audio, int32_t* codes_out);
/* code2wav_fn (infer_arvc.py:173-176) with the reference's window semantics (zero history):
 * codes host int32[B][8][T] -> pcm float[B][2048*T];  T <= voc_max_frames */
int sva_vocode_window(sva_batch* b, const int32_t* codes, int T, float* pcm_out);
/* the two halves of code2wav_fn as seams of their own (infer_arvc.py:175): firefly.quantizer.decode
 * (modules/vqgan/modules/fsq.py:112-116) codes int32[B][8][T] -> z float[B][4T][512] (channel-last; the reference tensor is
 * [B, 512, 4T]) and firefly.head (firefly.py:280-293) z -> pcm float[B][2048*T]; window semantics, T <= voc_max_frames */
int sva_quantizer_decode(sva_batch* b, const int32_t* codes, int T, float* z_out);
int sva_vocoder_head(sva_batch* b, const float* z, int T, float* pcm_out);
/* streaming-exact vocoder: continue from the ring-buffer state: codes int32[B][8][T] -> pcm[B][2048*T] */
int sva_vocode_stream(sva_batch* b, const int32_t* codes, int T, float* pcm_out);
int sva_vocode_reset(sva_batch* b);

/* AR seams with caller-supplied content codes (chunk_frames == 1):
 *   ARVCWrapper.prefill_src_condition4delay (modules/arvc_wrapper.py:114-119): codes int64[B][delay]
 *   ARVCWrapper.decode_one (:121-126 -> dual_ar_stream.py:817-837): code int64[B] -> codes_out int32[B][8],
 *   pos_out int32[B] (kv_pos[-1]); noise float[B][vocab + 8*codebook_size] or NULL; forced int32[B][8] or NULL */
int sva_ar_delay_fill(sva_batch* b, const int64_t* codes);
int sva_ar_decode_one(sva_batch* b, const int64_t* code, const float* noise, const int32_t* forced, int32_t* codes_out,
                      int32_t* pos_out);

/* Offline conversion, ARVCWrapper.generate (modules/arvc_wrapper.py:82-98 -> DualARWrapper.generate, dual_ar_stream.py:698-762)
 * on a batch created with n_streams = 1, chunk_frames = 1: one prefill of 33 + 2(R+delay) + 1 tokens, then S-1 decode steps
 * (the last `delay` frames are driven by the wait4end embeddings).  ref_cc int64[R], ref_ac int32[8][R], src_cc int64[S],
 * noise float[S][vocab + 8*codebook_size] or NULL (device RNG keyed by noise_seed); codes_out int32[8][S].
 * Follow with sva_vocode_window(codes, S) for code2wav_fn (evaluations/infer_arvc.py:349-360). */
int sva_generate(sva_batch* b, const int64_t* ref_cc, const int32_t* ref_ac, int R, const int64_t* src_cc, int S,
                 const float* style, const float* timbre, uint64_t noise_seed, const float* noise, int32_t* codes_out);

/* taps of the last step: "content_codes" int32[B][chunk], "audio_codes" int32[B][8][chunk] (as int32),
 * "slow_logits" float[B][vocab], "fast_logits" float[B][8][codebook_size], "hidden" float[B][dim],
 * "semantic" int32[B], "last_pos" int32[B], "mel" float[B][T][160] ... ; returns #bytes or <0 */
long sva_get_tap(sva_batch* b, const char* what, void* out, long out_bytes);
/* the stream state `pred_codes[..., -n:]` of evaluations/infer_arvc.py:520-523 for one slot: the last n predicted frames
 * (n <= frames decoded so far, n <= 4096 = the device ring), int32 codes_out[8][n].  Drains the batch's streams first.
 * *n_frames_total (optional) receives the number of frames the slot has decoded since sva_streams_begin. */
int sva_stream_codes(sva_batch* b, int slot, int n, int32_t* codes_out, long* n_frames_total);
/* per-stage device time of the last sva_step in ms: [encoder, ar, vocoder, total] (hipEvents) */
int sva_get_timings(sva_batch* b, float ms[4]);
/* dominant-kernel bookkeeping for bench.py: number of conv-GEMM launches and their summed
 * algorithmic FLOPs in the last step */
int sva_get_gemm_stats(sva_batch* b, double* flops, long* launches);
/* ... and their summed ALGORITHMIC bytes (every operand element once: input rows incl. the tap halo, weights, outputs, residual) */
int sva_get_gemm_bytes(sva_batch* b, double* bytes);

/* roofline leg of bench.py: bracket every conv-GEMM launch of the following steps with hipEvents on the
 * engine stream; sva_get_gemm_profile returns the summed kernel time and the launch count since enabling */
int sva_profile_gemm(sva_batch* b, int enable);
int sva_get_gemm_profile(sva_batch* b, double* total_ms, long* launches);
/* per-launch rows (M, N, K, taps, mode bits, microseconds) of the profiled steps; returns the number of rows */
long sva_get_gemm_profile_table(sva_batch* b, double* out, long max_rows);

/* ---- prompt path: device primitives of the two speaker-embedding encoders (SURVEY.md 8f N1 iii / iv) --------------------------
 * calculate_style_vec (evaluations/infer_arvc.py:179-211: Kaldi fbank -> CAM++, modules/campplus/DTDNN.py:50-137) and
 * calculate_timbre_latent (:213-223: MelSpectrogram -> ECAPA-TDNN -> PerceiverResampler -> FSQ,
 * modules/bicodec_speaker_encoder/speaker_encoder.py:136-144) run once per utterance; the host mirror
 * (streamvoiceanon_amd/prompt_encoders.py) owns the activation buffers and the topology, these entry points do the arithmetic
 * on device arrays (channel-last rows [T][C], row strides in floats; every pointer below is a DEVICE pointer from sva_dev_alloc
 * unless it says host).  All of them run on ONE engine-owned stream, in call order (sva_dev_upload / _download are ordered with them and
 * return when their copy is done).  That stream is non-blocking: nothing here is ordered with the caller's own streams, the legacy default stream or
 * a batch's streams -- a buffer produced elsewhere must be complete (sva_sync / hipStreamSynchronize of its producer) before it is passed in, and a
 * result consumed elsewhere must be fetched with sva_dev_download (which waits) or after a hipDeviceSynchronize. */
int sva_dev_alloc(sva_engine* e, long n_floats, float** out);            /* zero-initialised */
int sva_dev_free(sva_engine* e, float* p);
int sva_dev_upload(sva_engine* e, float* dst, const float* host_src, long n_floats);
int sva_dev_download(sva_engine* e, float* host_dst, const float* src, long n_floats);
/* nn.Conv1d / nn.Linear: y[t][n] = bias[n] + sum_{tap,c} x[(t*stride + tap*dil)*ldx + c] * W[n][tap*Cin + c]; x points at tap 0 of
 * output row 0 (the caller pads); f32-MFMA conv-GEMM when Cin % 16 == 0, a plain kernel otherwise */
int sva_op_conv(sva_engine* e, const float* x, long ldx, int T, int stride, int dil, int taps, int Cin, const float* W, const float* bias, int N,
                float* y, long ldy);
/* y = post(scale[c] * pre(x) + shift[c]); relu_mode 0 none, 1 ReLU after (batchnorm-relu), 2 ReLU before (Conv1dReluBn); eval-mode
 * BatchNorm arrives folded into scale / shift (either may be NULL) */
int sva_op_affine(sva_engine* e, const float* x, long ldx, int T, int C, const float* scale, const float* shift, int relu_mode, float* y, long ldy);
int sva_op_unary(sva_engine* e, float* x, long n, int op /* 1 log(max(x, p0)), 2 FSQ level-4 quantise, 3 sigmoid, 4 negate */, float p0);
int sva_op_colstats(sva_engine* e, const float* x, long ldx, int T, int C, float* mean, float* std_or_null, int unbiased);
int sva_op_cam_context(sva_engine* e, const float* y, long ldy, int T, int C, int seg_len, const float* mean, float* ctx, long ldc);   /* layers.py:103-119 */
int sva_op_mul(sva_engine* e, float* y, long ldy, const float* m, long ldm /* 0 = broadcast one row */, int T, int C, int sigmoid);
int sva_op_add(sva_engine* e, float* y, long ldy, const float* x, long ldx, int T, int C);
/* Conv2d k x k (k = 1 or 3, padding k/2, stride (stride_f, 1)) + folded BatchNorm (+ residual) (+ ReLU) on channel-first [C][F][T] */
int sva_op_conv2d(sva_engine* e, const float* x, int Cin, int F, int T, const float* W, int Cout, int k, int stride_f, const float* scale,
                  const float* shift, const float* res_or_null, int relu, float* y);
int sva_op_cf_to_rows(sva_engine* e, const float* x, int CF, int T, float* y, long ldy);
int sva_op_fbank_power(sva_engine* e, const float* wave, long n, float* frames_scratch /* [m][512] */, float* out, int ldo, int* frames_out /* host */);
int sva_op_stft_mag(sva_engine* e, const float* wave, long n, int n_fft, int win, int hop, float* frames_scratch, float* out, int ldo, int* frames_out /* host */);
int sva_op_attention(sva_engine* e, const float* q, const float* kv, int Lq, int Lk, int n_valid, int H, float* out, float* scratch /* [Lq][H][Lk] */);
int sva_op_geglu(sva_engine* e, const float* h, long ldh, int T, int Dh, float* out, long ldo);
int sva_op_l2norm(sva_engine* e, const float* x, int T, int C, const float* gamma, float scale, float* y);
/* The op sequence of one speaker-encoder call as a hipGraph: between _begin and _end the sva_op_* calls are recorded, not run (no
 * sva_dev_alloc / _upload / _download in between); _end returns an executable graph that sva_ops_graph_launch replays on the ops stream.
 * The reference computes these encoders once per utterance (evaluations/infer_arvc.py:179-223); the host mirror keeps one graph per
 * reference length (prompt_encoders.py), ~600 launches -> 1. */
int sva_ops_capture_begin(sva_engine* e);
int sva_ops_capture_end(sva_engine* e, void** graph_exec);
int sva_ops_graph_launch(sva_engine* e, void* graph_exec);
int sva_ops_graph_free(sva_engine* e, void* graph_exec);

/* Which decode the batch runs per frame: 1 = the persistent kernel of ar_decode.hip (two streams per launch: <= 6 streams, <= 4 with
 * ar_dtype = 1), 2 = the batched persistent kernel of ar_batch.hip (every stream of the batch in one launch: the larger batches), 0 = the
 * multi-launch decode (sampler edits set, layer sizes other than the reference's, or the residency check at sva_batch_create failed:
 * all workgroups of a persistent launch must fit the AR stream's CUs at once).  A persistent launch whose workgroups are NOT all
 * resident (GPU shared with another process) times out after ~50 ms: the next sva_sync / sva_step returns an error, every later step
 * fails, and sva_prefill_prompt + sva_streams_begin restart the streams on the multi-launch decode. */
int sva_batch_uses_persistent_decode(sva_batch* b);
/* debug options of the process ("key=value,key=value"; the same keys as the SVA_DEBUG environment variable, the engine's only
 * environment input -- listed in csrc/sva_common.h: ar_timing, pipe_trace, concurrency, ar_persistent, voc_fused_mask, autotune,
 * tune_log, tune_table, tune_dump).  Profiling and test switches; batches created afterwards see the change. */
int sva_debug_configure(const char* kv);
/* test hook: set the persistent kernel's device-side timeout word, as a launch with non-resident workgroups would */
int sva_test_force_ar_timeout(sva_batch* b);

/* The optional sampler edits of decode_one_token_ar (modules/dual_ar_stream.py:1175-1213 -> logits_to_probs :1099-1117):
 *   previous_tokens [1 + num_codebooks][W] int32 (row 0 edits the token head, row cb + 1 codebook cb; negative entries are
 *   skipped): repetition penalty  s < 0 ? s * p : s / p  on the listed tokens, each once;
 *   suppress_tokens [n_suppress]: token-head logits set to -inf (:1189 passes the list to the first sample() only).
 * They apply to every stream of the batch from the next decoded frame on (sva_generate: from frame 1 -- the prefill's decode
 * takes no sampling_kwargs, :722); W = 0 and n_suppress = 0 remove them.  While edits are set the batch decodes with the
 * multi-launch path (the persistent kernel has no edit stage); at most 4096 entries per list. */
int sva_set_sampler_edits(sva_batch* b, const int32_t* previous_tokens, int W, float repetition_penalty, const int32_t* suppress_tokens,
                          int n_suppress);

/* kernel unit-test hook: C = A[M,K] * W[N,K]^T (+bias) through the conv-GEMM kernel (host arrays) */
int sva_test_gemm(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C);

/* same through one specific dispatch choice of the GEMM dispatcher (kind 0: small-M K-split kernel, a = 16-row tiles per
 * workgroup, b = K-split waves, c = 16-column tiles per wave; kind 1: LDS-tiled f32-MFMA kernel, a = tile variant 0..7; kind 2: the
 * small-M kernel with its K axis also split over c >> 4 workgroups (fence-free tagged hand-off), c & 15 = column tiles, launched
 * twice; kind 3: the register-staged pipelined f32-MFMA kernel (gemm_pipe.hip), a = tile variant 0..6, needs K % 64 == 0; kind 4: the
 * six-product split-bf16 kernel (gemm_split.hip), a = tile variant 0..4, needs 16-byte aligned operands) */
int sva_test_gemm_choice(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C, int kind,
                         int a, int b, int c);
/* fp16-weight GEMM of the batched fp16 AR decode (csrc/gemm_f16w.hip; the reference's autocast(fp16) linear layers,
 * modules/dual_ar_stream.py:1168-1219): C = epi(A x fp16(W)^T), mode bits 1 = RMSNorm prologue, 2 = residual, 4 = SwiGLU pairs;
 * iters > 0 also returns the average microseconds per launch. */
int sva_test_gemm_f16w(int device, int M, int N, int K, const float* A, const float* W, const float* bias, const float* rms_w,
                       const float* res, int mode, float* C, int iters, float* out_us);
/* conv-GEMM fed from pre-split 16-bit operand planes (csrc/gemm_planes.hip; the Linear layers of firefly.py:421-440 and
 * windowed_transformer.py:134-143 at batch scale): mode 0 = three bf16 planes / six products (fp32-grade), 1 = two fp16 planes /
 * three products (fp32-grade inside the fp16 range), 2 = one fp16 plane (torch.autocast(fp16), evaluations/infer_arvc.py:493);
 * variant = tile variant 0..5; flags: 1 = A handed over as planes, 2 = C leaves as planes only (summed back to fp32 for the caller),
 * 4 = GELU epilogue, 8 = SiLU on load, 16 = range check (a non-finite output of the fp16 formats fails the call, as sva_sync does for a batch);
 * iters > 0 also returns the average microseconds per launch. */
int sva_test_gemm_planes(int device, int M, int N, int K, const float* A, const float* W, const float* bias, float* C, int mode,
                         int variant, int flags, int iters, float* out_us);
/* Prefill attention of the slow AR (modules/dual_ar_stream.py:338-356 with causal_mask[kv_pos]): M query rows [M][H*64] at positions
 * pos0 .. pos0 + M - 1 against keys / values [pos0 + M][H*64] placed in a cache of S positions (fp32, or fp16 when half_kv).
 * out_ref: the per-row kernel, out_mfma: the flash-style MFMA kernel; us[2] their launch times when iters > 0. */
int sva_test_prefill_attention(int device, int M, int H, int pos0, int S, const float* q, const float* keys, const float* vals,
                               int half_kv, float* out_ref, float* out_mfma, int iters, float* us);
/* the same rows through the decode frame's paired attention kernel (rows 2 i, 2 i + 1 share one pass over the slot's K / V; M even) */
int sva_test_pair_attention(int device, int M, int H, int pos0, int S, const float* q, const float* keys, const float* vals,
                               int half_kv, float* out_ref, float* out_mfma, int iters, float* us);

/* host cost (microseconds) of enqueueing one kernel from the calling thread, measured over `iters` launches of a one-element
 * kernel into an idle stream.  A synchronous single-stream step is ~170 launches (a pipelined one: four graph launches + the persistent
 * AR kernel), so the enqueueing thread's launch rate still matters for the synchronous mode
 * rate; on a multi-socket host it depends on the core the thread runs on -- see engine.py pin_enqueue_thread() */
int sva_host_launch_cost(int device, int iters, float* us_per_launch);

/* kernel unit-test hook: the nucleus sampler of decode_one (modules/dual_ar_stream.py:1092-1132) over logits[rows][V] with
 * explicit Exp(1) draws noise[rows][V], through one implementation: 1 LDS bitonic sort, 2 register sort, 3/4/5 threshold
 * bisection (workgroup shapes), 6/7 bisection with interpolated, key-snapped probes.  us_out (may be NULL) = average microseconds per launch over `iters` launches */
int sva_test_sampler(int device, int variant, int rows, int V, const float* logits, const float* noise, float temperature,
                     float top_p, int* tok_out, int iters, float* us_out);

/* microbenchmark of the conv-GEMM dispatcher: conv over [B][(taps-1)*dil + T][Cin] -> [B][T][N]; mode bits:
 * 1 GELU, 2 gamma+residual, 4 SiLU-on-load, 8 SwiGLU (w13); returns avg microseconds per launch in out_us[0] */
int sva_bench_gemm(int device, int B, int T, int N, int Cin, int taps, int dil, int mode, int iters, float* out_us);
/* One dispatch choice of the same kernel family on device-resident random data with `nrot` rotating weight copies (cold weights); kind -1 = the
   dispatcher, 0 / 1 / 2 / 4 / 6 = small-M / tiled / pipelined / split-bf16 / weight-streaming kernel with parameters (a, b, c) as in
   csrc/testhooks.hip.  out[0] = us per launch eager, out[1] = replayed as one graph, out[2] = max |C - C_dispatcher|, out[3] = max |C_dispatcher|. */
int sva_bench_gemm_choice(int device, int B, int T, int N, int Cin, int taps, int dil, int mode, int kind, int a, int b, int c, int nrot, int iters,
                          float* out);

#ifdef __cplusplus
}
#endif
#endif /* SVA_H */
