// Engine / batch data structures of the sva HIP engine.
#pragma once
#include "../../include/sva.h"
#include "kernels.h"
#include "ar_decode.h"
#include "ar_batch.h"

#include <array>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace sva {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    long numel() const {
        long n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

// Device arena: weights and workspaces are carved out of a few large hipMalloc chunks instead of ~1000 small allocations,
// so the driver can map them with large page fragments (fewer TLB misses when a step streams ~0.5 GB of weights through
// hundreds of short kernels) and creation / teardown cost a handful of runtime calls.
struct DevPool {
    std::vector<void*> chunks;
    char* cur = nullptr;
    size_t left = 0;
    size_t chunk_bytes = (size_t)256 << 20;
};

// [N][K] row-major weight (K = taps * Cin, tap-major) + optional bias [N]
constexpr int EDIT_CAP = 4096;           // previous tokens per head / suppressed tokens (sva_set_sampler_edits)

struct Lin {
    float* W = nullptr;
    float* Wk = nullptr;         // the same values in the fragment-major packing of gemm_stream.hip (K % 16 == 0)
    void* Wh = nullptr;          // fp16 copy in the same [N][K] layout (AR layers of an ar_dtype = 1 engine)
    float* b = nullptr;
    int N = 0, K = 0;
    unsigned short* Wp = nullptr;    // pre-split 16-bit planes [planes][N][K] of W * 2^e (gemm_planes.hip), wp_inv = 2^-e, pmode = their PlanesMode
    float wp_inv = 1.f;
    int pmode = -1;
};

// ConvNeXtBlock (modules/vqgan/modules/firefly.py:375-440)
struct CNX {
    float *dwT = nullptr, *dwb = nullptr, *lnw = nullptr, *lnb = nullptr, *gamma = nullptr;
    Lin pw1, pw2;
    int C = 0;
};

// Llama-style block; w13 = rows of w1/w3 interleaved in groups of 16 (SwiGLU fused in the epilogue).
// ls_* are the LayerScale gammas of the BSQ pre-transformer (null for the AR).
struct TrLayer {
    float* attn_norm = nullptr;
    float* ffn_norm = nullptr;
    float* ls_attn = nullptr;
    float* ls_ffn = nullptr;
    Lin wqkv, wo, w13, w2;
    // AR layers only: the persistent decode kernel's copies (fp32: wqkv / wo / w2 alias the Lin weights; fp16 with ar_dtype = 1),
    // m_w13 in that kernel's own row order (ar_decode.h)
    void *m_wqkv = nullptr, *m_wo = nullptr, *m_w13 = nullptr, *m_w2 = nullptr;
};

// activation tensor [B, H + Tmax, C] channel-last with H history / zero-pad rows in front
struct Act {
    float* p = nullptr;
    int H = 0, C = 0;
    long rows = 0;       // H + Tmax
    long bstride = 0;    // rows * C
};

// per-layer streaming state of the exact-incremental encoder (newest 4c mel frames per step)
struct EncStream {
    float* mag = nullptr;                  // [B][nm][1088]
    Act mel;                               // [B][6+nm][160]
    Act tmp0;                              // stem conv output [B][nm][128]
    std::vector<std::vector<Act>> x;       // x[i][j] = input of block j of stage i  [B][6+nm][C_i]
    Act xout[4];                           // output of the last block of stage i    [B][nm][C_i]
    Act feat;                              // [B][nm][512]
    Act d1, d1o, d2;                       // [B][6+nm/2][512], [B][nm/2][512], [B][6+nm/4][512]
    float *h1 = nullptr, *h2 = nullptr;    // ConvNeXt scratch
    ShiftDesc* d_shift = nullptr;
    int n_shift = 0;
};

// Merged incremental front-end pass: the newest 4c mel frames ride in the SAME launches as the head pass (they use the same
// weights).  Every tensor that feeds a k7 conv is [B][6 zero rows | Hh head rows | 6 history rows | n new rows][C]: a GEMM over
// Hh + 6 + n rows computes head and new rows together, its epilogue skips the 6 history rows of the destination
// (ConvGemm::skip_lo/hi), and after the step the newest 6 rows of (history ++ new) become the next step's history.
struct EncMerged {
    int Hh = 0, nm = 0;                    // head rows (mel rate), new mel frames per step
    float* mag = nullptr;                  // [B][R0][1088]                  R0 = Hh + 6 + nm
    Act mel;                               // [B][6 + R0][160]
    float* stem = nullptr;                 // [B][R0][128]
    std::vector<std::vector<Act>> x;       // x[i][j] = input of block j of stage i  [B][6 + R0][C_i]
    Act xout[4];                           // output of the last block of stage i    [B][R0][C_i]
    Act feat;                              // [B][R0][512]
    Act feat2;                             // second copy: the pipelined step alternates (main chain writes, side chain reads)
    float *h1b = nullptr, *h2b = nullptr;  // ConvNeXt scratch of the downsampler when it runs on the side chain [B][R1][512], [B][R1][2048]
    Act d1, d1o, d2, tok;                  // [B][6 + R1][512], [B][R1][512], [B][6 + R2][512], [B][R2][512]   R1 = Hh/2 + 6 + nm/2 ...
    float *h1 = nullptr, *h2 = nullptr;    // ConvNeXt scratch [B][R0][512], [B][R0][2048]
    ShiftDesc* d_shift = nullptr;
    int n_shift = 0;
};

// ConvNeXtEncoder (firefly.py:443-520) + the quantizer's 2x (conv k2 s2 + ConvNeXtBlock) downsampler: the tokenizer
// front-end and the vocoder's own encoder (firefly.encode, firefly.py:560-574) share this shape
struct EncFront {
    Lin stem;                              // conv k7 160 -> 128
    float *stem_lnw = nullptr, *stem_lnb = nullptr;
    std::vector<std::vector<CNX>> stages;
    float* trans_lnw[4] = {nullptr, nullptr, nullptr, nullptr};
    float* trans_lnb[4] = {nullptr, nullptr, nullptr, nullptr};
    Lin trans[4];
    float *final_lnw = nullptr, *final_lnb = nullptr;
    Lin ds_conv[2];
    CNX ds_cnx[2];
    bool loaded = false;
};

struct ResConv {
    Lin c1, c2;
    int k = 0, dil = 1;
    // narrow levels (C = 16 / 32): weight planes for voc_conv_kernel, K-blocked with K = k * C padded to whole 32-k blocks (C = 32: c1.Wp / c2.Wp themselves)
    unsigned short *q1 = nullptr, *q2 = nullptr;
    float q1_inv = 1.f, q2_inv = 1.f;
    int Kq = 0;
};

}  // namespace sva

struct sva_engine {
    sva_config cfg;
    int device = 0;
    // persistent AR decode launches of DIFFERENT batches of this engine are chained (event): two half-resident persistent grids
    // waiting for each other's CUs would only end at their spin timeouts
    hipStream_t ops_stream = nullptr;      // stream of the prompt-path primitives (prompt_ops.hip: created on first use), capturable
    hipEvent_t mega_ev = nullptr;          // created by sva_engine_finalize, destroyed with the engine
    bool mega_ev_valid = false;
    int persistent_batches = 0;            // live batches that decode with a persistent kernel (under mega_mu): a lone one skips the chain's event record per frame
    const void* mega_last = nullptr;       // (cleared when that batch is destroyed)
    std::mutex mega_mu;                    // batches of one engine may be driven by different host threads
    bool finalized = false;
    std::unordered_map<std::string, sva::HostTensor> host;
    sva::DevPool allocs;

    // ---- content encoder ----
    sva::Lin mel_fb;                       // [160][1088]  (K padded 1025 -> 1088)
    float2* twiddle = nullptr;             // [1024]
    float* hann = nullptr;                 // [2048]
    sva::EncFront tokf;                    // tok.backbone + tok.quantizer.downsample
    std::vector<sva::TrLayer> tr;
    float* tr_norm = nullptr;
    float* rope_enc = nullptr;             // [2048][32][2]
    float *bsq_W = nullptr, *bsq_b = nullptr;

    // ---- dual AR ----
    float *content_emb = nullptr, *codebook_emb = nullptr, *fast_emb = nullptr, *wait4start = nullptr;
    std::vector<sva::TrLayer> ar_layers, ar_fast_layers;
    float *ar_norm = nullptr, *ar_fast_norm = nullptr;
    sva::Lin ar_output, ar_fast_output, context_in, style_in;
    float *rope_ar = nullptr, *rope_fast = nullptr;
    void *m_output = nullptr, *m_fast_output = nullptr;     // heads in the persistent decode kernel's element type
    bool mega_ok = false;                  // layer counts / sizes match what ar_decode.hip is built for
    // the front-end's steady response to silence (what sva_streams_begin initialises the per-layer histories and the token cache
    // with), computed once per (streams, chunk) configuration by streaming zeros and reused by later batches: one row per history
    // buffer (in steady state every row of a layer is the same vector) + the token row
    struct SilenceState { std::vector<float*> rows; float* tok = nullptr; };
    std::map<std::pair<int, int>, SilenceState> silence;

    // ---- vocoder ----
    float *fsq_W = nullptr, *fsq_b = nullptr;   // [8][64][4], [8][64]
    sva::EncFront vocf;                    // voc.backbone + voc.quantizer.downsample (prompt path, optional)
    float *fsq_in_W = nullptr, *fsq_in_b = nullptr;   // [8][4][64], [8][4]   residual_fsq.rvqs.g.project_in
    sva::Lin up_conv[2];
    sva::CNX up_cnx[2];
    sva::Lin conv_pre;
    sva::Lin ups[5];
    int ups_k[5] = {16, 16, 4, 4, 4};
    int ups_s[5] = {8, 8, 2, 2, 2};
    sva::ResConv res[5][3][3];
    float *post_w = nullptr, *post_b = nullptr;  // [13][16], [1]
    int post_k = 13, pre_k = 13;
};

struct sva_batch {
    sva_engine* e = nullptr;
    sva_stream_params p;
    int B = 0;
    hipStream_t stream = nullptr;          // current launch stream (== main except inside a forked branch)
    hipStream_t main_stream = nullptr;
    hipStream_t aux[2] = {nullptr, nullptr};   // side streams for independent sub-chains (fork / join by events)
    // stage pipelining over consecutive chunk-steps (sva_step_device, p.pipeline): AR and vocoder streams, hand-off events
    hipStream_t sa = nullptr, sv = nullptr;
    hipEvent_t pipe_evD2C = nullptr;       // recorded once the transformer of a step no longer reads the token cache
    hipEvent_t tr_l0_event = nullptr;      // enc_transformer records this after its first layer's output projection
    int stream_cut = 1;                    // pipelined encoder: stages of the streaming pass that run on the main stream (0..3)
    int pipe_split_e = 1;                  // 0: encoder as one in-order stage
    std::vector<hipEvent_t> trace_ev;     // SVA_PIPE_TRACE=N: timestamps of the stage chains of the last N pipelined steps (debug)
    long trace_steps = 0;
    int trace_n = 0;
    int* d_step_x = nullptr;               // the side chain's copy of the chunk counter (pipelined encoder)
    hipStream_t out_stream = nullptr;      // stream that holds the PCM of the last step
    bool allow_pipe = false, pipe_dirty = false;
    hipEvent_t pipe_evVc[2] = {nullptr, nullptr}, pipe_evR = nullptr, pipe_evA[2] = {nullptr, nullptr};
    int* d_step_audio_buf[2] = {nullptr, nullptr};
    long long* d_codes_buf[2] = {nullptr, nullptr};
    int pipe_parity = 0;
    hipEvent_t evpool[64];
    int evi = 0;
    bool concurrency = true;
    bool fused_decode = true;              // B <= 2: GEMV path with fused norm / RoPE / KV-write / SwiGLU
    bool ar_failed = false;                // a persistent launch timed out: every step fails until sva_streams_begin, which falls back to the multi-launch decode
    bool use_mega = false;                 // B == 1: one persistent kernel per decoded frame (ar_decode.hip)
    int mega_per_launch = 1;               // streams per persistent launch (2 when 192 workgroups find a CU each)
    bool ar_partitioned = false;           // the AR stream has a CU partition of its own
    unsigned long long* d_gran = nullptr;  // its granule buffers (gx | gbig | gatt | glog)
    unsigned* d_epoch = nullptr;           // [1] running phase counter of the granule tags
    int* d_ar_fail = nullptr;              // [1] timeout code of the persistent kernel (0 = healthy)
    int* h_mm_ovf = nullptr;               // host-mapped: an fp16-planes GEMM of this batch produced a non-finite output (gemm_planes.hip)
    int* d_mm_ovf = nullptr;               //   ... its device address
    int* h_ar_fail = nullptr;              // host-mapped mirror of *d_ar_fail (written by the kernels at the end of a launch that saw it set)
    int* d_ar_fail_host = nullptr;         //   ... its device address
    long long* d_ar_dbg = nullptr;         // SVA_AR_TIMING=1: phase timestamps of workgroup 0
    float* kv_fast_mega = nullptr;         // [4][8][2][768] fast-AR K/V scratch of the persistent kernel
    // batched persistent decode kernel (ar_batch.hip): every stream of the batch in ONE launch per frame
    int enc_cus = 0;                       // CUs of the encoder / vocoder streams' mask when the batch is CU-partitioned (0: the whole device)
    bool use_abatch = false;
    bool counted_persistent = false;       // this batch is counted in sva_engine::persistent_batches
    int abatch_G = 0;                      // workgroups of its launch (all co-resident: checked at batch creation)
    unsigned long long* d_ab_gran = nullptr;      // hand-off granules, arrays at ab_offs (ar_batch_granule_words)
    size_t ab_offs[11] = {};
    unsigned* d_ab_epoch = nullptr;        // [2] running phase counter | exit counter
    bool kv_half = false;                  // slow KV cache holds __half (ar_dtype = 1)
    sva::DevPool allocs;

    // ---- device control block ----
    int* d_step = nullptr;                 // chunks consumed (ring position)
    int* d_last_pos = nullptr;             // [B] last written slow-AR KV position
    int* d_nframes = nullptr;              // [B] decoded frames (sampler noise counter)
    int* d_ncontent = nullptr;             // [B] content codes seen
    unsigned long long* d_seed = nullptr;  // [B]
    int* d_use_forced = nullptr;           // [1]
    // sampler edits (sva_set_sampler_edits): previous_tokens [1 + num_codebooks][EDIT_CAP], suppress list, {W, n_suppress, penalty}
    int* d_edit_prev = nullptr;
    int* d_edit_suppress = nullptr;
    int* d_edit_params = nullptr;          // [1 + num_codebooks][4]
    bool edits_on = false, edits_skip = false;
    int mega_max = 6;                      // most streams the persistent decode kernel serves for this engine's AR dtype
    int ar_cus = 0;                        // CUs of the AR stream's mask when the pipelined mode is partitioned (0: whole chip)

    // ---- encoder workspace ----
    int We = 0, N = 0, T0 = 0, T2 = 0;
    float* ring = nullptr;                 // [B][N]
    float* d_chunk = nullptr;              // [B][2048*c] staged input
    float* mag = nullptr;                  // [B][T0][1088]
    sva::Act mel;                          // [B][6+T0][160]
    sva::Act xs[4];                        // stage activations with 6 pad rows
    float *h1 = nullptr, *h2 = nullptr;    // [B][T0][512], [B][T0][2048]
    sva::Act feat;                         // [B][T0][512]
    sva::Act d1, d2;                       // [B][6+T0/2][512], [B][6+T0/4][512]
    float *tr_hn = nullptr, *tr_qkv = nullptr, *tr_att = nullptr, *tr_g = nullptr, *tr_z = nullptr;
    float* tr_x = nullptr;                 // [B][T2][512] transformer work copy
    sva::Act d2c;                          // [B][T2][512] steady token cache of the exact-incremental encoder
    sva::EncStream es;
    sva::EncMerged em;
    bool enc_merged = true;                // new frames folded into the head-pass launches (SVA_ENC_MERGED=0: separate streaming pass)
    sva::ShiftDesc* d_shift_d2c = nullptr;
    int Ht = 40;                           // head tokens recomputed every chunk (receptive field 38.25 tokens)
    bool enc_incremental = true;
    long long* d_codes = nullptr;          // [B][T2]
    int* d_fsq_codes = nullptr;            // [B][8][T2] firefly.encode output
    float* d_u = nullptr;                  // [B][T2][13]

    // ---- AR workspace ----
    int Mmax = 0;
    float *ax = nullptr, *ahn = nullptr, *aqkv = nullptr, *aatt = nullptr, *ag = nullptr;
    float* aatt_part = nullptr;            // [4][H][8][68] split-key attention partials of the fused decode path
    float *xf = nullptr;                   // [B][dim] fast-AR residual stream
    float *slow_logits = nullptr, *fast_logits = nullptr, *hidden = nullptr;
    int *d_slot = nullptr, *d_pos = nullptr;          // [Mmax]
    int *d_fast_slot = nullptr, *d_fast_pos = nullptr; // [B], [8][B]
    void* kv_slow = nullptr;               // [L][B][2][H][S][64]
    void* kv_fast = nullptr;               // [Lf][B][2][H][8][64]
    long kv_slow_slot = 0, kv_slow_layer = 0, kv_fast_slot = 0, kv_fast_layer = 0;
    float* cached_audio_emb = nullptr;     // [B][dim]
    float* cached_ref_emb = nullptr;       // [B][max_delay][dim]
    int* d_ref_tail = nullptr;             // [B][8][max_delay] last frames of the stored prompt's audio codes (device-side re-prefill)
    float* spk = nullptr;                  // [33][dim] scratch
    float *d_style = nullptr, *d_timbre = nullptr;     // [B][192], [B][32][128]
    int* d_sem = nullptr;                  // [B] semantic token (discarded by callers)
    int* d_tok = nullptr;                  // [B][8] sampled fast tokens of the current frame
    int* d_tok_raw = nullptr;              // [B][8] sampled (before teacher forcing)
    int* d_forced = nullptr;               // [B][8][chunk]
    float* d_noise = nullptr;              // [B][chunk][vocab + 8*cb]
    bool noise_on_device = false;
    int h_use_forced = -1;                 // host mirror of d_use_forced (-1 = unknown): skips the per-step reset launch
    // histories (per slot, linear with device counters)
    int hist_cap = 0;
    int* d_slot_list = nullptr;            // [B] scratch list of slots for partial delay fills
    int* d_content_hist = nullptr;         // [B][hist_cap] ring
    int* d_pred_hist = nullptr;            // [B][8][hist_cap]
    int* d_step_content = nullptr;         // [B][chunk] content codes of this step (int32)
    int* d_step_audio = nullptr;           // [B][8][chunk]
    // prompt (truncated to max_prompt_frames) kept for re-prefill / vocoder priming
    std::vector<std::vector<int64_t>> ref_content;   // [B][R']
    std::vector<std::vector<int32_t>> ref_audio;     // [B][8*R']
    std::vector<int> ref_len;
    int* d_prompt_cc = nullptr;            // [Pmax] scratch prompt content codes (int32)
    int* d_prompt_ac = nullptr;            // [8][Pmax]
    int Pmax = 0;

    // host mirrors of the deterministic per-slot state
    std::vector<int> h_last_pos, h_nframes;
    int h_ncontent = 0;                    // content codes seen (lock step)
    int h_step = 0;
    bool delay_filled = false;
    bool begun = false;
    std::vector<char> prefilled;

    // ---- vocoder workspace ----
    int Tv = 0;                            // max code frames per call
    sva::Act zq;                           // [B][Tv][512]
    sva::Act u0, v0, u1, pin;              // upsample stack, conv_pre input
    float *vh1 = nullptr, *vh2 = nullptr;  // ConvNeXt scratch
    sva::Act S[6];                         // S[i] = input of ups.i (i<5); S[5] = conv_post input
    sva::Act X[5];                         // resblock inputs
    sva::Act tb[5][3][3];                  // c1 outputs
    sva::Act yb[5][3][2];                  // y_{b,1}, y_{b,2}
    sva::Act y3[5][3];                     // y_{b,3}: branch outputs before the ParallelBlock mean (no history)
    // wide levels on the LDS-DMA planes kernel (stages.hip, vocode): the conv inputs silu(X), silu(tb), silu(yb) live as K-blocked operand
    // planes over the same dense rows as the fp32 tensors above (history rows included: they are the streaming state of such a level)
    int voc_dma[5] = {0, 0, 0, 0, 0};        // 1: K-blocked planes + the LDS-DMA GEMM's conv form (C >= 64); 2: row-major planes + voc_conv_kernel (C = 16 / 32)
    int voc_pmode = -1;
    unsigned short* XP[5] = {};
    unsigned short* tbP[5][3][3] = {};
    unsigned short* ybP[5][3][2] = {};
    bool voc_grouped = true;               // the three ResBlock branches of a level share one launch per conv stage
    float* d_pcm = nullptr;                // [B][2048*Tv]
    // per-step redirections of the device-buffer step (no staging copies): chunk source, PCM destination, codes source
    const float* step_src = nullptr;       // ring_write reads the caller's chunk directly
    float* pcm_dst = nullptr;              // conv_post_tanh writes straight into the caller's buffer ([B][2048*chunk])
    long pcm_dst_bstride = 0;
    const int* voc_codes = nullptr;        // FSQ decode reads the step's audio codes in place
    long voc_codes_bstride = 0, voc_codes_gstride = 0;
    hipEvent_t voc_codes_event = nullptr;  // recorded right after the FSQ decode has been enqueued (pipelined hand-off)
    bool pcm_direct_done = false;
    int* d_vcodes = nullptr;               // [B][8][Tv]
    std::vector<sva::ShiftDesc> shift_host;
    sva::ShiftDesc* d_shift = nullptr;

    // pinned staging
    float *hp_in = nullptr, *hp_out = nullptr;

    // timing / stats
    hipEvent_t ev[5];
    bool ev_ok = false;
    float last_ms[4] = {0, 0, 0, 0};
    double gemm_flops = 0, gemm_bytes = 0;
    long gemm_launches = 0;
    bool prof_on = false;
    int prof_n = 0;
    std::vector<hipEvent_t> prof_ev;
    std::vector<std::array<int, 5>> prof_shapes;

    // graph
    hipGraphExec_t graph_exec = nullptr;
    hipGraphExec_t pipe_graph_a[2] = {nullptr, nullptr};     // pipelined mode: the AR stage of a step, one per code-buffer parity
    int pipe_graph_mode = 1;                                 // 0: AR stage enqueued kernel by kernel
    // pipelined mode: the front-end chain, the transformer (cut after its first layer, where it releases the token cache) and the
    // vocoder (behind the FSQ decode, where it releases the step's codes) replayed as hipGraphs: ~135 launches per step off the
    // enqueueing thread, which is otherwise the bound once the GPU side of a single-stream step drops to ~1 ms
    bool stage_graphs = true;
    hipGraphExec_t gEm[2] = {nullptr, nullptr}, gEs[2] = {nullptr, nullptr};      // front-end cut behind the backbone: main part / side part, per parity
    hipEvent_t pipe_evFeat[2] = {nullptr, nullptr};
    int enc_cut = 1;
    int* step_bump = nullptr;              // set around a front-end call whose last kernel should also advance this counter
    int voc_fused_mask = -1;               // -1: default policy; else bit 0: the C = 16 level, bit 1: the C = 32 level
    bool voc_fused = true;                 // narrow vocoder levels (C <= 32) as one fused launch (voc_fused.hip)
    int* d_voc_frames = nullptr;           // code frames the streaming vocoder has consumed since its last reset
    int voc_rpf[5] = {0, 0, 0, 0, 0};      // rows of level i per code frame
    hipGraphExec_t gE = nullptr, gE2 = nullptr, gT0 = nullptr, gT1[2] = {nullptr, nullptr}, gV = nullptr;
    bool graph_ready = false;
    bool graph_step = false;       // last step ran through the graph (no per-stage events)
    bool forced_now = false;
    int steady_eager_steps = 0;
};
