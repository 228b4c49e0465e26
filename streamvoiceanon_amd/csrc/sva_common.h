// Common declarations of the sva HIP engine (gfx950 / MI355X only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <string>

namespace sva {

void set_error(const std::string& msg);

// The engine's only environment input: SVA_DEBUG="key=value,key=value" (read once per process).  Profiling, table generation and the
// parity tests' kernel selection -- nothing a deployment sets; every kernel choice of the product path is compiled in.
//   ar_timing=1      persistent AR kernel stamps its phases (tap "ar_timing"; tools/ar_timing.py)
//   pipe_trace=N     per-chain start / end events of the last N pipelined steps, printed when the batch is destroyed
//   concurrency=0    one stream for everything (rocprofv3 --pmc serialises dispatches: tools/pmc.sh, bench.py's traffic passes)
//   ar_persistent=0  multi-launch AR decode instead of the persistent kernel (A/B, parity of the fallback)
//   ar_batch=0|2     batched persistent decode kernel (ar_batch.hip) off / for every batch size incl. the ones ar_decode.hip serves (A/B, parity)
//   ar_batch_wgs=N   workgroups of its launch (default: one per 16-column tile of the widest phase, capped by what is resident)
//   voc_fused_mask=M which narrow HiFiGAN levels run as the fused kernel (bit 0: C = 16, bit 1: C = 32; parity tests)
//   autotune=1, tune_log=1, tune_table=0, tune_dump=PATH   timed search / its log / ignore the compiled-in table / dump the choices
//                    (tools/make_tune_table.py); tune_kinds=MASK: only table rows whose kernel kind has its bit set are honoured (bisection)
//   cu_partition=0|1  force the disjoint CU masks of the pipelined mode (AR 96 | encoder + vocoder 160) off / on (default: by stream count)
//   cu_ar=N           with cu_partition=1: CUs of the AR stream's mask (default: by batch size; A/B only: the round-3 partition A/B scripts, git history)
//   pipe_skip=mask    TIMING DIAGNOSTIC (results are garbage): leave out a chain of the pipelined step -- 1 encoder front, 2 side chain
//                     (downsampler + transformer + BSQ), 4 AR, 8 vocoder (tools/pipe_skip.sh)
//   planes_dbg=mask   TIMING DIAGNOSTIC (results are garbage): leave parts of the planes GEMM out -- 1 global loads of its K loop, 2 LDS stores, 4 MFMAs,
//                     8 epilogue (tools/planes_probe.py)
//   planes_dma=0      the planes GEMM never takes its persistent LDS-DMA form (variants 9 .. 14): round 4's register-staged tiles (A/B)
//   voc_dma=0|1       the HiFiGAN levels' ResBlock convs on operand planes (C >= 64: the LDS-DMA planes kernel's conv form, C = 16 / 32: voc_conv_kernel;
//                     activations between them as planes): never / at every batch size (default: from 10 code frames per step over the batch; parity tests force it at small batches)
//   ar_graph=0|1      the AR stage of the pipelined mode enqueued kernel by kernel / replayed as a graph (default: a graph unless the stage is ONE
//                     persistent launch -- a one-node graph only adds its replay cost; A/B)
//   ar_pairs=0        the decode frame's slow attention per row instead of per (stream, head) pair of rows (A/B, parity)
//   head_fuse=0       the heads' RMSNorm as a launch of its own instead of the head GEMM's prologue (A/B)
//   planes_min_streams=N  stream count from which the encoder's passes hand their operands over as planes (default 10; A/B)
//   planes_lw=0       the planes GEMM's loader-wave forms (variants 11 / 12 of the table) fall back to variant 10 (A/B)
//   voc_dma_variant=9|10|11|13|14  tile configuration of those convs where 128 x 128 tiles fill the chip (A/B)
//   reprefill=0       re-prefill as one whole-prompt prefill per slot behind a host synchronisation (round 3) instead of one pass over the
//                     appended rows of all due slots against the cached prompt prefix (A/B, parity)
//   f16_weights=0|2   ar_dtype = 1 batched decode on the fp32 copy of the rounded weights instead of gemm_f16w.hip (A/B) / on gemm_f16w.hip even when
//                     the library was built with a compiler the kernel was not validated with
struct DebugOptions {
    int ar_timing = 0, pipe_trace = 0, concurrency = 1, ar_persistent = 1, ar_batch = 1, ar_batch_wgs = 0, voc_fused_mask = -1, autotune = 0, tune_log = 0, tune_table = 1, f16_weights = 1, cu_partition = -1, cu_ar = 0, pipe_skip = 0, planes_dbg = 0, reprefill = 1, planes_dma = 1, voc_dma = -1, voc_dma_variant = 11, planes_lw = 1, planes_min_streams = 10, ar_graph = -1, ar_pairs = 1, head_fuse = 1, tune_kinds = -1;
    std::string tune_dump;
};
const DebugOptions& debug_options();

// hipFuncSetAttribute applies to the CURRENT device only, and one process may hold engines on several GPUs: a per-call-site,
// per-device "done" mask (bit = device ordinal), safe against concurrent first launches (setting the attribute twice is harmless)
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool needed() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        const unsigned long long bit = 1ull << (dev & 63);
        return !(mask.load(std::memory_order_relaxed) & bit);
    }
    void done() {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) mask.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
};

#define SVA_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::sva::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + \
                             __FILE__ + ":" + std::to_string(__LINE__) + ")");               \
            return -2;                                                                       \
        }                                                                                    \
    } while (0)

#define SVA_CHECK(cond, msg)                                                       \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ::sva::set_error(std::string(msg) + " [" #cond "] (" + __FILE__ + ":" + \
                             std::to_string(__LINE__) + ")");                      \
            return -1;                                                             \
        }                                                                          \
    } while (0)

// ---------------------------------------------------------------------------------------
// conv-GEMM: the one matrix kernel of the encoder / vocoder (and the f32 AR mode).
//   C[b, t, n] = epi( sum_{tap, k} A[b, t*stride + tap*dil, k] * W[n, tap*Cin + k] )
// Activations are channel-last [B, rows, C] with a few history / zero-pad rows in front of
// every tensor, so a causal Conv1d is a GEMM whose K axis is split into taps that read
// shifted rows (no im2col, no bounds checks).  ConvTranspose1d (k = 2*stride or k = stride)
// is the same GEMM with N = stride*Cout and 2 (or 1) taps -- see DESIGN.md.
// ---------------------------------------------------------------------------------------
enum ActKind : int { ACT_NONE = 0, ACT_GELU = 1, ACT_LOGCLAMP = 2 };
// operand-plane formats of gemm_planes.hip: H3 = 2 fp16 planes / three products (fp32-grade inside the fp16 range; sva_config::mm_mode = 1),
// H1 = 1 fp16 plane / one product (torch.autocast(fp16); voc_dtype = 1).  (0 was S6, three bf16 planes / six products: removed in round 5)
enum PlanesMode : int { PLANES_S6 = 0, PLANES_H3 = 1, PLANES_H1 = 2 };

struct ConvGemm {
    const float* A = nullptr;   // element (b, r, k): A[b*a_bstride + a_off + r*lda + k]
    long a_bstride = 0;
    long a_off = 0;
    int lda = 0;
    int T = 0;                  // output rows per batch item
    int M = 0;                  // B * T
    int stride = 1, dil = 1, taps = 1, Cin = 0;
    const float* W = nullptr;   // [N][taps*Cin], K contiguous
    const float* Wk = nullptr;  // optional fragment-major packing of W ([N / 16][K / 16][64 lanes][4]: gemm_stream.hip reads one contiguous KiB per wave-instruction)
    const void* Wh = nullptr;   // optional fp16 copy of W (ar_dtype = 1 AR layers): decode-sized problems stream it instead (gemm_f16w.hip)
    int N = 0;
    const float* bias = nullptr;    // [N]
    const float* gamma = nullptr;   // [N]  (ConvNeXt gamma / LayerScale)
    const float* res = nullptr;     // residual, element (b, t, n): res[b*r_bstride + r_off + t*ldr + n]
    long r_bstride = 0;
    long r_off = 0;
    int ldr = 0;
    float* C = nullptr;         // element (b, t, n): C[b*c_bstride + c_off + t*ldc + n]
    long c_bstride = 0;
    long c_off = 0;
    int ldc = 0;
    float scale = 1.f;
    int skip_lo = 0, skip_hi = 0;   // output rows t (within a batch item) in [skip_lo, skip_hi) are computed but not stored: the history
                                    // rows that sit between the head rows and the newest rows of the merged incremental encoder pass
    int act = ACT_NONE;
    int accumulate = 0;         // C += value   (ParallelBlock mean of three ResBlock branches)
    int a_silu = 0;             // apply SiLU to A on load (HiFiGAN: silu precedes every conv)
    int w13 = 0;                // SwiGLU: W rows interleave w1/w3 in groups of 16; C[., n/2] = silu(a)*b
    int ksplit = 1;                 // set by the dispatcher: K split over `ksplit` workgroups (blockIdx.z), partial tiles in
    float* ks_ws = nullptr;         //   ks_ws, arrival counters in ks_cnt; the last workgroup of a tile sums them in
    unsigned* ks_cnt = nullptr;     //   split order (deterministic) and runs the epilogue
    // fused ConvNeXt prologue (small-M kernel, M <= 16, taps == 1): the A rows are LayerNorm(dwconv7(x)) computed on the fly;
    // A then addresses the FIRST tap row (t - 6) of x, dw_wT is [7][Cin] tap-major, ln_* the LayerNorm affine
    const float* dw_wT = nullptr;
    const float* dw_b = nullptr;
    const float* ln_w = nullptr;
    const float* ln_b = nullptr;
    float ln_eps = 1e-6f;
    const float* rms_w = nullptr;   // [Cin] fused RMSNorm of the A rows (taps == 1): A' = A * rms_w * rsqrt(mean(A^2) + rms_eps);
    float rms_eps = 1e-5f;          //       only on the small-M path -- ask conv_gemm_can_fuse_rms() first
    // pre-split 16-bit operand planes (gemm_planes.hip), ALL K-BLOCKED (planes_split.h: [k / 32][rows][32] per plane, so that the 16 rows x
    // 32 k of an MFMA operand piece are one contiguous KiB).  Wp: parts of W * 2^e over N rows, wp_inv = 2^-e.  Ap / Cp: planes of the A / C
    // tensor over ap_rows / cp_rows DENSE rows (row of (b, t) = b * (bstride / ld) + off / ld + t (+ tap * dil): whole rows only; stride = 1;
    // several taps only in the LDS-DMA form), plane p at + p * pstride elements.  A / C may be null when Ap / Cp is set.
    const unsigned short* Wp = nullptr;
    long wp_pstride = 0;
    float wp_inv = 1.f;
    int pmode = -1;                 // PlanesMode of Wp (and of Ap / Cp)
    const unsigned short* Ap = nullptr;
    long ap_pstride = 0;
    long ap_rows = 0;
    unsigned short* Cp = nullptr;
    long cp_pstride = 0;
    long cp_rows = 0;
    int cp_silu = 0;                // LDS-DMA form: Cp receives the parts of silu(v) while C receives v (HiFiGAN: the next conv reads silu of this output)
    int cu_limit = 0;               // CUs the launch's stream may use (0 = the whole device): sizes the persistent grid of the planes-DMA kernel on CU-masked streams
    int* ovf = nullptr;             // fp16 planes only: set to 1 when an output is not finite (an operand outside the fp16 range); host-mapped
};

// up to three independent problems of identical shape (M, N, Cin, stride, epilogue flags; taps / dilation / pointers may
// differ) in ONE launch: blockIdx.z picks the member.  The three ResBlock branches (k = 3, 7, 11) of a HiFiGAN level.
struct ConvGemmGroup {
    ConvGemm g[3];
    int n = 1;
    int xcd_swz = 0;            // tiled kernels: remap the workgroup id so that each XCD owns a contiguous band of M tiles
};

// Workgroup b runs on XCD b % 8 (observed placement; a speed assumption only).  With the default N-fastest tile order the eight
// workgroups that share an A row panel sit on eight XCDs and each of the eight private L2s fetches the panel from the fabric;
// this bijective remap gives XCD k the k-th contiguous eighth of the tile sequence, i.e. a band of M tiles with all its N
// tiles -- the panel is fetched once and re-read from that XCD's L2 (cdna_hip_programming.md, "XCD swizzle must be bijective").
// host side: remap only grids that give every XCD several M bands' worth of tiles
inline int xcd_swizzle_for(unsigned nx, unsigned ny) {
    return nx >= 2 && ny >= 16 ? 1 : 0;
}
__device__ __forceinline__ void xcd_tile(int swz, int nx, int ny, int& bx, int& by) {
    if (!(swz & 1)) return;
    const int T = nx * ny, L = by * nx + bx;
    const int xcd = L & 7, idx = L >> 3, q = T >> 3, r = T & 7;
    const int V = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    by = V / nx; bx = V - by * nx;
}

// gemm_pipe.hip: LDS-DMA ring kernel (Cin % 64 == 0); variant = tile shape, see launch_pipe_gemm
// fused HiFiGAN ParallelBlock level for C = 16 / 32 (voc_fused.hip)
bool voc_level_supported(int C);
int launch_voc_level(const float* X, long x_bstride, int xH, int C, int B, int Tl, const float* const W[3][6], const float* const bias[3][6],
                     const int dil[3], float* const y3[3], long y_bstride, const int* frames_done, int rows_per_frame, hipStream_t st);
int conv_gemm_prepare_stream(hipStream_t st);
int conv_gemm_check_errors();          // split-K hand-off fault word of every stream (checked at sva_sync)
// gemm_split.hip: fp32 GEMM as six bf16 part products on v_mfma_f32_16x16x32_bf16 (variant 0..3 = 128x128, 128x64, 64x128, 64x64)
bool split_gemm_supported(const ConvGemm& g);
int launch_split_gemm(const ConvGemmGroup& gg, int variant, hipStream_t st);
bool pipe_gemm_supported(const ConvGemm& g);
int launch_pipe_gemm(const ConvGemmGroup& gg, int variant, hipStream_t st);
// gemm_planes.hip: 16-bit matrix pipes fed from pre-split operand planes (variant 0..3 = 128x128, 128x64, 64x128, 64x64; 4 / 5 = 128x128 / 64x64 with 64-deep K
// tiles; 6 = 256x128 and 7 = 128x128 on 8 waves)
int planes_count(int mode);
bool planes_gemm_supported(const ConvGemm& g);
int launch_planes_gemm(const ConvGemmGroup& gg, int variant, hipStream_t st);
// variants 9 .. 12 of it: the persistent LDS-DMA-fed form (9 / 10: 128 x 128 tiles, one / two workgroups per CU; 11 / 12: two loader waves beside the eight
// multiplying ones, 128 x 128 / 256 x 128 tiles; A as planes, N % 128 == 0; a group's tiles form one sequence)
bool planes_dma_gemm_supported(const ConvGemm& g);
// its conv form (taps over A planes, a group's tiles as one sequence, SiLU'd output planes; variants 9 / 10 / 11 and the narrow tiles 13 / 14): N % 64 == 0
bool planes_dma_conv_supported(const ConvGemm& g);
void planes_dma_set_cu_limit(int cus);        // CUs its grid may count on (0 = the device's); the engine sets it around launches on CU-masked streams
// Narrow HiFiGAN levels (C = 16 / 32): one ResBlock conv stage of the three branches as a halo-resident conv over ROW-MAJOR operand planes
// ([plane][dense rows][C] fp16 parts): a workgroup keeps its branch's whole weight in LDS and, per 128- / 256-row tile, the input rows with
// their (taps - 1) * dil halo -- every tap's MFMA operand is a shifted read of that one image (gemm_planes.hip: voc_conv_kernel)
struct VocConv {
    const unsigned short* Ap = nullptr;     // input planes, plane p at + p * ap_pstride
    long ap_pstride = 0, a_rows_b = 0, a_row0 = 0;      // dense rows per stream; row of (t = 0, tap 0) inside a stream = H - (taps - 1) * dil
    long a_rows_total = 0;                  // dense rows of the input planes tensor (reads are clamped to it)
    const unsigned short* Wp = nullptr;     // K-blocked weight planes [plane][KS][C][32], K = taps * C padded to whole 32-k blocks
    long wp_pstride = 0;
    float wp_inv = 1.f;
    const float* bias = nullptr;
    const float* res = nullptr;             // fp32 residual rows [B][.][C] (or null)
    long r_bstride = 0, r_off = 0;
    float* Cf = nullptr;                    // fp32 output rows (or null)
    long c_bstride = 0, c_off = 0;
    unsigned short* Cp = nullptr;           // planes of silu(output) (or null), same row geometry as an input: rows per stream, first row
    long cp_pstride = 0, c_rows_b = 0, c_row0 = 0;
    int taps = 1, dil = 1;
};
struct VocConvGroup {
    VocConv g[3];
    int n = 3;
    int B = 0, T = 0;                       // streams, output rows per stream (a multiple of the tile rows)
    int wg0[4] = {0, 0, 0, 0};              // workgroups [wg0[p], wg0[p + 1]) work on member p
    int* ovf = nullptr;
};
bool voc_conv_supported(int C, int T, int mode);
int launch_voc_conv(VocConvGroup gg, int C, int mode, int cu_limit, hipStream_t st);
// fp32 [rows][K] (row stride ld) -> K-blocked planes
int launch_to_planes(const float* src, long rows, int K, long ld, unsigned short* dst, long pstride, int mode, float scale, int silu, hipStream_t st);
// rows [row0, row0 + T) of each of nb streams of an activation tensor ([nb][rows_b][K] fp32) -> the same dense rows of its planes mirror
// (blocked = 0: row-major planes [plane][rows][K], the layout of voc_conv_kernel)
int launch_to_planes_act(const float* src, int nb, long rows_b, long row0, int T, int K, unsigned short* dst, long pstride, int mode, int silu, hipStream_t st, int blocked = 1);
int make_weight_planes(const float* dW, int N, int K, float max_abs, int mode, unsigned short* dst, float* inv, hipStream_t st);
// gemm_f16w.hip: fp16 weights on the f16 matrix pipes (fp32 activations split hi + lo), plain linear layers of the AR chain
bool f16w_gemm_supported(const ConvGemm& g);
bool f16w_gemm_validated_compiler();      // built with the compiler the kernel's workarounds were validated with
int launch_f16w_gemm(const ConvGemm& g, hipStream_t st);
// gemm_stream.hip: weight-streaming f32-MFMA kernel for few rows -- every operand fragment of a wave requested up front, whole K per workgroup.
// mt in {1, 2, 4} x nt in {1, 2} 16-row / 16-column tiles per workgroup, kw in {4, 8, 16} K-split waves; wmode bit 0 = non-temporal weight loads,
// bit 1 = Wsrc is the fragment-major packing of g.W ([N / 16][K / 16][64][4]); probe != 0: timing diagnostics
bool stream_gemm_supported(const ConvGemm& g);
int launch_stream_gemm(const ConvGemm& g, const float* Wsrc, int mt, int nt, int kw, int wmode, int probe, hipStream_t st);
int launch_conv_gemm(const ConvGemm& g, hipStream_t st);
int conv_gemm_last_kind();          // kernel family the calling thread's latest launch_conv_gemm[_group] picked: 0 small-M, 1 tiled, 2 pipelined, 6 weight-streaming (f32 MFMA), 4 split-bf16, 5 fp16 weights (f16 MFMA), 7 / 8 planes H3 / H1, 9 / 10 their LDS-DMA form
int launch_conv_gemm_group(const ConvGemm* gs, int n, hipStream_t st);
// true when launch_conv_gemm would route this (M, N) problem to the K-split small-M kernel, which can normalise its A rows
bool conv_gemm_can_fuse_rms(int M, int N);
int launch_conv_gemm_choice(const ConvGemm& g, hipStream_t st, int kind, int a, int b, int c);   // unit-test hook
int launch_conv_gemm_choice_z(const ConvGemm& g, hipStream_t st, int a, int b, int c, int z);        // small-M kernel with a grid-level K split

}  // namespace sva
