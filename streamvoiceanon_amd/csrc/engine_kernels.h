// Bookkeeping kernels of the per-chunk step (csrc/engine_kernels.hip), launched from engine.hip.
#pragma once
#include <hip/hip_runtime.h>

struct ReprefillArgs {
    int n;
    int slot[128], Rt[128], nf[128], na[128], row_off[129];
};

__global__ void ar_prepare_step_kernel(const float* cached_audio_emb, const float* content_emb, const long long* codes, int T2, int code_off, const int* last_pos, int D, float* x, int* slot, int* pos, int* step_content, int chunk, int ci);
__global__ void copy_rows_kernel(const float* src, long src_stride, long src_off, float* dst, int D);
__global__ void copy_rows2_kernel(const float* src, long src_stride, long src_off, float* dst1, float* dst2, int D);
__global__ void apply_forced_kernel(const int* raw, const int* forced, const int* use_forced, int chunk, int ci, int cb, int ncb, int* tok, int B);
__global__ void ar_finish_frame_kernel(const int* tok, int ncb, int* last_pos, int* nframes, int* pred_hist, int hist_cap, int* step_audio, int chunk, int ci, const long long* codes, int T2, int code_off, int* content_hist, int* ncontent, int B, int last_pos_inc);
__global__ void append_content_kernel(const long long* codes, int T2, int chunk, int* content_hist, int hist_cap, int* ncontent, int* step_content, int B, int* step_counter);
__global__ void broadcast_row_kernel(float* p, long bstride, int src_row, int lo, int hi, int C);
__global__ void fill_rows_kernel(float* p, long bstride, int C, const float* src);
__global__ void inc_kernel(int* p, int v);
__global__ void build_prompt_kernel(const float* spk, int nspk, const float* content_emb, const float* codebook_emb, const float* wait4start, const int* cc, const int* ac, int Pmax, int R, int d, int ncb, int cbsize, int D, float* x);
__global__ void build_delayfill_kernel(const float* content_emb, const int* content_hist, int hist_cap, const int* ncontent, const float* cached_ref_emb, int max_delay, const int* last_pos, int d, int D, float* x, int* slot, int* pos, float* cached_audio_emb, const int* slot_list);
__global__ void build_reprefill_kernel(const ReprefillArgs a, const float* content_emb, const float* codebook_emb, const int* content_hist, const int* pred_hist, int hist_cap, int ncontent, const int* ref_tail, int max_delay, int d, int ncb, int cbsize, int D, int nspk, float* x, int* slot_out, int* pos_out);
__global__ void finish_reprefill_kernel(const ReprefillArgs a, const float* codebook_emb, const int* pred_hist, int hist_cap, const int* ref_tail, int max_delay, int d, int ncb, int cbsize, int D, int nspk, float* cached_ref_emb, int* last_pos);
__global__ void add_list_kernel(int* p, const int* list, int n, int v);
__global__ void add_vec_kernel(int* p, int n, int v);
