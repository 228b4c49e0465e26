// Bookkeeping kernels of the per-chunk step: frame preparation / completion of the multi-launch AR decode, content and prediction
// history rings, prompt / delay-fill / re-prefill row builders (modules/dual_ar_stream.py:764-837, evaluations/infer_arvc.py:547-564).
#include "engine_kernels.h"

// small device helpers that live here because they touch the batch control block -------------------------
__global__ void ar_prepare_step_kernel(const float* __restrict__ cached_audio_emb, const float* __restrict__ content_emb,
                                       const long long* __restrict__ codes, int T2, int code_off, const int* __restrict__ last_pos,
                                       int D, float* __restrict__ x, int* __restrict__ slot, int* __restrict__ pos,
                                       int* __restrict__ step_content, int chunk, int ci) {
    // decode_one (dual_ar_stream.py:817-837): tokens [cached_new_audio_emb, src_cond] at (last+1, last+2)
    const int b = blockIdx.x;
    const int code = (int)codes[(long)b * T2 + code_off];
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        x[((long)b * 2) * D + i] = cached_audio_emb[(long)b * D + i];
        x[((long)b * 2 + 1) * D + i] = content_emb[(long)code * D + i];
    }
    if (threadIdx.x == 0) {
        slot[2 * b] = b; slot[2 * b + 1] = b;
        pos[2 * b] = last_pos[b] + 1; pos[2 * b + 1] = last_pos[b] + 2;
        step_content[b * chunk + ci] = code;
    }
}

__global__ void copy_rows_kernel(const float* __restrict__ src, long src_stride, long src_off, float* __restrict__ dst, int D) {
    const int r = blockIdx.x;
    for (int i = threadIdx.x; i < D; i += blockDim.x) dst[(long)r * D + i] = src[(long)r * src_stride + src_off + i];
}

__global__ void copy_rows2_kernel(const float* __restrict__ src, long src_stride, long src_off, float* __restrict__ dst1,
                                  float* __restrict__ dst2, int D) {
    const int r = blockIdx.x;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float v = src[(long)r * src_stride + src_off + i];
        dst1[(long)r * D + i] = v;
        dst2[(long)r * D + i] = v;
    }
}

__global__ void apply_forced_kernel(const int* __restrict__ raw, const int* __restrict__ forced, const int* __restrict__ use_forced,
                                    int chunk, int ci, int cb, int ncb, int* __restrict__ tok, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int t = raw[b * ncb + cb];
    if (*use_forced) t = forced[((long)b * ncb + cb) * chunk + ci];
    tok[b * ncb + cb] = t;
}

__global__ void ar_finish_frame_kernel(const int* __restrict__ tok, int ncb, int* __restrict__ last_pos, int* __restrict__ nframes,
                                       int* __restrict__ pred_hist, int hist_cap, int* __restrict__ step_audio, int chunk, int ci,
                                       const long long* __restrict__ codes, int T2, int code_off, int* __restrict__ content_hist,
                                       int* __restrict__ ncontent, int B, int last_pos_inc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int f = nframes[b];
    for (int i = 0; i < ncb; ++i) {
        const int t = tok[b * ncb + i];
        pred_hist[((long)b * ncb + i) * hist_cap + (f & (hist_cap - 1))] = t;     // ring (hist_cap is a power of two)
        step_audio[((long)b * ncb + i) * chunk + ci] = t;
    }
    nframes[b] = f + 1;
    last_pos[b] += last_pos_inc;
}

__global__ void append_content_kernel(const long long* __restrict__ codes, int T2, int chunk, int* __restrict__ content_hist,
                                      int hist_cap, int* __restrict__ ncontent, int* __restrict__ step_content, int B,
                                      int* __restrict__ step_counter) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && step_counter) *step_counter += 1;          // the chunk counter (ring position) advances with the step
    if (b >= B) return;
    const int n = ncontent[b];
    for (int i = 0; i < chunk; ++i) {
        const int code = (int)codes[(long)b * T2 + T2 - chunk + i];
        content_hist[(long)b * hist_cap + ((n + i) & (hist_cap - 1))] = code;
        step_content[b * chunk + i] = code;
    }
    ncontent[b] = n + chunk;
}

// dst rows [lo, hi) of every batch item <- row `src_row`
__global__ void broadcast_row_kernel(float* p, long bstride, int src_row, int lo, int hi, int C) {
    float* base = p + (long)blockIdx.y * bstride;
    const int r = lo + blockIdx.x;
    if (r >= hi || r == src_row) return;
    for (int i = threadIdx.x; i < C; i += blockDim.x) base[(long)r * C + i] = base[(long)src_row * C + i];
}

// rows [0, gridDim.x) of every batch item <- one source row
__global__ void fill_rows_kernel(float* p, long bstride, int C, const float* __restrict__ src) {
    float* dst = p + (long)blockIdx.y * bstride + (long)blockIdx.x * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) dst[i] = src[i];
}

__global__ void inc_kernel(int* p, int v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += v;
}

// prompt sequence builder (DualARWrapper.prefill_prompt, dual_ar_stream.py:764-796, layout in SURVEY.md A.1):
//   rows 0..32 = speaker prefix; then for i < R: row 33+2i = content_emb[cc[i]],
//   row 34+2i = (i < d ? wait4start[i] : audio_embed(ac[:, i-d]))
__global__ void build_prompt_kernel(const float* __restrict__ spk, int nspk, const float* __restrict__ content_emb,
                                    const float* __restrict__ codebook_emb, const float* __restrict__ wait4start,
                                    const int* __restrict__ cc, const int* __restrict__ ac, int Pmax, int R, int d, int ncb,
                                    int cbsize, int D, float* __restrict__ x) {
    const int r = blockIdx.x;
    float* o = x + (long)r * D;
    if (r < nspk) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) o[i] = spk[(long)r * D + i];
        return;
    }
    const int i = (r - nspk) >> 1;
    if (((r - nspk) & 1) == 0) {
        const float* s = content_emb + (long)cc[i] * D;
        for (int k = threadIdx.x; k < D; k += blockDim.x) o[k] = s[k];
    } else if (i < d) {
        const float* s = wait4start + (long)i * D;
        for (int k = threadIdx.x; k < D; k += blockDim.x) o[k] = s[k];
    } else {
        for (int k = threadIdx.x; k < D; k += blockDim.x) {
            float acc = 0.f;
            for (int q = 0; q < ncb; ++q) acc += codebook_emb[((long)ac[(long)q * Pmax + i - d] + (long)q * cbsize) * D + k];
            o[k] = acc;
        }
    }
}

// delay fill (prefill_src_condition4delay, dual_ar_stream.py:798-815): per slot the interleave
// [c_0, r_0, c_1, r_1, ..., c_{d-1}] (2d-1 rows; the dropped last row r_{d-1} becomes cached_new_audio_emb)
__global__ void build_delayfill_kernel(const float* __restrict__ content_emb, const int* __restrict__ content_hist, int hist_cap,
                                       const int* __restrict__ ncontent, const float* __restrict__ cached_ref_emb, int max_delay,
                                       const int* __restrict__ last_pos, int d, int D, float* __restrict__ x, int* __restrict__ slot,
                                       int* __restrict__ pos, float* __restrict__ cached_audio_emb, const int* __restrict__ slot_list) {
    const int rows = 2 * d - 1;
    const int li = blockIdx.x / rows, r = blockIdx.x % rows;
    const int b = slot_list[li];
    const int i = r >> 1;
    float* o = x + ((long)li * rows + r) * D;
    if ((r & 1) == 0) {
        const int code = content_hist[(long)b * hist_cap + ((ncontent[b] - d + i) & (hist_cap - 1))];
        for (int k = threadIdx.x; k < D; k += blockDim.x) o[k] = content_emb[(long)code * D + k];
    } else {
        for (int k = threadIdx.x; k < D; k += blockDim.x) o[k] = cached_ref_emb[((long)b * max_delay + i) * D + k];
    }
    if (r == 0)
        for (int k = threadIdx.x; k < D; k += blockDim.x)
            cached_audio_emb[(long)b * D + k] = cached_ref_emb[((long)b * max_delay + d - 1) * D + k];
    if (threadIdx.x == 0) {
        slot[li * rows + r] = b;
        pos[li * rows + r] = last_pos[b] + 1 + r;
    }
}
// Re-prefill of the due slots of a batch in ONE pass (infer_arvc.py:547-564: prompt <- [ref (truncated), last buffer_frames predicted
// frames] / [ref content, src content[-buffer-d:-d]]).  The attention is causal, so the K / V rows of the reference part of that prompt
// -- positions 0 .. 32 + 2 Rt -- are the ones the slot's cache has held since its first prefill: only the 2 na rows of the appended
// frames are new.  They are built here from the device-resident history rings (no host round trip) for every due slot and then run
// through the layers as one (sum of rows)-row pass against the cached prefix.  Row layout per slot, i = Rt .. Rt + na - 1:
//   position 33 + 2 i = content_emb[content_hist[c_lo + i - Rt]],  34 + 2 i = audio_embed(ac'[:, i - d]),
//   ac'[:, j] = ref_audio[:, j] for j < Rt (the last d reference frames: ref_tail) and pred_hist[nf - na + j - Rt] beyond.
__global__ void build_reprefill_kernel(const ReprefillArgs a, const float* __restrict__ content_emb, const float* __restrict__ codebook_emb,
                                       const int* __restrict__ content_hist, const int* __restrict__ pred_hist, int hist_cap, int ncontent,
                                       const int* __restrict__ ref_tail, int max_delay, int d, int ncb, int cbsize, int D, int nspk,
                                       float* __restrict__ x, int* __restrict__ slot_out, int* __restrict__ pos_out) {
    const int row = blockIdx.x;
    int li = 0;
    while (li + 1 < a.n && row >= a.row_off[li + 1]) ++li;
    const int r = row - a.row_off[li], b = a.slot[li], Rt = a.Rt[li], nf = a.nf[li], na = a.na[li];
    const int i = Rt + (r >> 1);
    float* o = x + (long)row * D;
    const int mask = hist_cap - 1;
    if ((r & 1) == 0) {
        const int c_lo = ncontent - d - na;                      // src_content_codes[-buffer-d:-d]
        const int code = content_hist[(long)b * hist_cap + ((c_lo + (i - Rt)) & mask)];
        for (int k = threadIdx.x; k < D; k += blockDim.x) o[k] = content_emb[(long)code * D + k];
    } else {
        const int j = i - d;                                     // frame of the concatenated audio codes
        int code[8];
        for (int q = 0; q < ncb; ++q)
            code[q] = j < Rt ? ref_tail[((long)b * ncb + q) * max_delay + (max_delay - (Rt - j))]
                             : pred_hist[((long)b * ncb + q) * hist_cap + ((nf - na + (j - Rt)) & mask)];
        for (int k = threadIdx.x; k < D; k += blockDim.x) {
            float acc = 0.f;
            for (int q = 0; q < ncb; ++q) acc += codebook_emb[((long)code[q] + (long)q * cbsize) * D + k];      // codebooks summed in order (build_prompt_kernel)
            o[k] = acc;
        }
    }
    if (threadIdx.x == 0) {
        slot_out[row] = b;
        pos_out[row] = nspk + 2 * Rt + r;
    }
}
// cached_ref_emb = embed(ac')[-d:] of the new prompt (dual_ar_stream.py:775) and last_pos = its last position, per due slot
__global__ void finish_reprefill_kernel(const ReprefillArgs a, const float* __restrict__ codebook_emb, const int* __restrict__ pred_hist, int hist_cap,
                                        const int* __restrict__ ref_tail, int max_delay, int d, int ncb, int cbsize, int D, int nspk,
                                        float* __restrict__ cached_ref_emb, int* __restrict__ last_pos) {
    const int li = blockIdx.x / d, jj = blockIdx.x % d;
    const int b = a.slot[li], Rt = a.Rt[li], nf = a.nf[li], na = a.na[li];
    const int j = Rt + na - d + jj;
    const int mask = hist_cap - 1;
    int code[8];
    for (int q = 0; q < ncb; ++q)
        code[q] = j < Rt ? ref_tail[((long)b * ncb + q) * max_delay + (max_delay - (Rt - j))]
                         : pred_hist[((long)b * ncb + q) * hist_cap + ((nf - na + (j - Rt)) & mask)];
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        float acc = 0.f;
        for (int q = 0; q < ncb; ++q) acc += codebook_emb[((long)code[q] + (long)q * cbsize) * D + k];
        cached_ref_emb[((long)b * max_delay + jj) * D + k] = acc;
    }
    if (jj == 0 && threadIdx.x == 0) last_pos[b] = nspk + 2 * (Rt + na) - 1;
}

__global__ void add_list_kernel(int* p, const int* list, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[list[i]] += v;
}
__global__ void add_vec_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += v;
}

