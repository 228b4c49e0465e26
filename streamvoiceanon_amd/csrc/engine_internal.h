// Helpers shared by the engine's translation units (packer.hip: engine lifecycle + weight packing; engine_kernels.hip: bookkeeping
// kernels; engine.hip: batches and the per-chunk step).
#pragma once
#include "engine.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace sva {

#define SVA_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc) return _rc;     \
    } while (0)

template <typename T>
inline int dev_alloc(DevPool& pool, T** out, size_t n, bool zero = true) {
    if (n == 0) n = 1;
    const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (bytes > pool.left) {
        const size_t want = pool.chunk_bytes;
        const size_t sz = bytes > want ? bytes : want;
        void* c = nullptr;
        SVA_HIP(hipMalloc(&c, sz));
        pool.chunks.push_back(c);
        pool.cur = (char*)c;
        pool.left = sz;
    }
    void* p = pool.cur;
    pool.cur += bytes;
    pool.left -= bytes;
    if (zero) SVA_HIP(hipMemset(p, 0, n * sizeof(T)));
    *out = (T*)p;
    return 0;
}
inline int upload(DevPool& pool, float** out, const std::vector<float>& v) {
    SVA_TRY(dev_alloc(pool, out, v.size(), false));
    SVA_HIP(hipMemcpy(*out, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
inline int alloc_act(DevPool& pool, Act& a, int B, int H, long Tmax, int C) {
    a.H = H;
    a.C = C;
    a.rows = H + Tmax;
    a.bstride = a.rows * C;
    return dev_alloc(pool, &a.p, (size_t)B * a.bstride, true);
}

// stages.hip: launch sequences of the three stages (enqueue on b->stream only)
constexpr int kResK[3] = {3, 7, 11};     // HiFiGAN ResBlock kernel sizes / dilations (firefly.py:149-219)
constexpr int kResD[3] = {1, 3, 5};
const DebugOptions& debug_options();
int gemm_call(sva_batch* b, const float* A, long a_bstride, long a_off, int lda, int nb, int T, int stride, int dil, int taps, int Cin, const Lin& w, float* C, long c_bstride, long c_off, int ldc, ConvGemm proto = ConvGemm());
hipEvent_t next_event(sva_batch* b);
int stream_fork(sva_batch* b, hipStream_t from, hipStream_t to);
int conv_act(sva_batch* b, const Act& in, int T_out, int stride, int dil, int taps, const Lin& w, Act& out, ConvGemm proto = ConvGemm());
int conv_desc(sva_batch* b, const Act& in, int T_out, int dil, int taps, const Lin& w, Act& out, ConvGemm& g);
int gemm_group_call(sva_batch* b, const ConvGemm* gs, int n);
bool planes_edge(const Lin& producer, const Lin& consumer, long rows, int streams);
int cnx_block_t(sva_batch* b, const CNX& c, Act& x, int T, float* h1, long h1_bs, float* h2, long h2_bs, Act* out = nullptr, int skip_lo = 0, int skip_hi = 0);
int cnx_block(sva_batch* b, const CNX& c, Act& x, int T, float* h1, float* h2, Act* out = nullptr);
int enc_frontend_window(sva_batch* b, const int* step_ptr, int n_chunk, int add, int Tm, const EncFront* front = nullptr, Act* tokens_out = nullptr);
int enc_frontend_stream(sva_batch* b, const int* step_ptr, int n_chunk, int add, int part = 0);
int enc_frontend_merged(sva_batch* b, const int* step_ptr, int n_chunk, int add, int part = 0, int fpar = 0);
int enc_transformer(sva_batch* b, const Act& xin, int need_rows, int part = 0);
int encode(sva_batch* b, const int* step_ptr, int n_chunk, int add);
int encode_incremental(sva_batch* b, const int* step_ptr, int n_chunk, int add, bool transformer_too = true);
int ar_layers_pass(sva_batch* b, std::vector<TrLayer>& layers, int M, const int* d_slot, const int* d_pos, const float* rope, float* kv, long kv_layer, long kv_slot, int S, float* x, int run_slot = -1, int run_pos0 = 0);
int ar_decode_frame(sva_batch* b, int ci);
// orders the persistent decode launches of different batches of one engine (stages.hip)
struct PersistentChain {
    sva_batch* b;
    hipStream_t st;
    bool active;
    int rc = 0;
    std::unique_lock<std::mutex> lk;
    PersistentChain(sva_batch* b, hipStream_t st, bool active);
    int finish();          // records the engine's event behind the launches enqueued since construction
};
int ar_decode_frame_batch(sva_batch* b, int ci);
int ar_decode_frame_mega(sva_batch* b, int ci, const long long* codes, int code_off);
int ar_frame_tail(sva_batch* b, int ci, long hid_stride, long hid_off, const long long* codes, int codes_ld, int code_off, int last_pos_inc);
int ar_prefill_slot(sva_batch* b, int slot, int R, bool tap_logits = false);
int ar_delay_fill(sva_batch* b, const std::vector<int>& slots);
int ar_delay_fill(sva_batch* b);
bool voc_level_is_fused(const sva_batch* b, int C);
int vocode(sva_batch* b, int T, bool shift, int part = 0);
int register_shift(sva_batch* b, Act& a, int rows_per_frame);

}  // namespace sva
