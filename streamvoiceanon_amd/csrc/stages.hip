// Stage launch sequences of the sva engine: the conv-GEMM call helpers, the content encoder's passes (window / streaming / merged
// incremental front-end, window transformer + BSQ: modules/vqgan/modules/firefly.py:375-520, windowed_transformer.py, bsq.py), the
// dual AR's layer pass, frame decode, prompt prefill and delay fill (modules/dual_ar_stream.py:764-837, 1168-1219) and the streaming
// vocoder (firefly.py:243-293).  Everything here only enqueues kernels on b->stream; engine.hip orchestrates the streams and graphs.
#include "engine_internal.h"
#include "engine_kernels.h"

#include <map>
#include <mutex>

using namespace sva;

namespace sva {
int gemm_call(sva_batch* b, const float* A, long a_bstride, long a_off, int lda, int nb, int T, int stride, int dil,
              int taps, int Cin, const Lin& w, float* C, long c_bstride, long c_off, int ldc, ConvGemm proto) {
    ConvGemm g = proto;
    g.A = A; g.a_bstride = a_bstride; g.a_off = a_off; g.lda = lda;
    g.T = T; g.M = nb * T; g.stride = stride; g.dil = dil; g.taps = taps; g.Cin = Cin;
    g.W = w.W; g.Wk = w.Wk; g.Wh = w.Wh; g.N = w.N; g.bias = w.b;
    g.Wp = w.Wp; g.wp_pstride = (long)w.N * w.K; g.wp_inv = w.wp_inv; g.pmode = w.pmode; g.ovf = b->d_mm_ovf;
    g.cu_limit = b->enc_cus;        // (the encoder / vocoder streams' CU mask when the batch runs its AR chain on a partition of its own)
    g.C = C; g.c_bstride = c_bstride; g.c_off = c_off; g.ldc = ldc;
    if (g.Ap) g.A = nullptr;        // operand planes replace the fp32 tensor (cnx_block_t / enc_transformer hand-overs)
    if (g.Cp) g.C = nullptr;
    SVA_CHECK(w.K == taps * Cin, "gemm_call: weight K mismatch");
    b->gemm_flops += 2.0 * g.M * (double)g.N * w.K;
    b->gemm_launches += 1;
    // algorithmic bytes: every operand element once (input rows incl. the tap halo, weights, outputs, residual)
    b->gemm_bytes += 4.0 * ((double)nb * ((double)(T - 1) * stride + (taps - 1) * dil + 1) * Cin + (double)g.N * w.K +
                            (double)g.M * (g.w13 ? g.N / 2 : g.N) * (g.res ? 2 : 1));
    if (b->prof_on) {        // bench.py roofline leg: bracket every conv-GEMM launch with hipEvents on the launch stream
        if (b->prof_n + 2 > (int)b->prof_ev.size()) {
            const size_t old = b->prof_ev.size();
            b->prof_ev.resize(old + 512);
            for (size_t i = old; i < b->prof_ev.size(); ++i) SVA_HIP(hipEventCreate(&b->prof_ev[i]));
        }
        b->prof_shapes.push_back({g.M, g.N, w.K, taps, g.w13 * 8 + g.a_silu * 4 + (g.res ? 2 : 0) + (g.act == ACT_GELU ? 1 : 0)});
        SVA_HIP(hipEventRecord(b->prof_ev[b->prof_n], b->stream));
        int rc = launch_conv_gemm(g, b->stream);
        SVA_HIP(hipEventRecord(b->prof_ev[b->prof_n + 1], b->stream));
        b->prof_shapes.back()[4] += 256 * (conv_gemm_last_kind() + 1);        // kernel family of the launch, for the per-pipe roofline
        b->prof_n += 2;
        return rc;
    }
    return launch_conv_gemm(g, b->stream);
}

// fork / join of independent sub-chains on side streams (captured into the hipGraph as parallel branches)
hipEvent_t next_event(sva_batch* b) { return b->evpool[(b->evi++) & 63]; }
int stream_fork(sva_batch* b, hipStream_t from, hipStream_t to) {
    hipEvent_t ev = next_event(b);
    SVA_HIP(hipEventRecord(ev, from));
    SVA_HIP(hipStreamWaitEvent(to, ev, 0));
    return 0;
}

// causal conv (FishConvNet) of `in` (history rows in front) into rows [out.H, out.H+T) of `out`
int conv_act(sva_batch* b, const Act& in, int T_out, int stride, int dil, int taps, const Lin& w, Act& out,
             ConvGemm proto) {
    const int padL = (taps - 1) * dil + 1 - stride;
    SVA_CHECK(in.H >= padL, "conv_act: not enough history rows");
    return gemm_call(b, in.p, in.bstride, (long)(in.H - padL) * in.C, in.C, b->B, T_out, stride, dil, taps, in.C, w, out.p,
                     out.bstride, (long)out.H * out.C, out.C, proto);
}

// descriptor of the same causal conv without launching it (grouped launches)
int conv_desc(sva_batch* b, const Act& in, int T_out, int dil, int taps, const Lin& w, Act& out, ConvGemm& g) {
    const int padL = (taps - 1) * dil;
    SVA_CHECK(in.H >= padL, "conv_desc: not enough history rows");
    SVA_CHECK(w.K == taps * in.C, "conv_desc: weight K mismatch");
    g.A = in.p; g.a_bstride = in.bstride; g.a_off = (long)(in.H - padL) * in.C; g.lda = in.C;
    g.T = T_out; g.M = b->B * T_out; g.stride = 1; g.dil = dil; g.taps = taps; g.Cin = in.C;
    g.W = w.W; g.Wk = w.Wk; g.N = w.N; g.bias = w.b;
    g.Wp = w.Wp; g.wp_pstride = (long)w.N * w.K; g.wp_inv = w.wp_inv; g.pmode = w.pmode; g.ovf = b->d_mm_ovf;
    g.C = out.p; g.c_bstride = out.bstride; g.c_off = (long)out.H * out.C; g.ldc = out.C;
    return 0;
}
// n <= 3 same-shape problems in one launch (bookkeeping as gemm_call: one "launch", summed FLOPs)
int gemm_group_call(sva_batch* b, const ConvGemm* gs, int n) {
    double fl = 0;
    int kmax = 0, tmax = 0;
    for (int i = 0; i < n; ++i) {
        fl += 2.0 * gs[i].M * (double)gs[i].N * gs[i].taps * gs[i].Cin;
        if (gs[i].taps * gs[i].Cin > kmax) { kmax = gs[i].taps * gs[i].Cin; tmax = gs[i].taps; }
    }
    b->gemm_flops += fl;
    b->gemm_launches += 1;
    for (int i = 0; i < n; ++i)
        b->gemm_bytes += 4.0 * ((double)(gs[i].M / gs[i].T) * ((double)(gs[i].T - 1) + (gs[i].taps - 1) * gs[i].dil + 1) * gs[i].Cin +
                                (double)gs[i].N * gs[i].taps * gs[i].Cin + (double)gs[i].M * gs[i].N * (gs[i].res ? 2 : 1));
    if (b->prof_on) {
        if (b->prof_n + 2 > (int)b->prof_ev.size()) {
            const size_t old = b->prof_ev.size();
            b->prof_ev.resize(old + 512);
            for (size_t i = old; i < b->prof_ev.size(); ++i) SVA_HIP(hipEventCreate(&b->prof_ev[i]));
        }
        // table row: the group as one problem with the summed K (its FLOPs = 2 M N sum K)
        int ksum = 0;
        for (int i = 0; i < n; ++i) ksum += gs[i].taps * gs[i].Cin;
        b->prof_shapes.push_back({gs[0].M, gs[0].N, ksum, tmax, 16 + gs[0].a_silu * 4 + (gs[0].res ? 2 : 0)});
        SVA_HIP(hipEventRecord(b->prof_ev[b->prof_n], b->stream));
        int rc = launch_conv_gemm_group(gs, n, b->stream);
        SVA_HIP(hipEventRecord(b->prof_ev[b->prof_n + 1], b->stream));
        b->prof_shapes.back()[4] += 256 * (conv_gemm_last_kind() + 1);
        b->prof_n += 2;
        return rc;
    }
    return launch_conv_gemm_group(gs, n, b->stream);
}

// one voc_conv_kernel launch (three branches of one conv stage of a narrow level), bookkeeping as gemm_group_call
int voc_conv_call(sva_batch* b, const VocConvGroup& gg, int C, int pmode) {
    double fl = 0, by = 0;
    int ksum = 0;
    for (int i = 0; i < gg.n; ++i) {
        const VocConv& g = gg.g[i];
        fl += 2.0 * gg.B * (double)gg.T * C * g.taps * C;
        by += 4.0 * ((double)gg.B * ((double)gg.T + (g.taps - 1) * g.dil) * C + (double)C * g.taps * C + (double)gg.B * gg.T * C * (g.res ? 2 : 1));
        ksum += g.taps * C;
    }
    b->gemm_flops += fl;
    b->gemm_launches += 1;
    b->gemm_bytes += by;
    if (b->prof_on) {
        if (b->prof_n + 2 > (int)b->prof_ev.size()) {
            const size_t old = b->prof_ev.size();
            b->prof_ev.resize(old + 512);
            for (size_t i = old; i < b->prof_ev.size(); ++i) SVA_HIP(hipEventCreate(&b->prof_ev[i]));
        }
        b->prof_shapes.push_back({gg.B * gg.T, C, ksum, 11, 16 + (gg.g[0].res ? 2 : 0) + 256 * ((pmode == PLANES_H3 ? 9 : 10) + 1)});
        SVA_HIP(hipEventRecord(b->prof_ev[b->prof_n], b->stream));
        int rc = launch_voc_conv(gg, C, pmode, b->enc_cus, b->stream);
        SVA_HIP(hipEventRecord(b->prof_ev[b->prof_n + 1], b->stream));
        b->prof_n += 2;
        return rc;
    }
    return launch_voc_conv(gg, C, pmode, b->enc_cus, b->stream);
}

__global__ void mean3_kernel(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2, long y_bstride,
                             float* __restrict__ out, long o_bstride, long o_off, long n4) {
    // ParallelBlock: torch.stack([...]).mean(0) (firefly.py:214-215) of the three branch outputs, float4 lanes
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 a = reinterpret_cast<const float4*>(y0 + (long)b * y_bstride)[i];
    const float4 c = reinterpret_cast<const float4*>(y1 + (long)b * y_bstride)[i];
    const float4 d = reinterpret_cast<const float4*>(y2 + (long)b * y_bstride)[i];
    float4 r;
    r.x = ((a.x + c.x) + d.x) / 3.0f; r.y = ((a.y + c.y) + d.y) / 3.0f; r.z = ((a.z + c.z) + d.z) / 3.0f; r.w = ((a.w + c.w) + d.w) / 3.0f;
    reinterpret_cast<float4*>(out + (long)b * o_bstride + o_off)[i] = r;
}

// Two chained GEMMs of `rows` rows can hand their intermediate tensor over as operand planes (gemm_planes.hip) when both weights carry
// planes of one fp16 format -- two fp16 planes (or one) fill exactly the bytes (half the bytes) of the fp32 tensor they replace, so
// they live in its buffer -- and the problem is at the scale where the planes kernel is the dispatcher's choice anyway.
// (rows: the GEMM pair's own rows.  Round 5, after the LDS-DMA forms: from 10 streams on (12 / 16 / 20 streams +6 / +5 / +9 % frames/s, 10 streams +1.4 % and the
// synchronous step 4.5 % shorter, 8 streams slower: profiles/r05_planes_min_streams.txt).  From that stream count on -- where the main passes run on planes anyway -- the shorter passes of the same step, the
// quantizer downsampler's 88 / 47 rows per stream, go the same way: 3008 x 2048 x 512 at 64 streams 72 -> 28 us, profiles/r05_planes_tune.log)
// (streams < 0: a batch whose encoder runs on a CU partition beside a multi-launch AR chain -- fp16 AR at 12-32 streams --: there the kernels tuned on the
// partition keep the better THROUGHPUT below 24 streams, 4745 against 4390 frames/s at 16, although the synchronous step is 15 % longer: r05_planes_min_streams.txt)
static bool planes_rows_ok(long rows, int streams) {
    const int ms = streams < 0 ? std::max(24, debug_options().planes_min_streams) : debug_options().planes_min_streams;
    const int n = streams < 0 ? -streams : streams;
    return rows >= 128L * ms || (n >= ms && rows >= 1024);
}
static int planes_streams(const sva_batch* b) { return b->enc_cus > 0 ? -b->B : b->B; }
bool planes_edge(const Lin& producer, const Lin& consumer, long rows, int streams) {
    return planes_rows_ok(rows, streams) && producer.Wp && consumer.Wp && producer.pmode == consumer.pmode && (producer.pmode == PLANES_H3 || producer.pmode == PLANES_H1) &&
           consumer.K % 32 == 0;
}
// A non-GEMM producer (dwconv7 + LayerNorm, RMSNorm rows) can write its output as the planes its consumer GEMM takes -- same scale rule;
// with BOTH operands as planes the GEMM runs in its persistent LDS-DMA form (gemm_planes.hip, planes_dma_kernel)
bool planes_input(const Lin& consumer, long rows, int streams) {
    return planes_rows_ok(rows, streams) && consumer.Wp && (consumer.pmode == PLANES_H3 || consumer.pmode == PLANES_H1) && consumer.K % 32 == 0 && debug_options().planes_dma != 0;
}

// ConvNeXtBlock (firefly.py:421-440) on x rows [x.H, x.H+T); result into `out` rows [out.H, out.H+T)
// (out == nullptr: in place).  Streaming users must NOT run in place: the dwconv history of the next step is the
// block INPUT, while downstream convs need history of the block OUTPUT.  h1 / h2 = scratch with batch strides.
int cnx_block_t(sva_batch* b, const CNX& c, Act& x, int T, float* h1, long h1_bs, float* h2, long h2_bs, Act* out, int skip_lo,
                int skip_hi) {
    const int C = c.C;
    SVA_CHECK(x.H >= 6 && x.C == C, "cnx_block: bad activation");
    Act& o = out ? *out : x;
    SVA_CHECK(o.C == C, "cnx_block: bad output activation");
    ConvGemm p1;
    p1.act = ACT_GELU;
    ConvGemm p2;
    // batch scale: the hidden tensor goes from pwconv1's GELU epilogue to pwconv2 as operand planes, in the scratch buffer's own memory
    const long rows = (long)b->B * T;
    if (planes_edge(c.pw1, c.pw2, rows, planes_streams(b))) {
        SVA_CHECK(h2_bs == (long)T * 4 * C, "cnx_block: planes hand-over needs dense hidden rows");
        p1.Cp = reinterpret_cast<unsigned short*>(h2); p1.cp_pstride = h2_bs * b->B; p1.cp_rows = rows;
        p2.Ap = p1.Cp; p2.ap_pstride = p1.cp_pstride; p2.ap_rows = rows;
    }
    if (b->B * T <= 16 && C <= 512) {
        // a handful of rows (streaming pass, upsampler at small B): depthwise conv + LayerNorm happen in the prologue of the
        // pointwise GEMM (every column block recomputes them -- a few thousand FMAs -- instead of a launch of their own)
        p1.dw_wT = c.dwT; p1.dw_b = c.dwb; p1.ln_w = c.lnw; p1.ln_b = c.lnb; p1.ln_eps = 1e-6f;
        SVA_TRY(gemm_call(b, x.p, x.bstride, (long)(x.H - 6) * C, C, b->B, T, 1, 1, 1, C, c.pw1, h2, h2_bs, 0, 4 * C, p1));
    } else {
        if (planes_input(c.pw1, rows, planes_streams(b)) && h1_bs == (long)T * C) {        // LayerNorm output straight into operand planes, in h1's own memory
            p1.Ap = reinterpret_cast<unsigned short*>(h1); p1.ap_pstride = h1_bs * b->B; p1.ap_rows = rows;
            SVA_TRY(launch_dwconv7_ln(x.p, x.bstride, (long)(x.H - 6) * C, b->B, T, C, c.dwT, c.dwb, c.lnw, c.lnb, 1e-6f, h1, h1_bs, b->stream,
                                      reinterpret_cast<unsigned short*>(h1), p1.ap_pstride, planes_count(c.pw1.pmode), rows));
        } else {
            SVA_TRY(launch_dwconv7_ln(x.p, x.bstride, (long)(x.H - 6) * C, b->B, T, C, c.dwT, c.dwb, c.lnw, c.lnb, 1e-6f, h1, h1_bs, b->stream));
        }
        SVA_TRY(gemm_call(b, h1, h1_bs, 0, C, b->B, T, 1, 1, 1, C, c.pw1, h2, h2_bs, 0, 4 * C, p1));
    }
    p2.gamma = c.gamma;
    p2.skip_lo = skip_lo; p2.skip_hi = skip_hi;
    p2.res = x.p; p2.r_bstride = x.bstride; p2.r_off = (long)x.H * C; p2.ldr = C;
    SVA_TRY(gemm_call(b, h2, h2_bs, 0, 4 * C, b->B, T, 1, 1, 1, 4 * C, c.pw2, o.p, o.bstride, (long)o.H * C, C, p2));
    return 0;
}
int cnx_block(sva_batch* b, const CNX& c, Act& x, int T, float* h1, float* h2, Act* out) {
    return cnx_block_t(b, c, x, T, h1, (long)T * c.C, h2, (long)T * 4 * c.C, out);
}

// ---- E: content encoder ------------------------------------------------------------------------------
// Conv front-end (mel -> stem -> 18 ConvNeXt -> 2x (conv k2 s2 + ConvNeXt)) on the FIRST `Tm` mel frames of the
// current window, zero left padding exactly as the reference's window pass (causal net: row j depends on rows <= j).
// Tm = T0: the full-window formulation; Tm = head rows: the head pass of the exact-incremental formulation.
int enc_frontend_window(sva_batch* b, const int* step_ptr, int n_chunk, int add, int Tm, const EncFront* front,
                        Act* tokens_out) {
    sva_engine* e = b->e;
    const EncFront& F = front ? *front : e->tokf;
    const sva_config& c = e->cfg;
    const int B = b->B, T0 = b->T0;
    hipStream_t st = b->stream;
    SVA_TRY(launch_stft_mag_ring(b->ring, step_ptr, n_chunk, add, B, b->N, e->twiddle, e->hann, b->mag, 1088, (long)T0 * 1088, 0, Tm, st));
    {   // mel = log(clamp(fb^T mag, 1e-5))  (spectrogram.py:110-115, 124-125)
        ConvGemm p;
        p.act = ACT_LOGCLAMP;
        SVA_TRY(gemm_call(b, b->mag, (long)T0 * 1088, 0, 1088, B, Tm, 1, 1, 1, 1088, e->mel_fb, b->mel.p, b->mel.bstride,
                          (long)b->mel.H * c.n_mels, c.n_mels, p));
    }
    // stem: causal conv k7 + LayerNorm(channels)  (firefly.py:458-468)
    SVA_TRY(gemm_call(b, b->mel.p, b->mel.bstride, 0, c.n_mels, B, Tm, 1, 1, 7, c.n_mels, F.stem, b->h1, (long)Tm * c.enc_dims[0], 0,
                      c.enc_dims[0]));
    SVA_TRY(launch_layernorm_rows(b->h1, (long)Tm * c.enc_dims[0], 0, c.enc_dims[0], B, Tm, c.enc_dims[0], F.stem_lnw, F.stem_lnb,
                                  1e-6f, b->xs[0].p, b->xs[0].bstride, (long)b->xs[0].H * c.enc_dims[0], c.enc_dims[0], st));
    for (int i = 0; i < 4; ++i) {
        const int C = c.enc_dims[i];
        if (i > 0) {   // LayerNorm(channels) + Conv1d k1  (firefly.py:471-476)
            const int Cp = c.enc_dims[i - 1];
            SVA_TRY(launch_layernorm_rows(b->xs[i - 1].p, b->xs[i - 1].bstride, (long)b->xs[i - 1].H * Cp, Cp, B, Tm, Cp, F.trans_lnw[i],
                                          F.trans_lnb[i], 1e-6f, b->h1, (long)Tm * Cp, 0, Cp, st));
            SVA_TRY(gemm_call(b, b->h1, (long)Tm * Cp, 0, Cp, B, Tm, 1, 1, 1, Cp, F.trans[i], b->xs[i].p, b->xs[i].bstride,
                              (long)b->xs[i].H * C, C));
        }
        for (auto& blk : F.stages[i]) SVA_TRY(cnx_block(b, blk, b->xs[i], Tm, b->h1, b->h2));
    }
    const int D = c.tr_dim;
    SVA_TRY(launch_layernorm_rows(b->xs[3].p, b->xs[3].bstride, (long)b->xs[3].H * D, D, B, Tm, D, F.final_lnw, F.final_lnb, 1e-6f,
                                  b->feat.p, b->feat.bstride, 0, D, st));
    // BSQ downsample x2: conv k2 s2 + ConvNeXtBlock  (bsq_no_upsample.py:48-61)
    SVA_TRY(conv_act(b, b->feat, Tm / 2, 2, 1, 2, F.ds_conv[0], b->d1));
    SVA_TRY(cnx_block(b, F.ds_cnx[0], b->d1, Tm / 2, b->h1, b->h2));
    {
        Act in = b->d1;      // read the new rows (no left padding needed: padL = 0)
        in.p = b->d1.p + (long)b->d1.H * D;
        in.H = 0;
        SVA_TRY(conv_act(b, in, Tm / 4, 2, 1, 2, F.ds_conv[1], b->d2));
    }
    if (tokens_out) {            // last block out of place: its rows land in the caller's token buffer (rows [H, H + Tm/4))
        SVA_TRY(cnx_block(b, F.ds_cnx[1], b->d2, Tm / 4, b->h1, b->h2, tokens_out));
    } else {
        SVA_TRY(cnx_block(b, F.ds_cnx[1], b->d2, Tm / 4, b->h1, b->h2));
    }
    return 0;
}

// Streaming pass of the exact-incremental formulation: the nm = 4c NEWEST mel frames of the window through the same
// conv front-end on per-layer 6-row histories (every tensor that feeds a k7 conv keeps its own history; ConvNeXt
// blocks are therefore out-of-place).  The resulting c token rows land in d2c[T2-c, T2).
// part 0: the whole pass; 1: front (mel, stem, the first kStreamCut stages and the transition out of them); 2: the rest.  The
// pipelined step runs the front on the main stream ahead of the head pass and the back on the side stream ahead of the
// transformer, which evens out the two encoder chains.
int enc_frontend_stream(sva_batch* b, const int* step_ptr, int n_chunk, int add, int part) {
    sva_engine* e = b->e;
    const EncFront& F = e->tokf;
    const sva_config& c = e->cfg;
    const int B = b->B, nm = 4 * b->p.chunk_frames;
    hipStream_t st = b->stream;
    EncStream& S = b->es;
    if (part != 2) {
    SVA_TRY(launch_stft_mag_ring(b->ring, step_ptr, n_chunk, add, B, b->N, e->twiddle, e->hann, S.mag, 1088, (long)nm * 1088, b->T0 - nm, nm, st));
    {
        ConvGemm p;
        p.act = ACT_LOGCLAMP;
        SVA_TRY(gemm_call(b, S.mag, (long)nm * 1088, 0, 1088, B, nm, 1, 1, 1, 1088, e->mel_fb, S.mel.p, S.mel.bstride, (long)S.mel.H * c.n_mels,
                          c.n_mels, p));
    }
    SVA_TRY(conv_act(b, S.mel, nm, 1, 1, 7, F.stem, S.tmp0));
    SVA_TRY(launch_layernorm_rows(S.tmp0.p, S.tmp0.bstride, 0, c.enc_dims[0], B, nm, c.enc_dims[0], F.stem_lnw, F.stem_lnb, 1e-6f,
                                  S.x[0][0].p, S.x[0][0].bstride, (long)S.x[0][0].H * c.enc_dims[0], c.enc_dims[0], st));
    }
    const int cut = b->stream_cut;
    for (int i = 0; i < 4; ++i) {
        if (part == 1 && i >= cut) return 0;
        if (part == 2 && i < cut) continue;
        const int C = c.enc_dims[i];
        const int nb = (int)F.stages[i].size();
        for (int j = 0; j < nb; ++j) {
            Act& out = j + 1 < nb ? S.x[i][j + 1] : S.xout[i];
            SVA_TRY(cnx_block_t(b, F.stages[i][j], S.x[i][j], nm, S.h1, (long)nm * C, S.h2, (long)nm * 4 * C, &out));
        }
        if (i < 3) {
            const int Cn = c.enc_dims[i + 1];
            SVA_TRY(launch_layernorm_rows(S.xout[i].p, S.xout[i].bstride, 0, C, B, nm, C, F.trans_lnw[i + 1], F.trans_lnb[i + 1], 1e-6f,
                                          S.h1, (long)nm * C, 0, C, st));
            SVA_TRY(gemm_call(b, S.h1, (long)nm * C, 0, C, B, nm, 1, 1, 1, C, F.trans[i + 1], S.x[i + 1][0].p, S.x[i + 1][0].bstride,
                              (long)S.x[i + 1][0].H * Cn, Cn));
        }
    }
    const int D = c.tr_dim;
    SVA_TRY(launch_layernorm_rows(S.xout[3].p, S.xout[3].bstride, 0, D, B, nm, D, F.final_lnw, F.final_lnb, 1e-6f, S.feat.p, S.feat.bstride, 0, D, st));
    SVA_TRY(conv_act(b, S.feat, nm / 2, 2, 1, 2, F.ds_conv[0], S.d1));
    SVA_TRY(cnx_block_t(b, F.ds_cnx[0], S.d1, nm / 2, S.h1, (long)(nm / 2) * D, S.h2, (long)(nm / 2) * 4 * D, &S.d1o));
    SVA_TRY(conv_act(b, S.d1o, nm / 4, 2, 1, 2, F.ds_conv[1], S.d2));
    Act tail = b->d2c;                       // rows [T2 - c, T2) of the steady token cache
    tail.p = b->d2c.p + (long)(b->T2 - nm / 4) * D;
    tail.H = 0;
    SVA_TRY(cnx_block_t(b, F.ds_cnx[1], S.d2, nm / 4, S.h1, (long)(nm / 4) * D, S.h2, (long)(nm / 4) * 4 * D, &tail));
    SVA_TRY(launch_shift_history(S.d_shift, S.n_shift, B, st));
    return 0;
}

__global__ void copy_tokens_kernel(const float* __restrict__ tok, long tok_bstride, int Ht, int gap, int c, float* __restrict__ d2c, long d_bstride,
                                   int T2, int D) {
    // head tokens -> d2c rows [0, Ht); newest c tokens -> d2c rows [T2 - c, T2)
    const int r = blockIdx.x, bi = blockIdx.y;
    const int src = r < Ht ? r : Ht + gap + (r - Ht);
    const int dst = r < Ht ? r : T2 - c + (r - Ht);
    const float4* s = reinterpret_cast<const float4*>(tok + (long)bi * tok_bstride + (long)src * D);
    float4* d = reinterpret_cast<float4*>(d2c + (long)bi * d_bstride + (long)dst * D);
    for (int i = threadIdx.x; i < D / 4; i += blockDim.x) d[i] = s[i];
}

// Merged incremental front-end pass (see EncMerged): head rows (the first 4*Ht mel frames of the window, zero left padding as in
// the reference's window pass) and the 4c newest mel frames (on per-layer 6-row histories) through ONE sequence of launches.
// Results: token rows [0, Ht) and [T2 - c, T2) of d2c.
// part 0 = everything; part 1 = up to the token features (does not touch the token cache d2c); part 2 = the hand-over: head and
// new tokens -> d2c, then the layers' history rows slide (the pipelined step waits for transformer(n-1)'s first layer only here).
// part 3 / 4 = the cut the balanced pipeline uses: 3 = STFT .. backbone .. final LayerNorm -> feat[fpar] + the history shift of
// those layers; 4 = quantizer downsampler (2 x (conv k2 s2 + ConvNeXt block)) from feat[fpar] with its own scratch, tokens -> d2c,
// its two history shifts (runs on the side stream in front of the transformer)
int enc_frontend_merged(sva_batch* b, const int* step_ptr, int n_chunk, int add, int part, int fpar) {
    sva_engine* e = b->e;
    const EncFront& F = e->tokf;
    const sva_config& c = e->cfg;
    EncMerged& M = b->em;
    const int B = b->B, Hh = M.Hh, nm = M.nm, R0 = Hh + 6 + nm, ch = b->p.chunk_frames;
    hipStream_t st = b->stream;
    const int D = c.tr_dim;
    Act& feat = fpar ? M.feat2 : M.feat;
    float* h1 = part == 4 ? M.h1b : M.h1;
    float* h2 = part == 4 ? M.h2b : M.h2;
    const int R1 = Hh / 2 + 6 + nm / 2, R2 = Hh / 4 + 6 + nm / 4;
    const int n_shift_ds = 2;                       // the last two descriptors are the downsampler's (d1, d2)
    if (part == 2) {
        hipLaunchKernelGGL(copy_tokens_kernel, dim3(b->Ht + ch, B), dim3(128), 0, st, M.tok.p, M.tok.bstride, b->Ht, 6, ch, b->d2c.p, b->d2c.bstride, b->T2, D);
        SVA_TRY(launch_shift_history(M.d_shift, M.n_shift, B, st));
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (part == 4) {
        SVA_TRY(gemm_call(b, feat.p, feat.bstride, 0, D, B, Hh / 2, 2, 1, 2, D, F.ds_conv[0], M.d1.p, M.d1.bstride, (long)M.d1.H * D, D));
        SVA_TRY(gemm_call(b, feat.p, feat.bstride, (long)(Hh + 6) * D, D, B, nm / 2, 2, 1, 2, D, F.ds_conv[0], M.d1.p, M.d1.bstride,
                          (long)(M.d1.H + Hh / 2 + 6) * D, D));
        SVA_TRY(cnx_block_t(b, F.ds_cnx[0], M.d1, R1, h1, (long)R1 * D, h2, (long)R1 * 4 * D, &M.d1o));
        SVA_TRY(gemm_call(b, M.d1o.p, M.d1o.bstride, 0, D, B, Hh / 4, 2, 1, 2, D, F.ds_conv[1], M.d2.p, M.d2.bstride, (long)M.d2.H * D, D));
        SVA_TRY(gemm_call(b, M.d1o.p, M.d1o.bstride, (long)(Hh / 2 + 6) * D, D, B, nm / 4, 2, 1, 2, D, F.ds_conv[1], M.d2.p, M.d2.bstride,
                          (long)(M.d2.H + Hh / 4 + 6) * D, D));
        SVA_TRY(cnx_block_t(b, F.ds_cnx[1], M.d2, R2, h1, (long)R2 * D, h2, (long)R2 * 4 * D, &M.tok));
        hipLaunchKernelGGL(copy_tokens_kernel, dim3(b->Ht + ch, B), dim3(128), 0, st, M.tok.p, M.tok.bstride, b->Ht, 6, ch, b->d2c.p, b->d2c.bstride, b->T2, D);
        SVA_TRY(launch_shift_history(M.d_shift + (M.n_shift - n_shift_ds), n_shift_ds, B, st));
        SVA_HIP(hipGetLastError());
        return 0;
    }
    const long mag_bs = (long)R0 * 1088;
    // head frames 0 .. Hh and the nm newest frames of the window in one launch (output rows [0, Hh) and [Hh + 6, R0))
    SVA_TRY(launch_stft_mag_ring2(b->ring, step_ptr, n_chunk, add, B, b->N, e->twiddle, e->hann, M.mag, 1088, mag_bs, 0, Hh, b->T0 - nm, nm, Hh + 6, st));
    {
        ConvGemm p;
        p.act = ACT_LOGCLAMP; p.skip_lo = Hh; p.skip_hi = Hh + 6;
        SVA_TRY(gemm_call(b, M.mag, mag_bs, 0, 1088, B, R0, 1, 1, 1, 1088, e->mel_fb, M.mel.p, M.mel.bstride, (long)M.mel.H * c.n_mels, c.n_mels, p));
    }
    const int C0 = c.enc_dims[0];
    SVA_TRY(gemm_call(b, M.mel.p, M.mel.bstride, 0, c.n_mels, B, R0, 1, 1, 7, c.n_mels, F.stem, M.stem, (long)R0 * C0, 0, C0));
    {   // LayerNorm of the stem output into the first block's input: head rows and new rows (its history rows stay)
        Act& X = M.x[0][0];
        SVA_TRY(launch_layernorm_rows(M.stem, (long)R0 * C0, 0, C0, B, R0, C0, F.stem_lnw, F.stem_lnb, 1e-6f, X.p, X.bstride, (long)X.H * C0, C0, st,
                                      Hh, Hh + 6));
    }
    for (int i = 0; i < 4; ++i) {
        const int C = c.enc_dims[i];
        const int nb = (int)F.stages[i].size();
        for (int j = 0; j < nb; ++j) {
            const bool last = j + 1 == nb;
            Act& out = last ? M.xout[i] : M.x[i][j + 1];
            SVA_TRY(cnx_block_t(b, F.stages[i][j], M.x[i][j], R0, M.h1, (long)R0 * C, M.h2, (long)R0 * 4 * C, &out, last ? 0 : Hh, last ? 0 : Hh + 6));
        }
        if (i < 3) {
            const int Cn = c.enc_dims[i + 1];
            SVA_TRY(launch_layernorm_rows(M.xout[i].p, M.xout[i].bstride, 0, C, B, R0, C, F.trans_lnw[i + 1], F.trans_lnb[i + 1], 1e-6f, M.h1, (long)R0 * C, 0, C, st));
            ConvGemm p;
            p.skip_lo = Hh; p.skip_hi = Hh + 6;
            Act& X = M.x[i + 1][0];
            SVA_TRY(gemm_call(b, M.h1, (long)R0 * C, 0, C, B, R0, 1, 1, 1, C, F.trans[i + 1], X.p, X.bstride, (long)X.H * Cn, Cn, p));
        }
    }
    SVA_TRY(launch_layernorm_rows(M.xout[3].p, M.xout[3].bstride, 0, D, B, R0, D, F.final_lnw, F.final_lnb, 1e-6f, feat.p, feat.bstride, 0, D, st));
    if (part == 3) return launch_shift_history(M.d_shift, M.n_shift - n_shift_ds, B, st, 1, b->step_bump, 1);       // (+ the chain's step counter)
    // BSQ downsample x2 (conv k2 s2 + ConvNeXtBlock, bsq_no_upsample.py:48-61); the strided convs run per row group
    SVA_TRY(gemm_call(b, feat.p, feat.bstride, 0, D, B, Hh / 2, 2, 1, 2, D, F.ds_conv[0], M.d1.p, M.d1.bstride, (long)M.d1.H * D, D));
    SVA_TRY(gemm_call(b, feat.p, feat.bstride, (long)(Hh + 6) * D, D, B, nm / 2, 2, 1, 2, D, F.ds_conv[0], M.d1.p, M.d1.bstride,
                      (long)(M.d1.H + Hh / 2 + 6) * D, D));
    SVA_TRY(cnx_block_t(b, F.ds_cnx[0], M.d1, R1, M.h1, (long)R1 * D, M.h2, (long)R1 * 4 * D, &M.d1o));
    SVA_TRY(gemm_call(b, M.d1o.p, M.d1o.bstride, 0, D, B, Hh / 4, 2, 1, 2, D, F.ds_conv[1], M.d2.p, M.d2.bstride, (long)M.d2.H * D, D));
    SVA_TRY(gemm_call(b, M.d1o.p, M.d1o.bstride, (long)(Hh / 2 + 6) * D, D, B, nm / 4, 2, 1, 2, D, F.ds_conv[1], M.d2.p, M.d2.bstride,
                      (long)(M.d2.H + Hh / 4 + 6) * D, D));
    SVA_TRY(cnx_block_t(b, F.ds_cnx[1], M.d2, R2, M.h1, (long)R2 * D, M.h2, (long)R2 * 4 * D, &M.tok));
    if (part == 1) return 0;
    return enc_frontend_merged(b, step_ptr, n_chunk, add, 2);
}

// pre_module (8-layer causal transformer on T2 tokens, windowed_transformer.py:103-143) + BSQ.  Reads the token
// features from `xin` without modifying them (the exact-incremental path keeps them as its steady cache).
// need_rows > 0: only the codes of the LAST need_rows tokens are consumed by the caller (streaming keeps codes[-c:],
// infer_arvc.py:518), so the last layer runs its query / output / FFN rows, the final norm and BSQ for those rows
// only (its K and V are still computed for every token).  need_rows = 0: all T2 codes (seam API).
int enc_transformer(sva_batch* b, const Act& xin, int need_rows, int part) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int B = b->B, T2 = b->T2, D = c.tr_dim, I = c.tr_inter;
    hipStream_t st = b->stream;
    const float* xr = xin.p;                 // residual source of the current sub-layer
    long xr_bs = xin.bstride, xr_off = (long)xin.H * D;
    float* xw = b->tr_x;                     // work copy [B][T2][D]
    const long xw_bs = (long)T2 * D;
    const int nl = (int)e->tr.size();
    // part 1 = layer 0 only, part 2 = layers 1.. + final norm + BSQ (the pipelined stage graphs are cut where `xin` is released)
    const int l_lo = part == 2 ? 1 : 0, l_hi = part == 1 ? 1 : nl;
    if (part == 2) { xr = xw; xr_bs = xw_bs; xr_off = 0; }
    for (int li = l_lo; li < l_hi; ++li) {
        TrLayer& L = e->tr[li];
        const bool tail = need_rows > 0 && li == nl - 1;
        const int Tr = tail ? need_rows : T2;             // rows of this layer's output that are needed
        const int r0 = T2 - Tr;
        // RMSNorm folded into the projection when the small-M kernel runs it (few streams); a separate pass otherwise
        if (conv_gemm_can_fuse_rms(B * T2, 3 * D)) {
            ConvGemm pn;
            pn.rms_w = L.attn_norm; pn.rms_eps = 1e-5f;
            SVA_TRY(gemm_call(b, xr, xr_bs, xr_off, D, B, T2, 1, 1, 1, D, L.wqkv, b->tr_qkv, (long)T2 * 3 * D, 0, 3 * D, pn));
        } else if (planes_input(L.wqkv, (long)B * T2, planes_streams(b))) {        // RMSNorm output as operand planes (in tr_hn's own memory)
            ConvGemm pn;
            pn.Ap = reinterpret_cast<unsigned short*>(b->tr_hn); pn.ap_pstride = (long)B * T2 * D; pn.ap_rows = (long)B * T2;
            SVA_TRY(launch_rmsnorm_rows(xr, xr_bs, xr_off, D, B, T2, D, L.attn_norm, 1e-5f, b->tr_hn, (long)T2 * D, 0, D, st,
                                        reinterpret_cast<unsigned short*>(b->tr_hn), pn.ap_pstride, planes_count(L.wqkv.pmode), pn.ap_rows));
            SVA_TRY(gemm_call(b, b->tr_hn, (long)T2 * D, 0, D, B, T2, 1, 1, 1, D, L.wqkv, b->tr_qkv, (long)T2 * 3 * D, 0, 3 * D, pn));
        } else {
            SVA_TRY(launch_rmsnorm_rows(xr, xr_bs, xr_off, D, B, T2, D, L.attn_norm, 1e-5f, b->tr_hn, (long)T2 * D, 0, D, st));
            SVA_TRY(gemm_call(b, b->tr_hn, (long)T2 * D, 0, D, B, T2, 1, 1, 1, D, L.wqkv, b->tr_qkv, (long)T2 * 3 * D, 0, 3 * D));
        }
        ConvGemm po;
        if (planes_input(L.wo, (long)B * Tr, planes_streams(b)) && enc_attention_can_write_planes(T2)) {          // attention output as operand planes (in tr_att's own memory)
            po.Ap = reinterpret_cast<unsigned short*>(b->tr_att); po.ap_pstride = (long)B * T2 * D; po.ap_rows = (long)B * T2;
            SVA_TRY(launch_enc_attention(b->tr_qkv, e->rope_enc, B, T2, c.tr_heads, 64, b->tr_att, r0, st,
                                         reinterpret_cast<unsigned short*>(b->tr_att), po.ap_pstride, planes_count(L.wo.pmode), po.ap_rows));
        } else {
            SVA_TRY(launch_enc_attention(b->tr_qkv, e->rope_enc, B, T2, c.tr_heads, 64, b->tr_att, r0, st));
        }
        po.gamma = L.ls_attn;
        po.res = xr; po.r_bstride = xr_bs; po.r_off = xr_off + (long)r0 * D; po.ldr = D;
        SVA_TRY(gemm_call(b, b->tr_att, (long)T2 * D, (long)r0 * D, D, B, Tr, 1, 1, 1, D, L.wo, xw, xw_bs, (long)r0 * D, D, po));
        if (li == 0 && b->tr_l0_event) SVA_HIP(hipEventRecord(b->tr_l0_event, st));     // `xin` is not read past this point
        xr = xw; xr_bs = xw_bs; xr_off = 0;
        ConvGemm pg;
        pg.w13 = 1;
        ConvGemm pd;
        if (planes_edge(L.w13, L.w2, (long)B * Tr, planes_streams(b))) {          // SwiGLU output -> w2 as operand planes, in tr_g's memory
            pg.Cp = reinterpret_cast<unsigned short*>(b->tr_g); pg.cp_pstride = (long)B * T2 * I; pg.cp_rows = (long)B * T2;
            pd.Ap = pg.Cp; pd.ap_pstride = pg.cp_pstride; pd.ap_rows = pg.cp_rows;
        }
        if (conv_gemm_can_fuse_rms(B * Tr, 2 * I)) {
            pg.rms_w = L.ffn_norm; pg.rms_eps = 1e-5f;
            SVA_TRY(gemm_call(b, xw, xw_bs, (long)r0 * D, D, B, Tr, 1, 1, 1, D, L.w13, b->tr_g, (long)T2 * I, (long)r0 * I, I, pg));
        } else {
            if (planes_input(L.w13, (long)B * Tr, planes_streams(b))) {
                pg.Ap = reinterpret_cast<unsigned short*>(b->tr_hn); pg.ap_pstride = (long)B * T2 * D; pg.ap_rows = (long)B * T2;
                SVA_TRY(launch_rmsnorm_rows(xw, xw_bs, (long)r0 * D, D, B, Tr, D, L.ffn_norm, 1e-5f, b->tr_hn, (long)T2 * D, (long)r0 * D, D, st,
                                            reinterpret_cast<unsigned short*>(b->tr_hn), pg.ap_pstride, planes_count(L.w13.pmode), pg.ap_rows));
            } else {
                SVA_TRY(launch_rmsnorm_rows(xw, xw_bs, (long)r0 * D, D, B, Tr, D, L.ffn_norm, 1e-5f, b->tr_hn, (long)T2 * D, (long)r0 * D, D, st));
            }
            SVA_TRY(gemm_call(b, b->tr_hn, (long)T2 * D, (long)r0 * D, D, B, Tr, 1, 1, 1, D, L.w13, b->tr_g, (long)T2 * I, (long)r0 * I, I, pg));
        }
        pd.gamma = L.ls_ffn;
        pd.res = xw; pd.r_bstride = xw_bs; pd.r_off = (long)r0 * D; pd.ldr = D;
        SVA_TRY(gemm_call(b, b->tr_g, (long)T2 * I, (long)r0 * I, I, B, Tr, 1, 1, 1, I, L.w2, xw, xw_bs, (long)r0 * D, D, pd));
    }
    if (part == 1) return 0;
    const int Tr = need_rows > 0 ? need_rows : T2, r0 = T2 - Tr;
    // final RMSNorm fused into the BSQ projection (normalised rows still land in tr_z for the "z" tap)
    SVA_CHECK(xw_bs == (long)T2 * D, "enc_transformer: work copy layout");
    SVA_TRY(launch_bsq(xw, xw_bs, (long)r0 * D, D, B, Tr, D, e->tr_norm, 1e-5f, b->tr_z, e->bsq_W, e->bsq_b, c.bsq_bits, b->d_codes, b->T2, r0,
                       b->d_u, st));
    return 0;
}

// full-window formulation (reference: the whole 128-frame window is re-encoded every chunk, infer_arvc.py:505-508)
int encode(sva_batch* b, const int* step_ptr, int n_chunk, int add) {
    SVA_TRY(enc_frontend_window(b, step_ptr, n_chunk, add, b->T0));
    return enc_transformer(b, b->d2, 0);
}

// exact-incremental formulation (SURVEY.md §7 hard part 1): window rows whose causal receptive field still touches
// the zero left padding -- mel frames 0..116, tokens 0..38 -- are recomputed every chunk ("head pass" on the first
// 160 mel frames = 40 tokens); every later token is the true causal feature of its absolute time, computed once by
// the streaming pass when it entered the window and kept in d2c, which slides by c tokens per chunk.  The 8-layer
// transformer + BSQ always run on all T2 tokens.  Same values as the window pass up to fp32 summation order.
int encode_incremental(sva_batch* b, const int* step_ptr, int n_chunk, int add, bool transformer_too) {
    const int c = b->p.chunk_frames;
    hipStream_t st = b->stream;
    SVA_TRY(launch_shift_history(b->d_shift_d2c, 1, b->B, st, 16));                   // steady tokens slide down by c
    if (b->enc_merged) {
        SVA_TRY(enc_frontend_merged(b, step_ptr, n_chunk, add));
        if (transformer_too) return enc_transformer(b, b->d2c, b->p.chunk_frames);
        return 0;
    }
    // the head pass and the streaming pass are independent chains: run the short one on a side stream
    const bool par = b->concurrency;
    if (par) {
        SVA_TRY(stream_fork(b, st, b->aux[0]));
        b->stream = b->aux[0];
    }
    int rc = enc_frontend_stream(b, step_ptr, n_chunk, add);                     // c newest tokens -> d2c tail
    b->stream = st;
    if (rc) return rc;
    SVA_TRY(enc_frontend_window(b, step_ptr, n_chunk, add, 4 * b->Ht, nullptr, &b->d2c));   // head pass -> d2c rows [0, Ht) directly
    if (par) SVA_TRY(stream_fork(b, b->aux[0], st));                             // join
    (void)c;
    if (transformer_too) return enc_transformer(b, b->d2c, b->p.chunk_frames);
    return 0;
}

// ---- A: slow / fast transformer passes ------------------------------------------------------------
// rows [M, dim] in b->ax, slot/pos arrays on device; KV written at pos, attention over 0..pos
// run_slot == -2: the rows are the decode frame's pairs (2 s, 2 s + 1) = positions (p, p + 1) of slot s: one attention workgroup per pair and head;
// run_slot >= 0: the rows sit at CONSECUTIVE positions run_pos0 .. run_pos0 + M - 1 of that one slot (prompt prefill, re-prefill,
// offline generate) -- their attention runs as the flash-style MFMA kernel instead of one workgroup per (head, row)
int ar_layers_pass(sva_batch* b, std::vector<TrLayer>& layers, int M, const int* d_slot, const int* d_pos, const float* rope,
                   float* kv, long kv_layer, long kv_slot, int S, float* x, int run_slot, int run_pos0) {
    const sva_config& c = b->e->cfg;
    const int D = c.ar_dim, I = c.ar_inter, H = c.ar_heads;
    hipStream_t st = b->stream;
    const bool half_kv = b->kv_half && S > 8;           // the slow cache of an ar_dtype = 1 batch (the fast cache stays fp32)
    if (M <= 4 && b->fused_decode && !half_kv) {
        // decode at B <= 2: 5 launches per layer -- QKV GEMV (+RMSNorm, +RoPE, +KV write), attention, wo GEMV (+residual),
        // w1|w3 GEMV (+RMSNorm, +SwiGLU), w2 GEMV (+residual)
        for (size_t l = 0; l < layers.size(); ++l) {
            TrLayer& L = layers[l];
            float* cache = kv + (long)l * kv_layer;
            Gemv q;
            q.X = x; q.ldx = D; q.M = M; q.W = L.wqkv.W; q.N = 3 * D; q.K = D; q.norm_w = L.attn_norm; q.eps = 1e-5f;
            q.Y = b->aqkv; q.ldy = 3 * D; q.mode = 2; q.slot = d_slot; q.pos = d_pos; q.rope = rope; q.kv = cache;
            q.kv_slot_stride = kv_slot; q.S = S; q.H = H;
            SVA_TRY(launch_gemv(q, st));
            Gemv o;
            o.M = M; o.W = L.wo.W; o.N = D; o.K = D; o.res = x; o.ldr = D; o.Y = x; o.ldy = D;
            if (S <= 8 && M <= 2) {        // fast AR: attention over <= 8 codebook positions recomputed inside the wo GEMV
                o.X = b->aqkv; o.ldx = 3 * D; o.mode = 3; o.slot = d_slot; o.pos = d_pos; o.kv = cache; o.kv_slot_stride = kv_slot;
                o.S = S; o.H = H;
            } else if (M > 2) {
                SVA_TRY(launch_ar_attention<float>(b->aqkv, M, H, 64, d_slot, d_pos, cache, kv_slot, S, b->aatt, st));
                o.X = b->aatt; o.ldx = D;
            } else {                       // slow AR: split-key attention (12 heads x M rows alone leave the chip idle), merged by the wo GEMV
                const int splits = 8;
                SVA_TRY(launch_ar_attention<float>(b->aqkv, M, H, 64, d_slot, d_pos, cache, kv_slot, S, nullptr, st, b->aatt_part, splits));
                o.X = b->aatt_part; o.ldx = 0; o.mode = 4; o.S = splits; o.H = H;
            }
            SVA_TRY(launch_gemv(o, st));
            Gemv u;
            u.X = x; u.ldx = D; u.M = M; u.W = L.w13.W; u.N = 2 * I; u.K = D; u.norm_w = L.ffn_norm; u.eps = 1e-5f;
            u.Y = b->ag; u.ldy = I; u.mode = 1;
            SVA_TRY(launch_gemv(u, st));
            Gemv dn;
            dn.X = b->ag; dn.ldx = I; dn.M = M; dn.W = L.w2.W; dn.N = D; dn.K = I; dn.res = x; dn.ldr = D; dn.Y = x; dn.ldy = D;
            SVA_TRY(launch_gemv(dn, st));
        }
        return 0;
    }
    for (size_t l = 0; l < layers.size(); ++l) {
        TrLayer& L = layers[l];
        float* cache = kv + (long)l * kv_layer;
        // RMSNorm folded into the projection whenever the small-M kernel runs it (M up to a few hundred rows)
        if (conv_gemm_can_fuse_rms(M, 3 * D)) {
            ConvGemm pn;
            pn.rms_w = L.attn_norm; pn.rms_eps = 1e-5f;
            SVA_TRY(gemm_call(b, x, (long)M * D, 0, D, 1, M, 1, 1, 1, D, L.wqkv, b->aqkv, (long)M * 3 * D, 0, 3 * D, pn));
        } else {
            SVA_TRY(launch_rmsnorm_rows(x, (long)M * D, 0, D, 1, M, D, L.attn_norm, 1e-5f, b->ahn, (long)M * D, 0, D, st));
            SVA_TRY(gemm_call(b, b->ahn, (long)M * D, 0, D, 1, M, 1, 1, 1, D, L.wqkv, b->aqkv, (long)M * 3 * D, 0, 3 * D));
        }
        if (S <= 8) {
            SVA_TRY(launch_ar_fast_attention(b->aqkv, M, H, d_slot, d_pos, rope, cache, kv_slot, S, b->aatt, st));
        } else if (half_kv) {
            __half* ch = reinterpret_cast<__half*>(kv) + (long)l * kv_layer;
            SVA_TRY(launch_rope_kvwrite<__half>(b->aqkv, M, H, 64, d_slot, d_pos, rope, ch, kv_slot, S, st));
            if (run_slot >= 0 && M >= 96) SVA_TRY(launch_ar_prefill_attention<__half>(b->aqkv, M, H, 64, run_slot, run_pos0, ch, kv_slot, S, b->aatt, st));
            else if (run_slot == -2 && M % 2 == 0 && debug_options().ar_pairs) SVA_TRY(launch_ar_attention_pairs<__half>(b->aqkv, M, H, 64, d_slot, d_pos, ch, kv_slot, S, b->aatt, st));
            else SVA_TRY(launch_ar_attention<__half>(b->aqkv, M, H, 64, d_slot, d_pos, ch, kv_slot, S, b->aatt, st));
        } else {
            SVA_TRY(launch_rope_kvwrite<float>(b->aqkv, M, H, 64, d_slot, d_pos, rope, cache, kv_slot, S, st));
            if (run_slot >= 0 && M >= 96) SVA_TRY(launch_ar_prefill_attention<float>(b->aqkv, M, H, 64, run_slot, run_pos0, cache, kv_slot, S, b->aatt, st));
            else if (run_slot == -2 && M % 2 == 0 && debug_options().ar_pairs) SVA_TRY(launch_ar_attention_pairs<float>(b->aqkv, M, H, 64, d_slot, d_pos, cache, kv_slot, S, b->aatt, st));
            else SVA_TRY(launch_ar_attention<float>(b->aqkv, M, H, 64, d_slot, d_pos, cache, kv_slot, S, b->aatt, st));
        }
        ConvGemm po;
        po.res = x; po.r_bstride = (long)M * D; po.r_off = 0; po.ldr = D;
        SVA_TRY(gemm_call(b, b->aatt, (long)M * D, 0, D, 1, M, 1, 1, 1, D, L.wo, x, (long)M * D, 0, D, po));
        ConvGemm pg;
        pg.w13 = 1;
        if (conv_gemm_can_fuse_rms(M, 2 * I)) {
            pg.rms_w = L.ffn_norm; pg.rms_eps = 1e-5f;
            SVA_TRY(gemm_call(b, x, (long)M * D, 0, D, 1, M, 1, 1, 1, D, L.w13, b->ag, (long)M * I, 0, I, pg));
        } else {
            SVA_TRY(launch_rmsnorm_rows(x, (long)M * D, 0, D, 1, M, D, L.ffn_norm, 1e-5f, b->ahn, (long)M * D, 0, D, st));
            SVA_TRY(gemm_call(b, b->ahn, (long)M * D, 0, D, 1, M, 1, 1, 1, D, L.w13, b->ag, (long)M * I, 0, I, pg));
        }
        ConvGemm pd;
        pd.res = x; pd.r_bstride = (long)M * D; pd.r_off = 0; pd.ldr = D;
        SVA_TRY(gemm_call(b, b->ag, (long)M * I, 0, I, 1, M, 1, 1, 1, I, L.w2, x, (long)M * D, 0, D, pd));
    }
    return 0;
}



// one decoded frame for every stream (decode_one_token_ar, dual_ar_stream.py:1168-1219)
int ar_frame_tail(sva_batch* b, int ci, long hid_stride, long hid_off, const long long* codes, int codes_ld, int code_off, int last_pos_inc);

int ar_decode_frame_mega(sva_batch* b, int ci, const long long* codes, int code_off);
int ar_decode_frame_batch(sva_batch* b, int ci);

// Persistent decode kernels (ar_decode.hip, ar_batch.hip) need every workgroup of their grid resident at once: one 8-wave /
// 256-register workgroup per CU.  Two batches of one engine on different streams (a pipelined batch next to a synchronous one, an
// ar_decode batch next to an ar_batch batch) would otherwise co-schedule two such grids, both half-resident, both spinning into their
// timeout -- so EVERY persistent launch of an engine (eager, or the replay of a captured AR stage) is chained behind the previous one
// of another batch by an event, under the engine's mutex (batches may be driven by different host threads).
PersistentChain::PersistentChain(sva_batch* b_, hipStream_t st_, bool active_) : b(b_), st(st_), active(active_), lk(b_->e->mega_mu, std::defer_lock) {
    if (!active) return;
    lk.lock();
    sva_engine* e = b->e;
    // The only batch of its engine that decodes persistently has nobody to be ordered against: no wait, no event record (a record is a barrier
    // packet on the AR queue every frame).  A second such batch synchronises the device when it is created (engine.hip), so launches that were
    // enqueued without a record are complete before the chain is needed.
    if (e->persistent_batches <= 1) { active = false; lk.unlock(); return; }
    if (e->mega_ev_valid && e->mega_last != b && hipStreamWaitEvent(st, e->mega_ev, 0) != hipSuccess) {
        set_error("PersistentChain: hipStreamWaitEvent failed");
        rc = 1;
    }
}
int PersistentChain::finish() {
    if (!active) return 0;
    sva_engine* e = b->e;
    if (e->mega_ev) {
        SVA_HIP(hipEventRecord(e->mega_ev, st));
        e->mega_ev_valid = true; e->mega_last = b;
    }
    return 0;
}

int ar_decode_frame(sva_batch* b, int ci) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int B = b->B, D = c.ar_dim, chunk = b->p.chunk_frames;
    hipStream_t st = b->stream;
    const int code_off = b->T2 - chunk + ci;
    const bool persistent = (b->use_mega || b->use_abatch) && !b->edits_on;     // (sampler edits run on the multi-launch decode: same KV, positions and counters)
    if (persistent) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool eager = hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;     // (events of other batches cannot enter a capture:
        PersistentChain chain(b, st, eager);                                                                     //  the replay of a captured AR stage is chained as a whole, engine.hip)
        SVA_TRY(chain.rc);
        if (b->use_mega) {
            SVA_TRY(ar_decode_frame_mega(b, ci, b->d_codes, code_off));
        } else {
            hipLaunchKernelGGL(ar_prepare_step_kernel, dim3(B), dim3(256), 0, st, b->cached_audio_emb, e->content_emb, b->d_codes, b->T2,
                               code_off, b->d_last_pos, D, b->ax, b->d_slot, b->d_pos, b->d_step_content, chunk, ci);
            SVA_TRY(ar_decode_frame_batch(b, ci));
        }
        return chain.finish();
    }
    hipLaunchKernelGGL(ar_prepare_step_kernel, dim3(B), dim3(256), 0, st, b->cached_audio_emb, e->content_emb, b->d_codes, b->T2,
                       code_off, b->d_last_pos, D, b->ax, b->d_slot, b->d_pos, b->d_step_content, chunk, ci);
    SVA_TRY(ar_layers_pass(b, e->ar_layers, 2 * B, b->d_slot, b->d_pos, e->rope_ar, (float*)b->kv_slow, b->kv_slow_layer,
                           b->kv_slow_slot, c.max_seq_len, b->ax, -2));         // (-2: ar_prepare_step_kernel's row pairs)
    return ar_frame_tail(b, ci, (long)2 * D, (long)D, b->d_codes, b->T2, code_off, 2);
}

// one decoded frame of EVERY stream of the batch in one launch of the batched persistent kernel (ar_batch.hip); the frame's input
// tokens are already in b->ax (ar_prepare_step_kernel)
int ar_decode_frame_batch(sva_batch* b, int ci) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const bool half = c.ar_dtype == 1;
    ArBatchArgs a;
    memset(&a, 0, sizeof(a));
    auto wsel = [&](const Lin& l) -> const void* { return half ? l.Wh : (const void*)l.W; };
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const TrLayer& L = e->ar_layers[l];
        a.slow[l] = ArLayerW{wsel(L.wqkv), wsel(L.wo), wsel(L.w13), wsel(L.w2), L.attn_norm, L.ffn_norm};
    }
    for (int l = 0; l < AR_FAST_LAYERS; ++l) {
        const TrLayer& L = e->ar_fast_layers[l];
        a.fast[l] = ArLayerW{wsel(L.wqkv), wsel(L.wo), wsel(L.w13), wsel(L.w2), L.attn_norm, L.ffn_norm};
    }
    a.out_w = wsel(e->ar_output); a.out_norm = e->ar_norm; a.fast_out_w = wsel(e->ar_fast_output); a.fast_norm = e->ar_fast_norm;
    a.codebook_emb = e->codebook_emb; a.fast_emb = e->fast_emb; a.rope_slow = e->rope_ar; a.rope_fast = e->rope_fast;
    a.B = b->B; a.G = b->abatch_G;
    a.cached_audio_emb = b->cached_audio_emb; a.last_pos = b->d_last_pos; a.nframes = b->d_nframes; a.seed = b->d_seed;
    a.kv_slow = b->kv_slow; a.kv_layer_stride = b->kv_slow_layer; a.kv_slot_stride = b->kv_slow_slot; a.S = c.max_seq_len;
    a.xs_in = b->ax;
    unsigned long long* Gr = b->d_ab_gran;
    const size_t* o = b->ab_offs;
    a.gxs = Gr + o[0]; a.gqkv = Gr + o[1]; a.gatt = Gr + o[2]; a.gg = Gr + o[3]; a.gxf = Gr + o[4]; a.gqkvf = Gr + o[5]; a.gattf = Gr + o[6];
    a.ggf = Gr + o[7]; a.gkvf = Gr + o[8]; a.glog = Gr + o[9]; a.gsem = Gr + o[10];
    a.epoch = b->d_ab_epoch; a.done = b->d_ab_epoch + 1; a.fail = b->d_ar_fail; a.fail_host = b->d_ar_fail_host; a.dbg = b->d_ar_dbg;
    a.slow_logits = b->slow_logits; a.fast_logits = b->fast_logits; a.hidden = b->hidden;
    a.sem = b->d_sem; a.tok_raw = b->d_tok_raw; a.tok = b->d_tok; a.step_audio = b->d_step_audio; a.pred_hist = b->d_pred_hist;
    a.hist_cap = b->hist_cap; a.chunk = b->p.chunk_frames; a.ci = ci;
    const int nstride = c.ar_vocab + c.num_codebooks * c.codebook_size;
    a.noise = b->noise_on_device ? nullptr : b->d_noise + (long)ci * nstride;
    a.noise_ld = (long)b->p.chunk_frames * nstride;
    a.forced = b->d_forced; a.use_forced = b->d_use_forced;
    const float tclamp = b->p.temperature > 1e-5f ? b->p.temperature : 1e-5f;
    a.inv_temp = 1.0f / tclamp; a.top_p = b->p.top_p; a.skip_semantic = b->p.skip_semantic;
    a.vocab = c.ar_vocab; a.codebook_size = c.codebook_size;
    return launch_ar_batch(a, half ? 1 : 0, b->stream);
}

// one decoded frame of a one-stream batch in ONE launch of the persistent kernel (ar_decode.hip)
int ar_decode_frame_mega(sva_batch* b, int ci, const long long* codes, int code_off) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    ArDecodeArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < AR_SLOW_LAYERS; ++l) {
        const TrLayer& L = e->ar_layers[l];
        a.slow[l] = ArLayerW{L.m_wqkv, L.m_wo, L.m_w13, L.m_w2, L.attn_norm, L.ffn_norm};
    }
    for (int l = 0; l < AR_FAST_LAYERS; ++l) {
        const TrLayer& L = e->ar_fast_layers[l];
        a.fast[l] = ArLayerW{L.m_wqkv, L.m_wo, L.m_w13, L.m_w2, L.attn_norm, L.ffn_norm};
    }
    a.out_w = e->m_output; a.out_norm = e->ar_norm; a.fast_out_w = e->m_fast_output; a.fast_norm = e->ar_fast_norm;
    a.content_emb = e->content_emb; a.codebook_emb = e->codebook_emb; a.fast_emb = e->fast_emb; a.rope_slow = e->rope_ar; a.rope_fast = e->rope_fast;
    a.codes = codes; a.code_off = code_off;
    a.cached_audio_emb = b->cached_audio_emb; a.last_pos = b->d_last_pos; a.nframes = b->d_nframes; a.seed = b->d_seed;
    a.kv_slow = b->kv_slow; a.kv_layer_stride = b->kv_slow_layer; a.S = c.max_seq_len; a.kv_fast = b->kv_fast_mega;
    a.gx = b->d_gran; a.gbig = a.gx + 2 * 768; a.gatt = a.gbig + 2 * 2304; a.glog = a.gatt + AR_WGS * 66; a.ga = a.glog + 1024;
    a.epoch = b->d_epoch; a.fail = b->d_ar_fail; a.fail_host = b->d_ar_fail_host; a.dbg = b->d_ar_dbg;
    a.slow_logits = b->slow_logits; a.fast_logits = b->fast_logits; a.hidden = b->hidden;
    a.sem = b->d_sem; a.tok_raw = b->d_tok_raw; a.tok = b->d_tok; a.step_audio = b->d_step_audio; a.pred_hist = b->d_pred_hist;
    a.step_content = b->d_step_content; a.hist_cap = b->hist_cap; a.chunk = b->p.chunk_frames; a.ci = ci;
    const int nstride = c.ar_vocab + c.num_codebooks * c.codebook_size;
    a.noise = b->noise_on_device ? nullptr : b->d_noise + (long)ci * nstride;
    a.forced = b->d_forced; a.use_forced = b->d_use_forced;
    const float tclamp = b->p.temperature > 1e-5f ? b->p.temperature : 1e-5f;
    a.inv_temp = 1.0f / tclamp; a.top_p = b->p.top_p; a.skip_semantic = b->p.skip_semantic;
    a.vocab = c.ar_vocab; a.codebook_size = c.codebook_size;
    // on its own CU partition the kernel pads its LDS request so that the 96 workgroups land on 96 different CUs; on shared CUs it
    // keeps its small footprint so that the other stages' GEMM workgroups fit beside it
    {   // strides between the per-stream blocks (elements of each pointer's type)
        const int ncb = c.num_codebooks, D = c.ar_dim;
        a.ss.codes = b->T2; a.ss.emb = D; a.ss.kv_slot = b->kv_slow_slot; a.ss.kv_fast = (long)AR_FAST_LAYERS * 8 * 2 * D;
        a.ss.gran = (long)ar_decode_granule_words(); a.ss.slow_logits = c.ar_vocab; a.ss.fast_logits = (long)ncb * c.codebook_size; a.ss.hidden = D;
        a.ss.tok = ncb; a.ss.step_audio = (long)ncb * b->p.chunk_frames; a.ss.pred_hist = (long)ncb * b->hist_cap;
        a.ss.step_content = b->p.chunk_frames; a.ss.noise = (long)b->p.chunk_frames * nstride; a.ss.forced = (long)ncb * b->p.chunk_frames;
    }
    const bool share = !b->ar_partitioned;
    // at most two streams per launch: 192 workgroups find a CU each and leave half of every register file to the other stages'
    // kernels; four streams in one launch (two 256-register workgroups on half of the CUs) measured 1.75 ms for the frame AND
    // locked the encoder / vocoder kernels out of those CUs (2.69 ms per pipelined step against 1.9 with two launches of two)
    const int per_launch = b->mega_per_launch;
    for (int s0 = 0; s0 < b->B; s0 += per_launch) {
        a.slot_base = s0;
        SVA_TRY(launch_ar_decode(a, c.ar_dtype == 1, b->kv_half, !share, b->stream, std::min(per_launch, b->B - s0)));
    }
    return 0;
}

// semantic head + 8-step fast AR + bookkeeping of one frame; the slow hidden state of slot s is row
// ax[s*hid_stride + hid_off .. +D)  (decode_one_token_ar, dual_ar_stream.py:1181-1219)
int ar_frame_tail(sva_batch* b, int ci, long hid_stride, long hid_off, const long long* codes, int codes_ld, int code_off, int last_pos_inc) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int B = b->B, D = c.ar_dim, chunk = b->p.chunk_frames, ncb = c.num_codebooks, cbs = c.codebook_size;
    hipStream_t st = b->stream;
    // hidden = pre-norm state of the content token (forward_generate :340-341); it also seeds the fast AR
    hipLaunchKernelGGL(copy_rows2_kernel, dim3(B), dim3(256), 0, st, b->ax, hid_stride, hid_off, b->hidden, b->xf, D);
    const int nstride = c.ar_vocab + ncb * cbs;
    const float* noise = b->noise_on_device ? nullptr : b->d_noise + (long)ci * nstride;
    const int ldn = chunk * nstride;
    const bool fused = B <= 4 && b->fused_decode;
    if (!b->p.skip_semantic) {
        if (fused) {
            Gemv hg;
            hg.X = b->hidden; hg.ldx = D; hg.M = B; hg.W = e->ar_output.W; hg.N = c.ar_vocab; hg.K = D; hg.norm_w = e->ar_norm;
            hg.eps = 1e-5f; hg.Y = b->slow_logits; hg.ldy = c.ar_vocab;
            SVA_TRY(launch_gemv(hg, st));
        } else if (conv_gemm_can_fuse_rms(B, c.ar_vocab) && debug_options().head_fuse) {        // the norm folded into the head's GEMM (a launch less per head: round 6)
            ConvGemm pn;
            pn.rms_w = e->ar_norm; pn.rms_eps = 1e-5f;
            SVA_TRY(gemm_call(b, b->hidden, (long)B * D, 0, D, 1, B, 1, 1, 1, D, e->ar_output, b->slow_logits, (long)B * c.ar_vocab, 0, c.ar_vocab, pn));
        } else {
            SVA_TRY(launch_rmsnorm_rows(b->hidden, (long)B * D, 0, D, 1, B, D, e->ar_norm, 1e-5f, b->ahn, (long)B * D, 0, D, st));
            SVA_TRY(gemm_call(b, b->ahn, (long)B * D, 0, D, 1, B, 1, 1, 1, D, e->ar_output, b->slow_logits, (long)B * c.ar_vocab, 0, c.ar_vocab));
        }
        if (b->edits_on && !b->edits_skip)
            SVA_TRY(launch_logit_edits(b->slow_logits, B, c.ar_vocab, c.ar_vocab, b->d_edit_prev, EDIT_CAP, b->d_edit_suppress, b->d_edit_params, st));
        SVA_TRY(launch_sampler(b->slow_logits, B, c.ar_vocab, c.ar_vocab, noise, ldn, b->d_seed, b->d_nframes, 0, 0, b->p.temperature,
                               b->p.top_p, b->d_sem, 1, st));
    }
    for (int cb = 0; cb < ncb; ++cb) {
        SVA_TRY(ar_layers_pass(b, e->ar_fast_layers, B, b->d_fast_slot, b->d_fast_pos + cb * B, e->rope_fast, (float*)b->kv_fast,
                               b->kv_fast_layer, b->kv_fast_slot, ncb, b->xf));
        float* lg = b->fast_logits + (long)cb * cbs;     // [B][8][cbs]
        if (fused) {
            Gemv fg;
            fg.X = b->xf; fg.ldx = D; fg.M = B; fg.W = e->ar_fast_output.W; fg.N = cbs; fg.K = D; fg.norm_w = e->ar_fast_norm; fg.eps = 1e-5f;
            fg.Y = lg; fg.ldy = ncb * cbs;
            SVA_TRY(launch_gemv(fg, st));
        } else if (conv_gemm_can_fuse_rms(B, cbs) && debug_options().head_fuse) {
            ConvGemm pn;
            pn.rms_w = e->ar_fast_norm; pn.rms_eps = 1e-5f;
            SVA_TRY(gemm_call(b, b->xf, (long)B * D, 0, D, 1, B, 1, 1, 1, D, e->ar_fast_output, lg, (long)B * ncb * cbs, 0, ncb * cbs, pn));
        } else {
            SVA_TRY(launch_rmsnorm_rows(b->xf, (long)B * D, 0, D, 1, B, D, e->ar_fast_norm, 1e-5f, b->ahn, (long)B * D, 0, D, st));
            SVA_TRY(gemm_call(b, b->ahn, (long)B * D, 0, D, 1, B, 1, 1, 1, D, e->ar_fast_output, lg, (long)B * ncb * cbs, 0, ncb * cbs));
        }
        if (b->edits_on && !b->edits_skip)          // codebook cb reads previous_tokens[cb + 1]; no suppress list (dual_ar_stream.py:1205-1213)
            SVA_TRY(launch_logit_edits(lg, B, cbs, ncb * cbs, b->d_edit_prev + (long)(cb + 1) * EDIT_CAP, EDIT_CAP, b->d_edit_suppress,
                                       b->d_edit_params + (cb + 1) * 4, st));
        // sample (+ teacher forcing) and gather the next fast-AR input embedding in the same launch
        const float* nz = noise ? noise + c.ar_vocab + (long)cb * cbs : nullptr;
        if (cbs <= 1024) {
            SVA_TRY(launch_sampler_small(lg, B, cbs, ncb * cbs, nz, ldn, b->d_seed, b->d_nframes, 1, cb * cbs, b->p.temperature, b->p.top_p,
                                         b->d_tok_raw + cb, b->d_tok + cb, ncb, b->d_forced + (long)cb * chunk + ci, ncb * chunk, b->d_use_forced,
                                         cb + 1 < ncb ? e->fast_emb : nullptr, D, b->xf, D, st));
        } else {
            SVA_TRY(launch_sampler(lg, B, cbs, ncb * cbs, nz, ldn, b->d_seed, b->d_nframes, 1, cb * cbs, b->p.temperature, b->p.top_p,
                                   b->d_tok_raw + cb, ncb, st));
            hipLaunchKernelGGL(apply_forced_kernel, dim3((B + 63) / 64), dim3(64), 0, st, b->d_tok_raw, b->d_forced, b->d_use_forced, chunk, ci,
                               cb, ncb, b->d_tok, B);
            if (cb + 1 < ncb) SVA_TRY(launch_gather_rows(e->fast_emb, b->d_tok + cb, ncb, 0, B, D, b->xf, D, st));
        }
    }
    // cached_new_audio_emb = embed(codes) (:834); positions advance by 2 (:835-836)
    SVA_TRY(launch_audio_embed(e->codebook_emb, b->d_tok, ncb, 1, B, ncb, cbs, D, b->cached_audio_emb, D, st));
    hipLaunchKernelGGL(ar_finish_frame_kernel, dim3((B + 63) / 64), dim3(64), 0, st, b->d_tok, ncb, b->d_last_pos, b->d_nframes,
                       b->d_pred_hist, b->hist_cap, b->d_step_audio, chunk, ci, codes, codes_ld, code_off, b->d_content_hist,
                       b->d_ncontent, B, last_pos_inc);
    SVA_HIP(hipGetLastError());
    return 0;
}

// prefill of ONE slot from prompt codes already staged in d_prompt_cc / d_prompt_ac (R frames)
int ar_prefill_slot(sva_batch* b, int slot, int R, bool tap_logits) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int D = c.ar_dim, d = b->p.delay, nspk = c.timbre_tokens + 1;
    hipStream_t st = b->stream;
    const int M = nspk + 2 * R;
    SVA_CHECK(M <= b->Mmax && M <= c.max_seq_len, "prompt too long for the KV cache");
    SVA_CHECK(R > d, "prompt must be longer than the delay");
    // speaker prefix: cat[context_in(timbre) (32 tok), style_in(style) (1 tok)]  (arvc_wrapper.py:108-109)
    SVA_TRY(gemm_call(b, b->d_timbre + (long)slot * c.timbre_tokens * c.timbre_dim, (long)c.timbre_tokens * c.timbre_dim, 0, c.timbre_dim, 1,
                      c.timbre_tokens, 1, 1, 1, c.timbre_dim, e->context_in, b->spk, (long)nspk * D, 0, D));
    SVA_TRY(gemm_call(b, b->d_style + (long)slot * c.style_dim, c.style_dim, 0, c.style_dim, 1, 1, 1, 1, 1, c.style_dim, e->style_in,
                      b->spk + (long)c.timbre_tokens * D, D, 0, D));
    hipLaunchKernelGGL(build_prompt_kernel, dim3(M), dim3(256), 0, st, b->spk, nspk, e->content_emb, e->codebook_emb, e->wait4start,
                       b->d_prompt_cc, b->d_prompt_ac, b->Pmax, R, d, c.num_codebooks, c.codebook_size, D, b->ax);
    // positions 0..M-1, all rows in this slot
    std::vector<int> hs(M, slot), hp(M);
    for (int i = 0; i < M; ++i) hp[i] = i;
    SVA_HIP(hipMemcpyAsync(b->d_slot, hs.data(), sizeof(int) * M, hipMemcpyHostToDevice, st));
    SVA_HIP(hipMemcpyAsync(b->d_pos, hp.data(), sizeof(int) * M, hipMemcpyHostToDevice, st));
    SVA_HIP(hipStreamSynchronize(st));       // hs/hp are stack-lifetime host buffers
    SVA_TRY(ar_layers_pass(b, e->ar_layers, M, b->d_slot, b->d_pos, e->rope_ar, (float*)b->kv_slow, b->kv_slow_layer, b->kv_slow_slot,
                           c.max_seq_len, b->ax, slot, 0));
    // cached_ref_emb = embed(ref_audio_codes)[-d:]  (:775)
    SVA_TRY(launch_audio_embed(e->codebook_emb, b->d_prompt_ac + (R - d), 1, b->Pmax, d, c.num_codebooks, c.codebook_size, D,
                               b->cached_ref_emb + (long)slot * c.max_delay * D, D, st));
    // logits / pre-norm hidden state of the last prompt token, as forward_generate returns them for a prefill
    // (dual_ar_stream.py:338-356); nobody consumes them downstream, they are taps for the parity tests ("slow_logits", "hidden")
    if (tap_logits) {       // (the initial prefill only: a re-prefill inside a stream must not overwrite the frame's taps)
        hipLaunchKernelGGL(copy_rows_kernel, dim3(1), dim3(256), 0, st, b->ax, (long)D, (long)(M - 1) * D, b->hidden + (long)slot * D, D);
        SVA_TRY(launch_rmsnorm_rows(b->hidden + (long)slot * D, D, 0, D, 1, 1, D, e->ar_norm, 1e-5f, b->ahn, D, 0, D, st));
        SVA_TRY(gemm_call(b, b->ahn, D, 0, D, 1, 1, 1, 1, 1, D, e->ar_output, b->slow_logits + (long)slot * c.ar_vocab, c.ar_vocab, 0, c.ar_vocab));
    }
    const int lp = M - 1;
    SVA_HIP(hipMemcpyAsync(b->d_last_pos + slot, &lp, sizeof(int), hipMemcpyHostToDevice, st));
    SVA_HIP(hipStreamSynchronize(st));
    b->h_last_pos[slot] = lp;
    return 0;
}

// prefill_src_condition4delay for the given slots (all slots at stream start; the re-prefilled ones later)
int ar_delay_fill(sva_batch* b, const std::vector<int>& slots) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int n = (int)slots.size(), D = c.ar_dim, d = b->p.delay, rows = 2 * d - 1;
    if (n == 0) return 0;
    hipStream_t st = b->stream;
    SVA_HIP(hipMemcpyAsync(b->d_slot_list, slots.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
    SVA_HIP(hipStreamSynchronize(st));          // `slots` may be a temporary
    hipLaunchKernelGGL(build_delayfill_kernel, dim3(n * rows), dim3(256), 0, st, e->content_emb, b->d_content_hist, b->hist_cap,
                       b->d_ncontent, b->cached_ref_emb, c.max_delay, b->d_last_pos, d, D, b->ax, b->d_slot, b->d_pos, b->cached_audio_emb,
                       b->d_slot_list);
    SVA_TRY(ar_layers_pass(b, e->ar_layers, n * rows, b->d_slot, b->d_pos, e->rope_ar, (float*)b->kv_slow, b->kv_slow_layer,
                           b->kv_slow_slot, c.max_seq_len, b->ax));
    hipLaunchKernelGGL(add_list_kernel, dim3((n + 63) / 64), dim3(64), 0, st, b->d_last_pos, b->d_slot_list, n, rows);
    for (int s_ : slots) b->h_last_pos[s_] += rows;
    SVA_HIP(hipGetLastError());
    return 0;
}
int ar_delay_fill(sva_batch* b) {
    std::vector<int> all(b->B);
    for (int i = 0; i < b->B; ++i) all[i] = i;
    return ar_delay_fill(b, all);
}

// Which HiFiGAN levels run as the fused LDS-resident kernel (voc_fused.hip).  Measured on MI355X (profiles/r02_voc_fused.txt): the
// C = 16 level takes 27-33 us fused against 7 launches / ~70 us at one stream and breaks even around 4-8 streams; from there on,
// and for C = 32 at any batch, the tap-split GEMM formulation is faster (the fused kernel recomputes an 18 (k - 1)-row halo per
// tile and runs one workgroup per CU), so the default fuses C = 16 for <= 4 streams.  SVA_DEBUG=voc_fused_mask=M (bit 0: C = 16,
// bit 1: C = 32) / sva_debug_configure override it for the parity tests.
bool voc_level_is_fused(const sva_batch* b, int C) {
    if (!b->voc_fused || !voc_level_supported(C)) return false;
    if (b->voc_fused_mask >= 0) return (b->voc_fused_mask & (C == 16 ? 1 : 2)) != 0;
    return C == 16 && b->B <= 4;
}

// ---- V: streaming vocoder on T new code frames held in d_vcodes [B][8][Tv] -------------------------------
// part 0: the whole vocoder; 1: firefly.quantizer.decode only (FSQ decode + upsampler, output = rows [pin.H, pin.H + 4T) of
// b->pin); 2: firefly.head only (HiFiGAN on those rows)
int vocode(sva_batch* b, int T, bool shift, int part) {
    sva_engine* e = b->e;
    const sva_config& c = e->cfg;
    const int B = b->B, V = c.voc_dim, G = c.num_codebooks;
    hipStream_t st = b->stream;
    SVA_CHECK(T >= 1 && T <= b->Tv, "vocode: T out of range");
    // parts 3 / 4: the FSQ decode alone / everything behind it (the pipelined stage graphs are cut where the step's codes are released)
    if (part != 2) {
    if (part != 4) {
    if (b->voc_codes) SVA_TRY(launch_fsq_decode(b->voc_codes, b->voc_codes_bstride, b->voc_codes_gstride, B, T, G, V / G, e->fsq_W, e->fsq_b, b->zq.p, b->zq.bstride, 0, V, st));
    else SVA_TRY(launch_fsq_decode(b->d_vcodes, (long)G * b->Tv, b->Tv, B, T, G, V / G, e->fsq_W, e->fsq_b, b->zq.p, b->zq.bstride, 0, V, st));
    if (b->voc_codes_event) SVA_HIP(hipEventRecord(b->voc_codes_event, st));
    }
    if (part == 3) return 0;
    // upsample.0/1: ConvTranspose k=s=2 (stateless) + ConvNeXtBlock  (fsq.py:61-74)
    SVA_TRY(gemm_call(b, b->zq.p, b->zq.bstride, 0, V, B, T, 1, 1, 1, V, e->up_conv[0], b->u0.p, b->u0.bstride, (long)b->u0.H * V, 2 * V));
    SVA_TRY(cnx_block(b, e->up_cnx[0], b->u0, 2 * T, b->vh1, b->vh2, &b->v0));
    SVA_TRY(gemm_call(b, b->v0.p, b->v0.bstride, 0, V, B, 2 * T, 1, 1, 1, V, e->up_conv[1], b->u1.p, b->u1.bstride,
                      (long)b->u1.H * V, 2 * V));
    SVA_TRY(cnx_block(b, e->up_cnx[1], b->u1, 4 * T, b->vh1, b->vh2, &b->pin));
    }
    if (part == 1) return 0;
    // conv_pre k13 (reads the upsampler output with 12 history rows) -> S[0]
    SVA_TRY(conv_act(b, b->pin, 4 * T, 1, 1, e->pre_k, e->conv_pre, b->S[0]));
    long Tl = 4L * T;
    for (int i = 0; i < 5; ++i) {
        const int s = e->ups_s[i];
        const int Cout = b->X[i].C;
        // SiLU -> ConvTranspose (k = 2s): 2-tap GEMM over rows q-1, q with N = s*Cout  (firefly.py:284-285, 131-138)
        {
            ConvGemm p;
            p.a_silu = 1;
            SVA_CHECK(b->S[i].H >= 1, "ups: history");
            SVA_TRY(gemm_call(b, b->S[i].p, b->S[i].bstride, (long)(b->S[i].H - 1) * b->S[i].C, b->S[i].C, B, (int)Tl, 1, 1, 2, b->S[i].C,
                              e->ups[i], b->X[i].p, b->X[i].bstride, (long)b->X[i].H * Cout, s * Cout, p));
        }
        Tl *= s;
        // ParallelBlock = mean of three ResBlock1 (firefly.py:183-190, 214-215).  The three branches are independent
        // chains of 6 convs: branch 0 stays on the main stream, branches 1/2 run on side streams; the last conv of each
        // branch accumulates (x 1/3) into the level output in the fixed order 0, 1, 2 (event chain => deterministic sum).
        Act& out = b->S[i + 1];
        if (voc_level_is_fused(b, Cout)) {
            // narrow levels: the whole ParallelBlock in one launch (three branches x time tiles x streams), intermediates in LDS,
            // the level input's history (18 (k - 1) rows) as the only streaming state
            const float* W[3][6]; const float* bs[3][6]; float* y3[3];
            for (int br = 0; br < 3; ++br) {
                for (int j = 0; j < 3; ++j) {
                    const ResConv& rcv = e->res[i][br][j];
                    SVA_CHECK(rcv.k == kResK[br] && rcv.dil == kResD[j], "voc_level: unexpected ResBlock geometry");
                    W[br][2 * j] = rcv.c1.W; bs[br][2 * j] = rcv.c1.b; W[br][2 * j + 1] = rcv.c2.W; bs[br][2 * j + 1] = rcv.c2.b;
                }
                y3[br] = b->y3[i][br].p;
            }
            SVA_TRY(launch_voc_level(b->X[i].p, b->X[i].bstride, b->X[i].H, Cout, B, (int)Tl, W, bs, kResD, y3, b->y3[i][0].bstride, b->d_voc_frames,
                                     b->voc_rpf[i], st));
            {   // bookkeeping as one conv-GEMM launch: algorithmic FLOPs of the 18 convs, bytes = input (+ history) + weights + output
                double kk = 0;
                for (int br = 0; br < 3; ++br) kk += 6.0 * kResK[br];
                b->gemm_flops += 2.0 * B * (double)Tl * Cout * Cout * kk;
                b->gemm_launches += 1;
                b->gemm_bytes += 4.0 * ((double)B * (Tl + b->X[i].H) * Cout + kk * Cout * Cout + 3.0 * B * (double)Tl * Cout);
            }
            const long n4 = Tl * Cout / 4;
            hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)((n4 + 255) / 256), B), dim3(256), 0, st, b->y3[i][0].p, b->y3[i][1].p, b->y3[i][2].p,
                               b->y3[i][0].bstride, out.p, out.bstride, (long)out.H * Cout, n4);
            SVA_HIP(hipGetLastError());
            continue;
        }
        if (b->voc_dma[i] == 2) {
            // narrow level (C = 16 / 32) of a large batch: every conv stage of the three branches as ONE voc_conv_kernel launch (gemm_planes.hip: the
            // tile's input rows + halo and the branch's whole weight resident in LDS), activations between the convs as ROW-MAJOR operand planes
            const int pm = b->voc_pmode;
            SVA_TRY(launch_to_planes_act(b->X[i].p, B, b->X[i].rows, b->X[i].H, (int)Tl, Cout, b->XP[i], (long)B * b->X[i].bstride, pm, 1, st, 0));
            Act* y[3] = {&b->X[i], &b->X[i], &b->X[i]};
            const unsigned short* yp[3] = {b->XP[i], b->XP[i], b->XP[i]};
            for (int j = 0; j < 3; ++j) {
                VocConvGroup g1, g2;
                g1.B = g2.B = B; g1.T = g2.T = (int)Tl; g1.ovf = g2.ovf = b->d_mm_ovf; g1.n = g2.n = 3;
                for (int br = 0; br < 3; ++br) {
                    const ResConv& rcv = e->res[i][br][j];
                    const int padL = (rcv.k - 1) * rcv.dil;
                    Act& t = b->tb[i][br][j];
                    SVA_CHECK(y[br]->H >= padL && t.H >= padL, "voc_conv: not enough history rows");
                    VocConv& a = g1.g[br];
                    a.Ap = yp[br]; a.ap_pstride = (long)B * y[br]->bstride; a.a_rows_b = y[br]->rows; a.a_row0 = y[br]->H - padL; a.a_rows_total = (long)B * y[br]->rows;
                    a.Wp = rcv.q1; a.wp_pstride = (long)Cout * rcv.Kq; a.wp_inv = rcv.q1_inv; a.bias = rcv.c1.b; a.taps = rcv.k; a.dil = rcv.dil;
                    a.Cp = b->tbP[i][br][j]; a.cp_pstride = (long)B * t.bstride; a.c_rows_b = t.rows; a.c_row0 = t.H;
                    Act& dst = j < 2 ? b->yb[i][br][j] : b->y3[i][br];
                    VocConv& c2 = g2.g[br];
                    c2.Ap = b->tbP[i][br][j]; c2.ap_pstride = (long)B * t.bstride; c2.a_rows_b = t.rows; c2.a_row0 = t.H - padL; c2.a_rows_total = (long)B * t.rows;
                    c2.Wp = rcv.q2; c2.wp_pstride = (long)Cout * rcv.Kq; c2.wp_inv = rcv.q2_inv; c2.bias = rcv.c2.b; c2.taps = rcv.k; c2.dil = rcv.dil;
                    c2.res = y[br]->p; c2.r_bstride = y[br]->bstride; c2.r_off = (long)y[br]->H * Cout;
                    c2.Cf = dst.p; c2.c_bstride = dst.bstride; c2.c_off = (long)dst.H * Cout;
                    if (j < 2) { c2.Cp = b->ybP[i][br][j]; c2.cp_pstride = (long)B * dst.bstride; c2.c_rows_b = dst.rows; c2.c_row0 = dst.H; }
                }
                SVA_TRY(voc_conv_call(b, g1, Cout, pm));
                SVA_TRY(voc_conv_call(b, g2, Cout, pm));
                for (int br = 0; br < 3 && j < 2; ++br) { y[br] = &b->yb[i][br][j]; yp[br] = b->ybP[i][br][j]; }
            }
            const long n4 = Tl * Cout / 4;
            hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)((n4 + 255) / 256), B), dim3(256), 0, st, b->y3[i][0].p, b->y3[i][1].p, b->y3[i][2].p,
                               b->y3[i][0].bstride, out.p, out.bstride, (long)out.H * Cout, n4);
            SVA_HIP(hipGetLastError());
            continue;
        }
        if (b->voc_dma[i]) {
            // wide level of a large batch: every conv on the LDS-DMA planes kernel, the three branches' tiles as one sequence per conv stage.
            // Operands are planes throughout: silu(X) from a split pass over the new rows, then each conv's epilogue writes the parts of
            // silu(output) for its consumer (c1: only those; c2: also the fp32 sum the next residual / the mean reads)
            const int pm = b->voc_pmode;
            auto planes_in = [&](ConvGemm& g, const Act& a, const unsigned short* P) {
                g.A = nullptr; g.a_silu = 0;
                g.Ap = P; g.ap_pstride = (long)B * a.bstride; g.ap_rows = (long)B * a.rows;
            };
            auto planes_out = [&](ConvGemm& g, const Act& a, unsigned short* P) {
                g.Cp = P; g.cp_pstride = (long)B * a.bstride; g.cp_rows = (long)B * a.rows; g.cp_silu = 1;
            };
            SVA_TRY(launch_to_planes_act(b->X[i].p, B, b->X[i].rows, b->X[i].H, (int)Tl, Cout, b->XP[i], (long)B * b->X[i].bstride, pm, 1, st));
            Act* y[3] = {&b->X[i], &b->X[i], &b->X[i]};
            const unsigned short* yp[3] = {b->XP[i], b->XP[i], b->XP[i]};
            for (int j = 0; j < 3; ++j) {
                ConvGemm g1[3], g2[3];
                for (int br = 0; br < 3; ++br) {
                    const ResConv& rcv = e->res[i][br][j];
                    Act& t = b->tb[i][br][j];
                    SVA_TRY(conv_desc(b, *y[br], (int)Tl, rcv.dil, rcv.k, rcv.c1, t, g1[br]));
                    planes_in(g1[br], *y[br], yp[br]);
                    planes_out(g1[br], t, b->tbP[i][br][j]);
                    g1[br].C = nullptr;
                    Act& dst = j < 2 ? b->yb[i][br][j] : b->y3[i][br];
                    g2[br].res = y[br]->p; g2[br].r_bstride = y[br]->bstride; g2[br].r_off = (long)y[br]->H * Cout; g2[br].ldr = Cout;
                    SVA_TRY(conv_desc(b, t, (int)Tl, rcv.dil, rcv.k, rcv.c2, dst, g2[br]));
                    planes_in(g2[br], t, b->tbP[i][br][j]);
                    if (j < 2) planes_out(g2[br], dst, b->ybP[i][br][j]);
                    g1[br].cu_limit = g2[br].cu_limit = b->enc_cus;
                }
                SVA_TRY(gemm_group_call(b, g1, 3));
                SVA_TRY(gemm_group_call(b, g2, 3));
                for (int br = 0; br < 3 && j < 2; ++br) { y[br] = &b->yb[i][br][j]; yp[br] = b->ybP[i][br][j]; }
            }
            const long n4 = Tl * Cout / 4;
            hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)((n4 + 255) / 256), B), dim3(256), 0, st, b->y3[i][0].p, b->y3[i][1].p, b->y3[i][2].p,
                               b->y3[i][0].bstride, out.p, out.bstride, (long)out.H * Cout, n4);
            SVA_HIP(hipGetLastError());
            continue;
        }
        if (b->voc_grouped) {
            // one launch per conv stage for the three branches (same M, N, Cin; k = 3 / 7 / 11 taps): 12 launches + the
            // mean per level instead of 18 on three streams -- at small B the step is bound by the number of kernels
            Act* y[3] = {&b->X[i], &b->X[i], &b->X[i]};
            for (int j = 0; j < 3; ++j) {
                ConvGemm g1[3], g2[3];
                for (int br = 0; br < 3; ++br) {
                    const ResConv& rcv = e->res[i][br][j];
                    g1[br].a_silu = 1;
                    SVA_TRY(conv_desc(b, *y[br], (int)Tl, rcv.dil, rcv.k, rcv.c1, b->tb[i][br][j], g1[br]));
                    Act& dst = j < 2 ? b->yb[i][br][j] : b->y3[i][br];
                    g2[br].a_silu = 1;
                    g2[br].res = y[br]->p; g2[br].r_bstride = y[br]->bstride; g2[br].r_off = (long)y[br]->H * Cout; g2[br].ldr = Cout;
                    SVA_TRY(conv_desc(b, b->tb[i][br][j], (int)Tl, rcv.dil, rcv.k, rcv.c2, dst, g2[br]));
                }
                SVA_TRY(gemm_group_call(b, g1, 3));
                SVA_TRY(gemm_group_call(b, g2, 3));
                for (int br = 0; br < 3; ++br) y[br] = j < 2 ? &b->yb[i][br][j] : &b->y3[i][br];
            }
            const long n4 = Tl * Cout / 4;
            hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)((n4 + 255) / 256), B), dim3(256), 0, st, b->y3[i][0].p, b->y3[i][1].p, b->y3[i][2].p,
                               b->y3[i][0].bstride, out.p, out.bstride, (long)out.H * Cout, n4);
            SVA_HIP(hipGetLastError());
            continue;
        }
        const bool par = b->concurrency;
        if (par) {
            SVA_TRY(stream_fork(b, st, b->aux[0]));
            SVA_TRY(stream_fork(b, st, b->aux[1]));
        }
        hipStream_t prev_last = nullptr;
        for (int br = 0; br < 3; ++br) {
            hipStream_t sbr = (par && br > 0) ? b->aux[br - 1] : st;
            b->stream = sbr;
            Act* y = &b->X[i];
            int rc = 0;
            for (int j = 0; j < 3 && !rc; ++j) {
                const ResConv& rcv = e->res[i][br][j];
                ConvGemm p1;
                p1.a_silu = 1;
                rc = conv_act(b, *y, (int)Tl, 1, rcv.dil, rcv.k, rcv.c1, b->tb[i][br][j], p1);
                if (rc) break;
                ConvGemm p2;
                p2.a_silu = 1;
                p2.res = y->p; p2.r_bstride = y->bstride; p2.r_off = (long)y->H * Cout; p2.ldr = Cout;
                if (j < 2) {
                    rc = conv_act(b, b->tb[i][br][j], (int)Tl, 1, rcv.dil, rcv.k, rcv.c2, b->yb[i][br][j], p2);
                    y = &b->yb[i][br][j];
                } else {
                    p2.scale = 1.0f / 3.0f;
                    p2.accumulate = br > 0;
                    if (par && br > 0) {
                        hipError_t he = hipSuccess;
                        hipEvent_t ev = next_event(b);
                        he = hipEventRecord(ev, prev_last);
                        if (he == hipSuccess) he = hipStreamWaitEvent(sbr, ev, 0);
                        if (he != hipSuccess) { b->stream = st; SVA_HIP(he); }
                    }
                    rc = conv_act(b, b->tb[i][br][j], (int)Tl, 1, rcv.dil, rcv.k, rcv.c2, out, p2);
                }
            }
            b->stream = st;
            if (rc) return rc;
            prev_last = sbr;
        }
        if (par) SVA_TRY(stream_fork(b, b->aux[1], st));      // join: branch 2's last conv is ordered after 0 and 1
    }
    SVA_TRY(launch_conv_post_tanh(b->S[5].p, b->S[5].bstride, (long)(b->S[5].H - (e->post_k - 1)) * b->S[5].C, B, (int)Tl, b->S[5].C, e->post_k,
                                  e->post_w, e->post_b, b->pcm_dst ? b->pcm_dst : b->d_pcm, b->pcm_dst ? b->pcm_dst_bstride : 2048L * b->Tv, 0, st));
    if (b->pcm_dst) b->pcm_direct_done = true;
    if (shift) {
        // update T in the descriptors if it changed (host table re-uploaded; rare)
        bool dirty = false;
        for (auto& d : b->shift_host) {
            const int t = (int)((long)T * d.pad);      // pad holds rows-per-code-frame of this tensor
            if (d.T != t) { d.T = t; dirty = true; }
        }
        if (dirty) {
            SVA_HIP(hipMemcpyAsync(b->d_shift, b->shift_host.data(), sizeof(ShiftDesc) * b->shift_host.size(), hipMemcpyHostToDevice, st));
            SVA_HIP(hipStreamSynchronize(st));
        }
        SVA_TRY(launch_shift_history(b->d_shift, (int)b->shift_host.size(), B, st, b->B <= 8 ? 4 : 1));
        SVA_TRY(launch_add_i32(b->d_voc_frames, T, st));      // (the fused levels ask whether their halo lies inside the stream)
    }
    return 0;
}

int register_shift(sva_batch* b, Act& a, int rows_per_frame) {
    if (a.H == 0) return 0;
    ShiftDesc d;
    d.ptr = a.p; d.bstride = a.bstride; d.H = a.H; d.T = 0; d.C = a.C; d.pad = rows_per_frame;
    b->shift_host.push_back(d);
    return 0;
}

}  // namespace sva
