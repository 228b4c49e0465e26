// Engine lifecycle and weight packing of the sva engine: configuration defaults, sva_engine_create / load_weight / destroy, the Packer
// (reference state-dict tensors -> the [N][K] tap-major matrices, folded weight norm, interleaved SwiGLU rows, fp16 copies, the persistent
// decode kernels' layouts) and sva_engine_finalize (incl. the pre-split operand planes of gemm_planes.hip).  Reference: the module
// constructors the checkpoints are loaded into (modules/vqgan/modules/firefly.py, modules/dual_ar_stream.py, modules/bsq.py).
#include "engine_internal.h"

using namespace sva;

extern "C" int sva_config_default(sva_config* c) {
    if (!c) return -1;
    memset(c, 0, sizeof(*c));
    c->n_mels = 160;
    int dep[4] = {3, 3, 9, 3}, dims[4] = {128, 256, 384, 512};
    for (int i = 0; i < 4; ++i) { c->enc_depths[i] = dep[i]; c->enc_dims[i] = dims[i]; }
    c->tr_layers = 8; c->tr_heads = 8; c->tr_dim = 512; c->tr_inter = 1536; c->bsq_bits = 13;
    c->ar_dim = 768; c->ar_heads = 12; c->ar_layers = 12; c->ar_fast_layers = 4; c->ar_inter = 2304;
    c->ar_vocab = 8192; c->codebook_size = 1000; c->num_codebooks = 8; c->max_delay = 8; c->max_seq_len = 2048;
    c->timbre_dim = 128; c->timbre_tokens = 32; c->style_dim = 192; c->voc_dim = 512; c->ar_dtype = 0; c->mm_mode = 1; c->voc_dtype = 0;
    return 0;
}
extern "C" int sva_stream_params_default(sva_stream_params* p) {
    if (!p) return -1;
    memset(p, 0, sizeof(*p));
    p->n_streams = 1; p->encode_window_frames = 128; p->decode_window_frames = 64; p->chunk_frames = 1;
    p->delay = 2; p->max_seq_frames = 768; p->buffer_frames = 32; p->max_prompt_frames = 256;
    p->temperature = 0.7f; p->top_p = 0.7f; p->voc_max_frames = 1; p->use_graph = 0; p->skip_semantic = 0;
    return 0;
}

// ============================================================================================
// engine: weights
// ============================================================================================
extern "C" int sva_engine_create(const sva_config* cfg, int device, sva_engine** out) {
    SVA_CHECK(cfg && out, "null argument");
    SVA_CHECK(cfg->tr_dim == cfg->enc_dims[3] && cfg->voc_dim == 512, "unsupported dims");
    SVA_CHECK(cfg->ar_dim / cfg->ar_heads == 64 && cfg->tr_dim / cfg->tr_heads == 64, "head_dim must be 64");
    SVA_CHECK(cfg->ar_dtype == 0 || cfg->ar_dtype == 1, "ar_dtype must be 0 (fp32 weights / fp32 KV) or 1 (fp16 weights / fp16 slow KV)");
    int ndev = 0;
    SVA_HIP(hipGetDeviceCount(&ndev));
    SVA_CHECK(ndev > 0 && device < ndev, "no such HIP device (the product path has no CPU fallback)");
    SVA_HIP(hipSetDevice(device));
    sva_engine* e = new sva_engine();
    e->cfg = *cfg;
    e->device = device;
    *out = e;
    return 0;
}

extern "C" int sva_engine_load_weight(sva_engine* e, const char* name, int ndim, const int64_t* shape, const float* data) {
    SVA_CHECK(e && name && data, "null argument");
    SVA_CHECK(!e->finalized, "engine already finalized");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data, data + t.numel());
    e->host[name] = std::move(t);
    return 0;
}

extern "C" void sva_engine_destroy(sva_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    for (void* p : e->allocs.chunks) hipFree(p);
    if (e->mega_ev) (void)hipEventDestroy(e->mega_ev);
    if (e->ops_stream) (void)hipStreamDestroy(e->ops_stream);
    delete e;
}

namespace {

struct Packer {
    sva_engine* e;
    std::string err;

    const HostTensor* find(const std::string& n) {
        auto it = e->host.find(n);
        if (it != e->host.end()) return &it->second;
        return nullptr;
    }
    // ar_dtype = 1: the AR's matrices ("arvc." Linear weights) hold fp16 values, as under the reference's
    // torch.autocast(fp16) decode (evaluations/infer_arvc.py:483, 493); the fp32 copies used by the batched / prefill GEMMs carry
    // the same rounded values, so every path computes with one set of numbers
    static float round_half(float v) { return (float)(_Float16)v; }
    int upload_half(void** out, const std::vector<float>& v) {
        std::vector<uint16_t> hbits(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const _Float16 hv = (_Float16)v[i];
            memcpy(&hbits[i], &hv, 2);
        }
        uint16_t* d = nullptr;
        SVA_TRY(dev_alloc(e->allocs, &d, hbits.size(), false));
        SVA_HIP(hipMemcpy(d, hbits.data(), hbits.size() * 2, hipMemcpyHostToDevice));
        *out = d;
        return 0;
    }
    // plain weight or folded weight-norm pair (firefly.py:105-111, 295-301: w = g * v / ||v||, norm over dims 1..)
    bool weight(const std::string& prefix, HostTensor& out) {
        if (const HostTensor* t = find(prefix + ".weight")) {
            out = *t;
            if (e->cfg.ar_dtype == 1 && prefix.compare(0, 5, "arvc.") == 0)
                for (auto& v : out.data) v = round_half(v);
            return true;
        }
        const HostTensor* g = find(prefix + ".parametrizations.weight.original0");
        const HostTensor* v = find(prefix + ".parametrizations.weight.original1");
        if (!g || !v) { err = "missing weight " + prefix + ".weight"; return false; }
        out = *v;
        const long rows = v->shape[0], inner = v->numel() / rows;
        for (long r = 0; r < rows; ++r) {
            double s = 0;
            for (long i = 0; i < inner; ++i) s += (double)v->data[r * inner + i] * v->data[r * inner + i];
            const float sc = (float)(g->data[r] / sqrt(s));
            for (long i = 0; i < inner; ++i) out.data[r * inner + i] = v->data[r * inner + i] * sc;
        }
        return true;
    }
    int vec(const std::string& n, float** out, long expect) {
        const HostTensor* t = find(n);
        SVA_CHECK(t, ("missing tensor " + n).c_str());
        SVA_CHECK(t->numel() == expect, ("bad size for " + n).c_str());
        return upload(e->allocs, out, t->data);
    }
    int bias_of(const std::string& prefix, Lin& l) {
        if (const HostTensor* t = find(prefix + ".bias")) {
            SVA_CHECK(t->numel() == l.N, ("bad bias size " + prefix).c_str());
            return upload(e->allocs, &l.b, t->data);
        }
        l.b = nullptr;
        return 0;
    }
    // fragment-major copy of an uploaded [N][K] matrix for the weight-streaming GEMM (gemm_stream.hip): tile t of 16 rows, block kb of 16 k,
    // lane l = (row & 15) + 16 * (k4): four consecutive k -- what lane l feeds to the four MFMA steps of the block; rows beyond N are zeros
    int frag(Lin& l, const std::vector<float>& w) {
        if (l.K % 16 != 0 || l.N < 16) return 0;
        const long nkb = l.K / 16, nt16 = (l.N + 15) / 16;
        std::vector<float> p((size_t)nt16 * nkb * 256, 0.f);
        for (long t = 0; t < nt16; ++t)
            for (long kb = 0; kb < nkb; ++kb)
                for (int ln = 0; ln < 64; ++ln) {
                    const long n = t * 16 + (ln & 15);
                    if (n >= l.N) continue;
                    memcpy(&p[((t * nkb + kb) * 64 + ln) * 4], &w[(size_t)n * l.K + kb * 16 + 4 * (ln >> 4)], 4 * sizeof(float));
                }
        return upload(e->allocs, &l.Wk, p);
    }
    // nn.Linear [N, K]
    int linear(const std::string& prefix, Lin& l, int N, int K) {
        HostTensor w;
        SVA_CHECK(weight(prefix, w), err.c_str());
        SVA_CHECK(w.numel() == (long)N * K, ("bad shape " + prefix).c_str());
        l.N = N; l.K = K;
        SVA_TRY(upload(e->allocs, &l.W, w.data));
        SVA_TRY(frag(l, w.data));
        return bias_of(prefix, l);
    }
    // nn.Conv1d weight [Cout, Cin, k] -> [Cout][k][Cin]
    int conv(const std::string& prefix, Lin& l, int Cout, int Cin, int k) {
        HostTensor w;
        SVA_CHECK(weight(prefix, w), err.c_str());
        SVA_CHECK(w.numel() == (long)Cout * Cin * k, ("bad shape " + prefix).c_str());
        std::vector<float> p((size_t)Cout * k * Cin);
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < Cin; ++i)
                for (int j = 0; j < k; ++j) p[((size_t)o * k + j) * Cin + i] = w.data[((size_t)o * Cin + i) * k + j];
        l.N = Cout; l.K = k * Cin;
        SVA_TRY(upload(e->allocs, &l.W, p));
        SVA_TRY(frag(l, p));
        return bias_of(prefix, l);
    }
    // nn.ConvTranspose1d weight [Cin, Cout, k], stride s, k == 2s (FishTransConvNet, firefly.py:114-138):
    //   y[q*s + r, co] = b[co] + sum_ci x[q, ci] W[ci, co, r] + sum_ci x[q-1, ci] W[ci, co, r + s]
    // packed as a 2-tap GEMM with N = s*Cout: row n = r*Cout + co, tap 0 (x[q-1]) = W[:, co, r+s], tap 1 (x[q]) = W[:, co, r]
    // k == s: 1 tap, row n = r*Cout + co = W[:, co, r]
    int conv_t(const std::string& prefix, Lin& l, int Cin, int Cout, int k, int s) {
        HostTensor w;
        SVA_CHECK(weight(prefix, w), err.c_str());
        SVA_CHECK(w.numel() == (long)Cin * Cout * k, ("bad shape " + prefix).c_str());
        SVA_CHECK(k == 2 * s || k == s, "conv_t: kernel must be stride or 2*stride");
        const int taps = k / s;
        std::vector<float> p((size_t)s * Cout * taps * Cin);
        for (int r = 0; r < s; ++r)
            for (int co = 0; co < Cout; ++co)
                for (int tap = 0; tap < taps; ++tap)
                    for (int ci = 0; ci < Cin; ++ci) {
                        const int kk = (taps == 2) ? (tap == 0 ? r + s : r) : r;
                        p[(((size_t)r * Cout + co) * taps + tap) * Cin + ci] = w.data[((size_t)ci * Cout + co) * k + kk];
                    }
        l.N = s * Cout; l.K = taps * Cin;
        SVA_TRY(upload(e->allocs, &l.W, p));
        SVA_TRY(frag(l, p));
        const HostTensor* b = find(prefix + ".bias");
        SVA_CHECK(b && b->numel() == Cout, ("missing bias " + prefix).c_str());
        std::vector<float> bb((size_t)s * Cout);
        for (int r = 0; r < s; ++r)
            for (int co = 0; co < Cout; ++co) bb[(size_t)r * Cout + co] = b->data[co];
        return upload(e->allocs, &l.b, bb);
    }
    int cnx(const std::string& p, CNX& c, int C) {
        c.C = C;
        const HostTensor* dw = find(p + "dwconv.conv.weight");
        SVA_CHECK(dw && dw->numel() == (long)C * 7, ("missing " + p + "dwconv").c_str());
        std::vector<float> t((size_t)7 * C);
        for (int ch = 0; ch < C; ++ch)
            for (int j = 0; j < 7; ++j) t[(size_t)j * C + ch] = dw->data[(size_t)ch * 7 + j];
        SVA_TRY(upload(e->allocs, &c.dwT, t));
        SVA_TRY(vec(p + "dwconv.conv.bias", &c.dwb, C));
        SVA_TRY(vec(p + "norm.weight", &c.lnw, C));
        SVA_TRY(vec(p + "norm.bias", &c.lnb, C));
        SVA_TRY(vec(p + "gamma", &c.gamma, C));
        SVA_TRY(linear(p + "pwconv1", c.pw1, 4 * C, C));
        SVA_TRY(linear(p + "pwconv2", c.pw2, C, 4 * C));
        return 0;
    }
    // w1 / w3 rows interleaved in groups of 16 -> [2*I][D]
    int w13(const std::string& p, Lin& l, int I, int D, void** mega_w13 = nullptr) {
        HostTensor w1, w3;
        SVA_CHECK(weight(p + "feed_forward.w1", w1), err.c_str());
        SVA_CHECK(weight(p + "feed_forward.w3", w3), err.c_str());
        SVA_CHECK(w1.numel() == (long)I * D && w3.numel() == (long)I * D && I % 16 == 0, ("bad ffn shape " + p).c_str());
        std::vector<float> out((size_t)2 * I * D);
        for (int g = 0; g < I / 16; ++g)
            for (int r = 0; r < 16; ++r) {
                memcpy(&out[((size_t)g * 32 + r) * D], &w1.data[((size_t)g * 16 + r) * D], sizeof(float) * D);
                memcpy(&out[((size_t)g * 32 + 16 + r) * D], &w3.data[((size_t)g * 16 + r) * D], sizeof(float) * D);
            }
        l.N = 2 * I; l.K = D; l.b = nullptr;
        if (mega_w13) {
            // the persistent decode kernel's row order: wave w owns rows [12w, 12w + 12) = w1 rows 6w..6w+5, w3 rows 6w..6w+5
            SVA_CHECK(I % 6 == 0, "ffn size must be a multiple of 6");
            std::vector<float> mp((size_t)2 * I * D);
            for (int w = 0; w < I / 6; ++w)
                for (int r = 0; r < 6; ++r) {
                    memcpy(&mp[((size_t)w * 12 + r) * D], &w1.data[((size_t)w * 6 + r) * D], sizeof(float) * D);
                    memcpy(&mp[((size_t)w * 12 + 6 + r) * D], &w3.data[((size_t)w * 6 + r) * D], sizeof(float) * D);
                }
            if (e->cfg.ar_dtype == 1) SVA_TRY(upload_half(mega_w13, mp));
            else { float* f = nullptr; SVA_TRY(upload(e->allocs, &f, mp)); *mega_w13 = f; }
        }
        if (e->cfg.ar_dtype == 1 && mega_w13) SVA_TRY(upload_half(&l.Wh, out));      // the batched chain's interleaved layout in fp16
        SVA_TRY(frag(l, out));
        return upload(e->allocs, &l.W, out);
    }
    // fp16 copy of an already uploaded [N][K] matrix's host values (ar_dtype = 1) or the fp32 device pointer itself
    int mega_copy(const std::string& prefix, const Lin& l, void** out) {
        if (e->cfg.ar_dtype != 1) { *out = l.W; return 0; }
        HostTensor w;
        SVA_CHECK(weight(prefix, w), err.c_str());
        return upload_half(out, w.data);
    }
    int llama(const std::string& p, TrLayer& L, int D, int I, bool layerscale, bool mega = false) {
        SVA_TRY(vec(p + "attention_norm.weight", &L.attn_norm, D));
        SVA_TRY(vec(p + "ffn_norm.weight", &L.ffn_norm, D));
        SVA_TRY(linear(p + "attention.wqkv", L.wqkv, 3 * D, D));
        SVA_TRY(linear(p + "attention.wo", L.wo, D, D));
        SVA_TRY(w13(p, L.w13, I, D, mega ? &L.m_w13 : nullptr));
        SVA_TRY(linear(p + "feed_forward.w2", L.w2, D, I));
        if (mega) {
            SVA_TRY(mega_copy(p + "attention.wqkv", L.wqkv, &L.m_wqkv));
            SVA_TRY(mega_copy(p + "attention.wo", L.wo, &L.m_wo));
            SVA_TRY(mega_copy(p + "feed_forward.w2", L.w2, &L.m_w2));
            if (e->cfg.ar_dtype == 1) { L.wqkv.Wh = L.m_wqkv; L.wo.Wh = L.m_wo; L.w2.Wh = L.m_w2; }      // the same fp16 matrices feed the batched chain
        }
        if (layerscale) {
            SVA_TRY(vec(p + "attention_layer_scale.gamma", &L.ls_attn, D));
            SVA_TRY(vec(p + "ffn_layer_scale.gamma", &L.ls_ffn, D));
        }
        return 0;
    }
    // precompute_freqs_cis (dual_ar_stream.py:993-1001 / windowed_transformer.py:356-365): cos/sin rounded to
    // bf16.  Normally supplied by the host mirror (computed with torch, bit-identical to the reference); this
    // fallback evaluates the same formula here.
    int rope(const std::string& name, float** out, int L, int hd) {
        if (const HostTensor* t = find(name)) {
            SVA_CHECK(t->numel() == (long)L * hd, ("bad rope table " + name).c_str());
            return upload(e->allocs, out, t->data);
        }
        std::vector<float> tab((size_t)L * hd);
        for (int t = 0; t < L; ++t)
            for (int j = 0; j < hd / 2; ++j) {
                const float freq = 1.0f / powf(10000.f, (float)(2 * j) / (float)hd);
                const float ang = (float)t * freq;
                float cs[2] = {(float)cos((double)ang), (float)sin((double)ang)};
                for (int q = 0; q < 2; ++q) {           // round-to-nearest-even to bf16
                    uint32_t u;
                    memcpy(&u, &cs[q], 4);
                    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
                    memcpy(&cs[q], &u, 4);
                    tab[((size_t)t * (hd / 2) + j) * 2 + q] = cs[q];
                }
            }
        return upload(e->allocs, out, tab);
    }
};

}  // namespace

extern "C" int sva_engine_finalize(sva_engine* e) {
    SVA_CHECK(e && !e->finalized, "bad engine");
    SVA_HIP(hipSetDevice(e->device));
    (void)hipGetLastError();       // drop a stale error of an unchecked teardown call (hipFree / hip*Destroy) of an earlier handle
    const sva_config& c = e->cfg;
    Packer P{e, ""};
    // ConvNeXt encoder + 2x (conv k2 s2 + ConvNeXt): shared shape of the tokenizer front-end and of the vocoder's
    // own encoder (firefly.encode of the prompt)
    auto load_front = [&](EncFront& F, const std::string& bb, const std::string& qd) -> int {
        SVA_TRY(P.conv(bb + "downsample_layers.0.0.conv", F.stem, c.enc_dims[0], c.n_mels, 7));
        SVA_TRY(P.vec(bb + "downsample_layers.0.1.weight", &F.stem_lnw, c.enc_dims[0]));
        SVA_TRY(P.vec(bb + "downsample_layers.0.1.bias", &F.stem_lnb, c.enc_dims[0]));
        F.stages.resize(4);
        for (int i = 0; i < 4; ++i) {
            if (i > 0) {
                const std::string d = bb + "downsample_layers." + std::to_string(i) + ".";
                SVA_TRY(P.vec(d + "0.weight", &F.trans_lnw[i], c.enc_dims[i - 1]));
                SVA_TRY(P.vec(d + "0.bias", &F.trans_lnb[i], c.enc_dims[i - 1]));
                SVA_TRY(P.conv(d + "1", F.trans[i], c.enc_dims[i], c.enc_dims[i - 1], 1));
            }
            F.stages[i].resize(c.enc_depths[i]);
            for (int j = 0; j < c.enc_depths[i]; ++j)
                SVA_TRY(P.cnx(bb + "stages." + std::to_string(i) + "." + std::to_string(j) + ".", F.stages[i][j], c.enc_dims[i]));
        }
        SVA_TRY(P.vec(bb + "norm.weight", &F.final_lnw, c.enc_dims[3]));
        SVA_TRY(P.vec(bb + "norm.bias", &F.final_lnb, c.enc_dims[3]));
        const int D = c.tr_dim;
        for (int i = 0; i < 2; ++i) {
            const std::string d = qd + std::to_string(i) + ".";
            SVA_TRY(P.conv(d + "0.conv", F.ds_conv[i], D, D, 2));
            SVA_TRY(P.cnx(d + "1.", F.ds_cnx[i], D));
        }
        F.loaded = true;
        return 0;
    };
    // ---- encoder ----
    {
        // mel filterbank [1025][160] -> W [160][1088] (K padded with zeros)
        const HostTensor* fb = P.find("tok.spec_transform.fb");
        SVA_CHECK(fb && fb->numel() == 1025L * c.n_mels, "missing tok.spec_transform.fb [1025, n_mels] (host mirror supplies it)");
        std::vector<float> w((size_t)c.n_mels * 1088, 0.f);
        for (int f = 0; f < 1025; ++f)
            for (int m = 0; m < c.n_mels; ++m) w[(size_t)m * 1088 + f] = fb->data[(size_t)f * c.n_mels + m];
        e->mel_fb.N = c.n_mels; e->mel_fb.K = 1088; e->mel_fb.b = nullptr;
        SVA_TRY(upload(e->allocs, &e->mel_fb.W, w));
        std::vector<float> hann(2048);
        for (int i = 0; i < 2048; ++i) hann[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 2048.0));   // torch.hann_window (periodic)
        if (const HostTensor* hw = P.find("tok.spec_transform.spectrogram.window")) {
            SVA_CHECK(hw->numel() == 2048, "bad window");
            hann = hw->data;
        }
        SVA_TRY(upload(e->allocs, &e->hann, hann));
        std::vector<float> tw(2048);
        for (int k = 0; k < 1024; ++k) {
            tw[2 * k] = (float)cos(2.0 * M_PI * k / 2048.0);
            tw[2 * k + 1] = (float)(-sin(2.0 * M_PI * k / 2048.0));
        }
        float* twp;
        SVA_TRY(upload(e->allocs, &twp, tw));
        e->twiddle = (float2*)twp;
        SVA_TRY(load_front(e->tokf, "tok.backbone.", "tok.quantizer.downsample."));
        const int D = c.tr_dim;
        e->tr.resize(c.tr_layers);
        for (int l = 0; l < c.tr_layers; ++l)
            SVA_TRY(P.llama("tok.quantizer.pre_module.layers." + std::to_string(l) + ".", e->tr[l], D, c.tr_inter, true));
        SVA_TRY(P.vec("tok.quantizer.pre_module.norm.weight", &e->tr_norm, D));
        SVA_TRY(P.rope("tok.quantizer.pre_module.freqs_cis", &e->rope_enc, 2048, 64));
        SVA_TRY(P.vec("tok.quantizer.residual_bsq.rvqs.0.project_in.weight", &e->bsq_W, (long)c.bsq_bits * D));
        SVA_TRY(P.vec("tok.quantizer.residual_bsq.rvqs.0.project_in.bias", &e->bsq_b, c.bsq_bits));
    }
    // ---- AR ----
    {
        const int D = c.ar_dim;
        const std::string m = "arvc.decoder.model.";
        {   // content embedding followed by the wait4end rows: offline generate() feeds wait4end_j where a content token is due
            // (dual_ar_stream.py:716), addressed as code = vocab + j
            const HostTensor* ce = P.find("arvc.embedding.weight");
            const HostTensor* we = P.find("arvc.decoder.wait4end_embedding.weight");
            SVA_CHECK(ce && ce->numel() == (long)c.ar_vocab * D, "missing arvc.embedding.weight");
            std::vector<float> ext((size_t)(c.ar_vocab + c.max_delay) * D, 0.f);
            memcpy(ext.data(), ce->data.data(), sizeof(float) * (size_t)c.ar_vocab * D);
            if (we && we->numel() == (long)c.max_delay * D) memcpy(ext.data() + (size_t)c.ar_vocab * D, we->data.data(), sizeof(float) * (size_t)c.max_delay * D);
            SVA_TRY(upload(e->allocs, &e->content_emb, ext));
        }
        SVA_TRY(P.vec(m + "codebook_embeddings.weight", &e->codebook_emb, (long)c.codebook_size * c.num_codebooks * D));
        SVA_TRY(P.vec(m + "fast_embeddings.weight", &e->fast_emb, (long)c.codebook_size * D));
        SVA_TRY(P.vec("arvc.decoder.wait4start_embedding.weight", &e->wait4start, (long)c.max_delay * D));
        e->ar_layers.resize(c.ar_layers);
        // the persistent batch-1 decode kernel (ar_decode.hip) is built for the reference's sizes
        e->mega_ok = c.ar_layers == AR_SLOW_LAYERS && c.ar_fast_layers == AR_FAST_LAYERS && D == 768 && c.ar_inter == 2304 && c.ar_heads == 12 &&
                     c.num_codebooks == 8 && c.ar_vocab <= 22 * AR_WAVES && c.codebook_size <= 1024 && c.codebook_size % 2 == 0;
        const bool mg = e->mega_ok;
        for (int l = 0; l < c.ar_layers; ++l) SVA_TRY(P.llama(m + "layers." + std::to_string(l) + ".", e->ar_layers[l], D, c.ar_inter, false, mg));
        e->ar_fast_layers.resize(c.ar_fast_layers);
        for (int l = 0; l < c.ar_fast_layers; ++l)
            SVA_TRY(P.llama(m + "fast_layers." + std::to_string(l) + ".", e->ar_fast_layers[l], D, c.ar_inter, false, mg));
        SVA_TRY(P.vec(m + "norm.weight", &e->ar_norm, D));
        SVA_TRY(P.vec(m + "fast_norm.weight", &e->ar_fast_norm, D));
        SVA_TRY(P.linear(m + "output", e->ar_output, c.ar_vocab, D));
        SVA_TRY(P.linear(m + "fast_output", e->ar_fast_output, c.codebook_size, D));
        if (mg) {
            SVA_TRY(P.mega_copy(m + "output", e->ar_output, &e->m_output));
            SVA_TRY(P.mega_copy(m + "fast_output", e->ar_fast_output, &e->m_fast_output));
            if (c.ar_dtype == 1) { e->ar_output.Wh = e->m_output; e->ar_fast_output.Wh = e->m_fast_output; }
        }
        SVA_TRY(P.linear("arvc.context_in", e->context_in, D, c.timbre_dim));
        SVA_TRY(P.linear("arvc.style_in", e->style_in, D, c.style_dim));
        SVA_TRY(P.rope(m + "freqs_cis", &e->rope_ar, c.max_seq_len, 64));
        SVA_TRY(P.rope(m + "fast_freqs_cis", &e->rope_fast, c.num_codebooks, 64));
    }
    // ---- vocoder ----
    {
        const int V = c.voc_dim, G = c.num_codebooks, gd = V / G;
        std::vector<float> fw((size_t)G * gd * 4), fb((size_t)G * gd);
        for (int g = 0; g < G; ++g) {
            const std::string p = "voc.quantizer.residual_fsq.rvqs." + std::to_string(g) + ".project_out";
            const HostTensor* w = P.find(p + ".weight");
            const HostTensor* b = P.find(p + ".bias");
            SVA_CHECK(w && b && w->numel() == (long)gd * 4 && b->numel() == gd, ("missing " + p).c_str());
            memcpy(&fw[(size_t)g * gd * 4], w->data.data(), sizeof(float) * gd * 4);
            memcpy(&fb[(size_t)g * gd], b->data.data(), sizeof(float) * gd);
        }
        SVA_TRY(upload(e->allocs, &e->fsq_W, fw));
        SVA_TRY(upload(e->allocs, &e->fsq_b, fb));
        // prompt path (firefly.encode): optional -- a streaming-only deployment does not ship these tensors
        if (P.find("voc.backbone.downsample_layers.0.0.conv.weight")) {
            SVA_TRY(load_front(e->vocf, "voc.backbone.", "voc.quantizer.downsample."));
            std::vector<float> iw((size_t)G * 4 * gd), ib((size_t)G * 4);
            for (int g = 0; g < G; ++g) {
                const std::string p = "voc.quantizer.residual_fsq.rvqs." + std::to_string(g) + ".project_in";
                const HostTensor* w = P.find(p + ".weight");
                const HostTensor* b = P.find(p + ".bias");
                SVA_CHECK(w && b && w->numel() == (long)gd * 4 && b->numel() == 4, ("missing " + p).c_str());
                memcpy(&iw[(size_t)g * gd * 4], w->data.data(), sizeof(float) * gd * 4);
                memcpy(&ib[(size_t)g * 4], b->data.data(), sizeof(float) * 4);
            }
            SVA_TRY(upload(e->allocs, &e->fsq_in_W, iw));
            SVA_TRY(upload(e->allocs, &e->fsq_in_b, ib));
        }
        for (int i = 0; i < 2; ++i) {
            const std::string u = "voc.quantizer.upsample." + std::to_string(i) + ".";
            SVA_TRY(P.conv_t(u + "0.conv", e->up_conv[i], V, V, 2, 2));
            SVA_TRY(P.cnx(u + "1.", e->up_cnx[i], V));
        }
        const std::string h = "voc.head.";
        SVA_TRY(P.conv(h + "conv_pre.conv", e->conv_pre, V, V, e->pre_k));
        int ch = V;
        const int rk[3] = {3, 7, 11}, rd[3] = {1, 3, 5};
        for (int i = 0; i < 5; ++i) {
            SVA_TRY(P.conv_t(h + "ups." + std::to_string(i) + ".conv", e->ups[i], ch, ch / 2, e->ups_k[i], e->ups_s[i]));
            ch /= 2;
            for (int b = 0; b < 3; ++b)
                for (int j = 0; j < 3; ++j) {
                    const std::string q = h + "resblocks." + std::to_string(i) + ".blocks." + std::to_string(b) + ".";
                    ResConv& rc = e->res[i][b][j];
                    rc.k = rk[b];
                    rc.dil = rd[j];   // convs1 AND convs2 carry dilation d_j (firefly.py:153-180)
                    SVA_TRY(P.conv(q + "convs1." + std::to_string(j) + ".conv", rc.c1, ch, ch, rk[b]));
                    SVA_TRY(P.conv(q + "convs2." + std::to_string(j) + ".conv", rc.c2, ch, ch, rk[b]));
                }
        }
        HostTensor pw;
        SVA_CHECK(P.weight(h + "conv_post.conv", pw), P.err.c_str());
        SVA_CHECK(pw.numel() == (long)ch * e->post_k, "bad conv_post shape");
        std::vector<float> pt((size_t)e->post_k * ch);
        for (int cc = 0; cc < ch; ++cc)
            for (int j = 0; j < e->post_k; ++j) pt[(size_t)j * ch + cc] = pw.data[(size_t)cc * e->post_k + j];
        SVA_TRY(upload(e->allocs, &e->post_w, pt));
        SVA_TRY(P.vec(h + "conv_post.conv.bias", &e->post_b, 1));
    }
    // ---- pre-split operand planes of the encoder / vocoder weights (gemm_planes.hip) ----
    {
        const int enc_mode = c.mm_mode == 1 ? PLANES_H3 : -1;
        const int voc_mode = c.voc_dtype == 1 ? PLANES_H1 : enc_mode;
        SVA_CHECK((c.mm_mode == 0 || c.mm_mode == 1) && (c.voc_dtype == 0 || c.voc_dtype == 1),
                  "bad mm_mode / voc_dtype (mm_mode = 2, pre-split bf16 planes, was measured slower than mm_mode = 0 and removed in round 5)");
        std::vector<float> host;
        auto planes = [&](Lin& l, int mode, int min_n = 64) -> int {
            if (mode < 0 || !l.W || l.N < (mode == PLANES_H1 ? 32 : min_n) || l.K % 32 != 0) return 0;
            const long n = (long)l.N * l.K;
            host.resize(n);
            SVA_HIP(hipMemcpy(host.data(), l.W, sizeof(float) * n, hipMemcpyDeviceToHost));
            float mx = 0.f;
            for (long i = 0; i < n; ++i) mx = std::max(mx, fabsf(host[i]));
            SVA_TRY(dev_alloc(e->allocs, &l.Wp, (size_t)planes_count(mode) * n, false));
            SVA_TRY(make_weight_planes(l.W, l.N, l.K, mx, mode, l.Wp, &l.wp_inv, 0));
            l.pmode = mode;
            return 0;
        };
        auto front = [&](EncFront& F, int mode) -> int {
            if (!F.loaded) return 0;
            SVA_TRY(planes(F.stem, mode));
            for (int i = 0; i < 4; ++i) {
                SVA_TRY(planes(F.trans[i], mode));
                for (auto& cx : F.stages[i]) { SVA_TRY(planes(cx.pw1, mode)); SVA_TRY(planes(cx.pw2, mode)); }
            }
            for (int i = 0; i < 2; ++i) { SVA_TRY(planes(F.ds_conv[i], mode)); SVA_TRY(planes(F.ds_cnx[i].pw1, mode)); SVA_TRY(planes(F.ds_cnx[i].pw2, mode)); }
            return 0;
        };
        // (narrow levels: K = k * C padded with zero columns to whole 32-k blocks)
        auto planes_padded = [&](const Lin& l, int mode, unsigned short** out, float* inv, int* Kq) -> int {
            if (mode < 0 || !l.W) return 0;
            const int Kp = (l.K + 31) / 32 * 32;
            *Kq = Kp;
            const long n = (long)l.N * l.K;
            host.resize(n);
            SVA_HIP(hipMemcpy(host.data(), l.W, sizeof(float) * n, hipMemcpyDeviceToHost));
            std::vector<float> pad((size_t)l.N * Kp, 0.f);
            float mx = 0.f;
            for (int r = 0; r < l.N; ++r)
                for (int k = 0; k < l.K; ++k) { const float v = host[(long)r * l.K + k]; pad[(size_t)r * Kp + k] = v; mx = std::max(mx, fabsf(v)); }
            float* tmp = nullptr;
            SVA_HIP(hipMalloc((void**)&tmp, sizeof(float) * pad.size()));
            SVA_HIP(hipMemcpy(tmp, pad.data(), sizeof(float) * pad.size(), hipMemcpyHostToDevice));
            SVA_TRY(dev_alloc(e->allocs, out, (size_t)planes_count(mode) * pad.size(), false));
            int rc = make_weight_planes(tmp, l.N, Kp, mx, mode, *out, inv, 0);
            (void)hipDeviceSynchronize();
            (void)hipFree(tmp);
            return rc;
        };
        SVA_TRY(front(e->tokf, enc_mode));
        SVA_TRY(front(e->vocf, enc_mode));           // firefly.encode of the prompt produces FSQ indices: encoder grade
        for (auto& L : e->tr) { SVA_TRY(planes(L.wqkv, enc_mode)); SVA_TRY(planes(L.wo, enc_mode)); SVA_TRY(planes(L.w13, enc_mode)); SVA_TRY(planes(L.w2, enc_mode)); }
        for (int i = 0; i < 2; ++i) { SVA_TRY(planes(e->up_conv[i], voc_mode)); SVA_TRY(planes(e->up_cnx[i].pw1, voc_mode)); SVA_TRY(planes(e->up_cnx[i].pw2, voc_mode)); }
        SVA_TRY(planes(e->conv_pre, voc_mode));
        for (int i = 0; i < 5; ++i) {
            SVA_TRY(planes(e->ups[i], voc_mode));
            for (int bb = 0; bb < 3; ++bb)
                for (int j = 0; j < 3; ++j) {        // (down to the C = 32 level: the ResBlock convs' LDS-DMA form takes 32-column tiles)
                    SVA_TRY(planes(e->res[i][bb][j].c1, voc_mode, 32));
                    SVA_TRY(planes(e->res[i][bb][j].c2, voc_mode, 32));
                    ResConv& rc = e->res[i][bb][j];
                    if (rc.c1.N == 32 && rc.c1.Wp && rc.c2.Wp) { rc.q1 = rc.c1.Wp; rc.q2 = rc.c2.Wp; rc.q1_inv = rc.c1.wp_inv; rc.q2_inv = rc.c2.wp_inv; rc.Kq = rc.c1.K; }
                    if (rc.c1.N == 16) {
                        SVA_TRY(planes_padded(rc.c1, voc_mode, &rc.q1, &rc.q1_inv, &rc.Kq));
                        SVA_TRY(planes_padded(rc.c2, voc_mode, &rc.q2, &rc.q2_inv, &rc.Kq));
                    }
                }
        }
    }
    SVA_HIP(hipEventCreateWithFlags(&e->mega_ev, hipEventDisableTiming));      // chains persistent decode launches of different batches (stages.hip)
    e->host.clear();
    e->finalized = true;
    SVA_HIP(hipDeviceSynchronize());
    return 0;
}
