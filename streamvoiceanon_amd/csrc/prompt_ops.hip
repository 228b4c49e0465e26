// Device primitives of the prompt path's two speaker-embedding encoders (SURVEY.md 8f N1 iii / iv): CAM++ style vector and
// SparkTTS timbre latents (evaluations/infer_arvc.py:179-223).  They run ONCE per utterance on a few seconds of 16 kHz audio
// (~1 GFLOP), so these are plain, exact fp32 kernels -- correctness and zero host round trips matter here, not roofline -- behind
// small C entry points; the network topology lives in the host mirror (streamvoiceanon_amd/prompt_encoders.py), which owns the
// activation buffers.  Activations are channel-last [T][C] like everywhere else in the engine, so every Linear / Conv1d is
// the engine's conv-GEMM (f32 MFMA) with taps over shifted rows; the 2-D front of CAM++ is channel-first [C][F][T].
#include "engine.h"
#include "device_util.h"

#include <math.h>

using namespace sva;

namespace {

__global__ void affine_kernel(const float* __restrict__ x, long ldx, int T, int C, const float* __restrict__ scale, const float* __restrict__ shift,
                              int relu_mode, float* __restrict__ y, long ldy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * C) return;
    const int t = (int)(i / C), c = (int)(i - (long)t * C);
    float v = x[t * ldx + c];
    if (relu_mode == 2) v = fmaxf(v, 0.f);                 // bn(relu(x))   (ECAPA's Conv1dReluBn)
    v = v * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
    if (relu_mode == 1) v = fmaxf(v, 0.f);                 // relu(bn(x))   (CAM++'s batchnorm-relu)
    y[t * ldy + c] = v;
}

// unary ops: 1 log(max(x, floor)), 2 FSQ level-4 quantise (finite_scalar_quantization.py:126-139), 3 sigmoid
__global__ void unary_kernel(float* __restrict__ x, long n, int op, float p0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    if (op == 1) v = logf(fmaxf(v, p0));
    else if (op == 2) {
        const float half_l = (4 - 1) * (1.f + 1e-3f) / 2.f;
        const float shift = atanhf(0.5f / half_l);
        v = rintf(tanhf(v + shift) * half_l - 0.5f) / 2.f;       // torch.round = round half to even = rintf
    } else if (op == 3) v = 1.f / (1.f + expf(-v));
    else if (op == 4) v = -v;
    x[i] = v;
}

// column statistics over the T rows: mean[c], std[c] (unbiased or not)
// Block = 64 columns x CS_R row lanes: lane (c, r) sums rows r, r + CS_R, ... (two independent chains), the CS_R partials of a column are
// added in lane order through LDS -- a fixed summation order, in double.  (Round 4's one-thread-per-column loop was a chain of T dependent
// loads: 52 us per call, 54 calls per CAM++ pass -- 2.8 of the style encoder's 7 ms: profiles/r05_style_encoder_kernel_stats.csv)
constexpr int CS_R = 8;
__global__ __launch_bounds__(64 * CS_R) void colstats_kernel(const float* __restrict__ x, long ldx, int T, int C, float* __restrict__ mean, float* __restrict__ stdv, int unbiased) {
    __shared__ double part[CS_R][64];
    __shared__ double mu[64];
    const int cx = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const bool on = c < C;
    double s0 = 0.0, s1 = 0.0;
    if (on) {
        int t = r;
        for (; t + CS_R < T; t += 2 * CS_R) { s0 += (double)x[t * ldx + c]; s1 += (double)x[(t + CS_R) * ldx + c]; }
        if (t < T) s0 += (double)x[t * ldx + c];
    }
    part[r][cx] = s0 + s1;
    __syncthreads();
    if (r == 0) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < CS_R; ++i) s += part[i][cx];
        mu[cx] = s / T;
        if (on) mean[c] = (float)mu[cx];
    }
    if (!stdv) return;
    __syncthreads();
    const double m = mu[cx];
    double q0 = 0.0, q1 = 0.0;
    if (on) {
        int t = r;
        for (; t + CS_R < T; t += 2 * CS_R) {
            const double d0 = (double)x[t * ldx + c] - m, d1 = (double)x[(t + CS_R) * ldx + c] - m;
            q0 += d0 * d0; q1 += d1 * d1;
        }
        if (t < T) { const double d0 = (double)x[t * ldx + c] - m; q0 += d0 * d0; }
    }
    __syncthreads();
    part[r][cx] = q0 + q1;
    __syncthreads();
    if (r == 0 && on) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < CS_R; ++i) q += part[i][cx];
        stdv[c] = (float)sqrt(q / (unbiased ? (T - 1) : T));
    }
}

// CAMLayer context (modules/campplus/layers.py:103-119): ctx[t][c] = mean_t(y)[c] + mean over the 100-frame segment of t
// (block = 64 columns x 4 row lanes; the four partial sums of a column are added in lane order)
__global__ __launch_bounds__(256) void cam_context_kernel(const float* __restrict__ y, long ldy, int T, int C, int seg, const float* __restrict__ mean, float* __restrict__ ctx, long ldc) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, s = blockIdx.y;
    const bool on = c < C;
    const int lo = s * seg, hi = min(T, lo + seg);
    float a = 0.f;
    if (on)
        for (int t = lo + r; t < hi; t += 4) a += y[t * ldy + c];
    part[r][cx] = a;
    __syncthreads();
    if (!on) return;
    a = ((part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx])) / (float)(hi - lo) + mean[c];      // avg_pool1d(ceil_mode=True): the last window divides by its valid length
    for (int t = lo + r; t < hi; t += 4) ctx[t * ldc + c] = a;
}

// y[t][c] *= (sig ? sigmoid(m) : m)[t * ldm + c]   (ldm = 0: one row broadcast over time -- SE connection)
__global__ void mul_kernel(float* __restrict__ y, long ldy, const float* __restrict__ m, long ldm, int T, int C, int sig) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * C) return;
    const int t = (int)(i / C), c = (int)(i - (long)t * C);
    float g = m[t * ldm + c];
    if (sig) g = 1.f / (1.f + expf(-g));
    y[t * ldy + c] *= g;
}
__global__ void add_kernel(float* __restrict__ y, long ldy, const float* __restrict__ x, long ldx, int T, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * C) return;
    const int t = (int)(i / C), c = (int)(i - (long)t * C);
    y[t * ldy + c] += x[t * ldx + c];
}

// generic conv / linear for shapes the MFMA kernel does not take (Cin not a multiple of 16): one thread per output
__global__ void conv_naive_kernel(const float* __restrict__ x, long ldx, int T, int stride, int dil, int taps, int Cin, const float* __restrict__ W,
                                  const float* __restrict__ bias, int N, float* __restrict__ y, long ldy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * N) return;
    const int t = (int)(i / N), n = (int)(i - (long)t * N);
    float a = bias ? bias[n] : 0.f;
    for (int tap = 0; tap < taps; ++tap) {
        const float* xr = x + ((long)t * stride + (long)tap * dil) * ldx;
        const float* wr = W + ((long)n * taps + tap) * Cin;
        for (int c = 0; c < Cin; ++c) a = fmaf(xr[c], wr[c], a);
    }
    y[t * ldy + n] = a;
}

// Conv2d k x k (k = 1 or 3, padding k / 2, stride (sf, 1), no bias) + folded BatchNorm (+ residual) (+ ReLU), channel-first
// [C][F][T]: the FCM front of CAM++ (modules/campplus/DTDNN.py:14-48, layers.py:227-266).  One thread per output element.
__global__ void conv2d_kernel(const float* __restrict__ x, int Cin, int F, int T, const float* __restrict__ W, int Cout, int k, int sf, const float* __restrict__ scale,
                              const float* __restrict__ shift, const float* __restrict__ res, int relu, float* __restrict__ y, int Fo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cout * Fo * T) return;
    const int t = (int)(i % T), fo = (int)((i / T) % Fo), co = (int)(i / ((long)T * Fo));
    const int pad = k / 2;
    float a = 0.f;
    for (int ci = 0; ci < Cin; ++ci)
        for (int kf = 0; kf < k; ++kf) {
            const int f = fo * sf + kf - pad;
            if (f < 0 || f >= F) continue;
            for (int kt = 0; kt < k; ++kt) {
                const int tt = t + kt - pad;
                if (tt < 0 || tt >= T) continue;
                a = fmaf(x[((long)ci * F + f) * T + tt], W[(((long)co * Cin + ci) * k + kf) * k + kt], a);
            }
        }
    a = a * scale[co] + shift[co];
    if (res) a += res[i];
    if (relu) a = fmaxf(a, 0.f);
    y[i] = a;
}
// [C][F][T] channel-first -> channel-last rows [T][C*F] (x.reshape(B, C*F, T), DTDNN.py:46-47)
__global__ void cf_to_rows_kernel(const float* __restrict__ x, int CF, int T, float* __restrict__ y, long ldy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)CF * T) return;
    const int t = (int)(i % T), cf = (int)(i / T);
    y[t * ldy + cf] = x[i];
}

// Kaldi fbank framing (torchaudio.compliance.kaldi: snip_edges, remove_dc_offset, preemphasis 0.97, povey window), one wave per
// frame: frames[m][npad] (zero padded to the FFT size)
__global__ void fbank_frames_kernel(const float* __restrict__ wave, int ws, int sh, int npad, float* __restrict__ frames) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const float* src = wave + (long)m * sh;
    float s = 0.f;
    for (int i = lane; i < ws; i += 64) s += src[i];
    s = wave_sum(s);
    const float mean = s / (float)ws;
    for (int i = lane; i < npad; i += 64) {
        float v = 0.f;
        if (i < ws) {
            const float cur = src[i] - mean, prev = src[i > 0 ? i - 1 : 0] - mean;
            const float w = powf(0.5f - 0.5f * cospif(2.0f * (float)i / (float)(ws - 1)), 0.85f);
            v = (cur - 0.97f * prev) * w;
        }
        frames[(long)m * npad + i] = v;
    }
}
// torch.stft framing with center = True, reflect padding, periodic Hann window of `win` samples centred in n_fft
__global__ void stft_frames_kernel(const float* __restrict__ wave, int n, int n_fft, int win, int hop, float* __restrict__ frames) {
    const int m = blockIdx.x;
    const int off = (n_fft - win) / 2;
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
        float v = 0.f;
        if (i >= off && i < off + win) {
            long idx = (long)m * hop + i - n_fft / 2;
            if (idx < 0) idx = -idx;
            if (idx >= n) idx = 2L * (n - 1) - idx;
            const float w = 0.5f - 0.5f * cospif(2.0f * (float)(i - off) / (float)win);
            v = wave[idx] * w;
        }
        frames[(long)m * n_fft + i] = v;
    }
}
// |DFT|^power of real frames: out[m][k], k = 0 .. n_fft / 2; columns up to ldo zeroed.  One workgroup per frame.
__global__ void dft_kernel(const float* __restrict__ frames, int n_fft, int power2, float* __restrict__ out, int ldo) {
    extern __shared__ float fr[];
    const int m = blockIdx.x;
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x) fr[i] = frames[(long)m * n_fft + i];
    __syncthreads();
    for (int k = threadIdx.x; k < ldo; k += blockDim.x) {
        float v = 0.f;
        if (k <= n_fft / 2) {
            double re = 0.0, im = 0.0;
            for (int i = 0; i < n_fft; ++i) {
                float sn, cs;
                sincospif(2.0f * (float)((k * i) % n_fft) / (float)n_fft, &sn, &cs);
                re += (double)fr[i] * cs;
                im -= (double)fr[i] * sn;
            }
            const double p = re * re + im * im;
            v = power2 ? (float)p : (float)sqrt(p);
        }
        out[(long)m * ldo + k] = v;
    }
}

// attention of Lq queries over Lk keys (only the first n_valid count), H heads of 64: q [Lq][H*64], kv [Lk][2*H*64] (k | v)
// (Attend.forward, perceiver_encoder.py:115-150).  One workgroup of 64 threads per (query, head).
__global__ void attention_kernel(const float* __restrict__ q, const float* __restrict__ kv, int Lk, int n_valid, int H, float* __restrict__ out, float* __restrict__ scratch) {
    const int iq = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int D = H * 64;
    float* sc = scratch + ((long)iq * H + h) * Lk;
    const float qd = q[(long)iq * D + h * 64 + lane] * 0.125f;
    float mx = -INFINITY;
    for (int j = 0; j < n_valid; ++j) {
        const float s = wave_sum(qd * kv[(long)j * 2 * D + h * 64 + lane]);
        if (lane == 0) sc[j] = s;
        mx = fmaxf(mx, s);
    }
    __syncthreads();
    float den = 0.f, acc = 0.f;
    for (int j = 0; j < n_valid; ++j) {
        const float e = expf(sc[j] - mx);
        den += e;
        acc = fmaf(e, kv[(long)j * 2 * D + D + h * 64 + lane], acc);
    }
    out[(long)iq * D + h * 64 + lane] = acc / den;
}
// GEGLU (perceiver_encoder.py:208-211): out[t][d] = gelu(h[t][Dh + d]) * h[t][d], d < Dh; columns Dh .. ldo zeroed
__global__ void geglu_kernel(const float* __restrict__ h, long ldh, int T, int Dh, float* __restrict__ out, long ldo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * ldo) return;
    const int t = (int)(i / ldo), d = (int)(i - (long)t * ldo);
    float v = 0.f;
    if (d < Dh) {
        const float g = h[t * ldh + Dh + d];
        v = 0.5f * g * (1.f + erff(g * 0.70710678118654752f)) * h[t * ldh + d];
    }
    out[i] = v;
}
// RMSNorm of perceiver_encoder.py:176-193: F.normalize(x, dim=-1) * sqrt(C) * gamma; one wave per row
__global__ void l2norm_kernel(const float* __restrict__ x, int C, const float* __restrict__ gamma, float scale, float* __restrict__ y) {
    const int t = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += x[(long)t * C + c] * x[(long)t * C + c];
    s = wave_sum(s);
    const float inv = scale / fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < C; c += 64) y[(long)t * C + c] = x[(long)t * C + c] * inv * gamma[c];
}

inline unsigned nblk(long n, int th = 256) { return (unsigned)((n + th - 1) / th); }

}  // namespace

#define SVA_TRY_OPS(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)
// Every op runs on the ENGINE'S ops stream (created on first use, non-blocking), in call order: a real stream can be captured into a
// hipGraph (sva_ops_capture_*), the legacy default stream cannot.
static int ops_stream(sva_engine* e, hipStream_t* out) {
    if (!e->ops_stream) {
        SVA_HIP(hipStreamCreateWithFlags(&e->ops_stream, hipStreamNonBlocking));
        SVA_TRY_OPS(conv_gemm_prepare_stream(e->ops_stream));          // (split-K scratch up front: a launch inside a capture must not allocate)
    }
    *out = e->ops_stream;
    return 0;
}
#define OPS_ENTER(e)                                   \
    SVA_CHECK((e) != nullptr, "null engine");          \
    SVA_HIP(hipSetDevice((e)->device));                \
    (void)hipGetLastError();                           \
    hipStream_t os_ = nullptr;                         \
    { int rc_ = ops_stream((e), &os_); if (rc_) return rc_; }

extern "C" int sva_dev_alloc(sva_engine* e, long n, float** out) {
    OPS_ENTER(e);
    SVA_CHECK(out && n > 0, "bad argument");
    SVA_HIP(hipMalloc((void**)out, sizeof(float) * (size_t)n));
    SVA_HIP(hipMemsetAsync(*out, 0, sizeof(float) * (size_t)n, os_));
    // (a caller may hand the fresh buffer to a stream of its own -- the zero fill is complete when this returns; inside a capture of
    // the ops stream there is nothing to wait for: hipMalloc is not capturable and this entry point is not called there)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(os_, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) SVA_HIP(hipStreamSynchronize(os_));
    return 0;
}
extern "C" int sva_dev_free(sva_engine* e, float* p) {
    OPS_ENTER(e);
    SVA_HIP(hipDeviceSynchronize());
    SVA_HIP(hipFree(p));
    return 0;
}
extern "C" int sva_dev_upload(sva_engine* e, float* dst, const float* src, long n) {
    OPS_ENTER(e);
    SVA_HIP(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, os_));
    SVA_HIP(hipStreamSynchronize(os_));          // (src is the caller's pageable buffer)
    return 0;
}
extern "C" int sva_dev_download(sva_engine* e, float* dst, const float* src, long n) {
    OPS_ENTER(e);
    SVA_HIP(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, os_));
    SVA_HIP(hipStreamSynchronize(os_));
    return 0;
}

// y[t][n] = bias[n] + sum_{tap, c} x[(t * stride + tap * dil) * ldx + c] * W[n][tap * Cin + c]   (x points at tap 0 of row 0)
extern "C" int sva_op_conv(sva_engine* e, const float* x, long ldx, int T, int stride, int dil, int taps, int Cin, const float* W, const float* bias, int N,
                           float* y, long ldy) {
    OPS_ENTER(e);
    SVA_CHECK(x && W && y && T > 0 && N > 0 && Cin > 0, "bad argument");
    const bool mfma = Cin % 16 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0;
    if (mfma) {
        ConvGemm g;
        g.A = x; g.a_bstride = 0; g.a_off = 0; g.lda = (int)ldx; g.T = T; g.M = T; g.stride = stride; g.dil = dil; g.taps = taps; g.Cin = Cin;
        g.W = W; g.N = N; g.bias = bias; g.C = y; g.c_bstride = 0; g.c_off = 0; g.ldc = (int)ldy;
        return launch_conv_gemm(g, os_);
    }
    hipLaunchKernelGGL(conv_naive_kernel, dim3(nblk((long)T * N)), dim3(256), 0, os_, x, ldx, T, stride, dil, taps, Cin, W, bias, N, y, ldy);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_affine(sva_engine* e, const float* x, long ldx, int T, int C, const float* scale, const float* shift, int relu_mode, float* y, long ldy) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(affine_kernel, dim3(nblk((long)T * C)), dim3(256), 0, os_, x, ldx, T, C, scale, shift, relu_mode, y, ldy);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_unary(sva_engine* e, float* x, long n, int op, float p0) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(unary_kernel, dim3(nblk(n)), dim3(256), 0, os_, x, n, op, p0);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_colstats(sva_engine* e, const float* x, long ldx, int T, int C, float* mean, float* stdv, int unbiased) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(colstats_kernel, dim3(nblk(C, 64)), dim3(64 * CS_R), 0, os_, x, ldx, T, C, mean, stdv, unbiased);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_cam_context(sva_engine* e, const float* y, long ldy, int T, int C, int seg, const float* mean, float* ctx, long ldc) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(cam_context_kernel, dim3(nblk(C, 64), (T + seg - 1) / seg), dim3(256), 0, os_, y, ldy, T, C, seg, mean, ctx, ldc);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_mul(sva_engine* e, float* y, long ldy, const float* m, long ldm, int T, int C, int sigmoid) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(mul_kernel, dim3(nblk((long)T * C)), dim3(256), 0, os_, y, ldy, m, ldm, T, C, sigmoid);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_add(sva_engine* e, float* y, long ldy, const float* x, long ldx, int T, int C) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(add_kernel, dim3(nblk((long)T * C)), dim3(256), 0, os_, y, ldy, x, ldx, T, C);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_conv2d(sva_engine* e, const float* x, int Cin, int F, int T, const float* W, int Cout, int k, int stride_f, const float* scale,
                             const float* shift, const float* res, int relu, float* y) {
    OPS_ENTER(e);
    SVA_CHECK(k == 1 || k == 3, "conv2d: k must be 1 or 3");
    const int Fo = (F + 2 * (k / 2) - k) / stride_f + 1;
    hipLaunchKernelGGL(conv2d_kernel, dim3(nblk((long)Cout * Fo * T)), dim3(256), 0, os_, x, Cin, F, T, W, Cout, k, stride_f, scale, shift, res, relu, y, Fo);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_cf_to_rows(sva_engine* e, const float* x, int CF, int T, float* y, long ldy) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(cf_to_rows_kernel, dim3(nblk((long)CF * T)), dim3(256), 0, os_, x, CF, T, y, ldy);
    SVA_HIP(hipGetLastError());
    return 0;
}
// Kaldi fbank power spectrum: wave[n] -> out[m][ldo] (257 bins, rest zero), m = 1 + (n - 400) / 160; returns m through *frames_out
extern "C" int sva_op_fbank_power(sva_engine* e, const float* wave, long n, float* frames_scratch, float* out, int ldo, int* frames_out) {
    OPS_ENTER(e);
    const int ws = 400, sh = 160, npad = 512;
    SVA_CHECK(n >= ws && ldo >= npad / 2 + 1, "fbank: input shorter than one 25 ms frame");
    const int m = 1 + (int)((n - ws) / sh);
    hipLaunchKernelGGL(fbank_frames_kernel, dim3(m), dim3(64), 0, os_, wave, ws, sh, npad, frames_scratch);
    hipLaunchKernelGGL(dft_kernel, dim3(m), dim3(256), npad * sizeof(float), os_, frames_scratch, npad, 1, out, ldo);
    SVA_HIP(hipGetLastError());
    if (frames_out) *frames_out = m;
    return 0;
}
// torch.stft magnitude (center, reflect, periodic Hann of `win` in n_fft): wave[n] -> out[1 + n / hop][ldo]
extern "C" int sva_op_stft_mag(sva_engine* e, const float* wave, long n, int n_fft, int win, int hop, float* frames_scratch, float* out, int ldo, int* frames_out) {
    OPS_ENTER(e);
    SVA_CHECK(n > n_fft / 2 && ldo >= n_fft / 2 + 1, "stft: input too short for reflect padding");
    const int m = 1 + (int)(n / hop);
    hipLaunchKernelGGL(stft_frames_kernel, dim3(m), dim3(256), 0, os_, wave, (int)n, n_fft, win, hop, frames_scratch);
    hipLaunchKernelGGL(dft_kernel, dim3(m), dim3(256), n_fft * sizeof(float), os_, frames_scratch, n_fft, 0, out, ldo);
    SVA_HIP(hipGetLastError());
    if (frames_out) *frames_out = m;
    return 0;
}
extern "C" int sva_op_attention(sva_engine* e, const float* q, const float* kv, int Lq, int Lk, int n_valid, int H, float* out, float* scratch) {
    OPS_ENTER(e);
    SVA_CHECK(n_valid >= 1 && n_valid <= Lk, "attention: bad key count");
    hipLaunchKernelGGL(attention_kernel, dim3(Lq, H), dim3(64), 0, os_, q, kv, Lk, n_valid, H, out, scratch);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_geglu(sva_engine* e, const float* h, long ldh, int T, int Dh, float* out, long ldo) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(geglu_kernel, dim3(nblk((long)T * ldo)), dim3(256), 0, os_, h, ldh, T, Dh, out, ldo);
    SVA_HIP(hipGetLastError());
    return 0;
}
extern "C" int sva_op_l2norm(sva_engine* e, const float* x, int T, int C, const float* gamma, float scale, float* y) {
    OPS_ENTER(e);
    hipLaunchKernelGGL(l2norm_kernel, dim3(T), dim3(64), 0, os_, x, C, gamma, scale, y);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ---- the op sequence of one encoder call as a hipGraph: the host mirror issues ~600 launches per calculate_prompt;
// between sva_ops_capture_begin and _end they are RECORDED on the ops stream instead of executed (no allocation, upload or download in
// between: prompt_encoders.py replays its allocations from the recording run), and the instantiated graph replays them in one launch ----
extern "C" int sva_ops_capture_begin(sva_engine* e) {
    OPS_ENTER(e);
    SVA_HIP(hipStreamSynchronize(os_));
    SVA_HIP(hipStreamBeginCapture(os_, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int sva_ops_capture_end(sva_engine* e, void** graph_exec) {
    OPS_ENTER(e);
    SVA_CHECK(graph_exec, "null output");
    hipGraph_t graph = nullptr;
    SVA_HIP(hipStreamEndCapture(os_, &graph));
    hipGraphExec_t ex = nullptr;
    const hipError_t ie = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    SVA_HIP(ie);
    *graph_exec = ex;
    return 0;
}
extern "C" int sva_ops_graph_launch(sva_engine* e, void* graph_exec) {
    OPS_ENTER(e);
    SVA_CHECK(graph_exec, "null graph");
    SVA_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), os_));
    return 0;
}
extern "C" int sva_ops_graph_free(sva_engine* e, void* graph_exec) {
    OPS_ENTER(e);
    SVA_HIP(hipStreamSynchronize(os_));
    if (graph_exec) SVA_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
    return 0;
}
