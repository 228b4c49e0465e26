// Non-GEMM kernels of the sva engine (gfx950).  wave = 64 lanes everywhere.
//
// Reference behaviour restated by each kernel is cited next to it (paths relative to the
// StreamVoiceAnon repository); the CPU oracle (oracle/sva_oracle.py) is the checker.
#include "kernels.h"
#include "device_util.h"
#include "planes_split.h"

namespace sva {

__device__ __forceinline__ float silu_acc(float x) { return x / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------
// E0: audio ring write.  The reference shifts a [1, W*2048] window left by one chunk and
// appends the new chunk (evaluations/infer_arvc.py:495-496); here the window is a ring whose
// oldest sample sits at ((step + add) * n) % N, `step` being the device-side chunk counter.
// ------------------------------------------------------------------------------------------
__global__ void ring_write_kernel(float* ring, const int* step, int N, const float* chunk, int n) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int start = (int)(((long)(*step) * n) % N);
    int idx = start + i;
    if (idx >= N) idx -= N;
    ring[(long)b * N + idx] = chunk[(long)b * n + i];
}
int launch_ring_write(float* ring, int* step, int B, int N, const float* chunk, int n, hipStream_t st) {
    dim3 grid((n + 255) / 256, B);
    hipLaunchKernelGGL(ring_write_kernel, grid, dim3(256), 0, st, ring, step, N, chunk, n);
    SVA_HIP(hipGetLastError());
    return 0;
}

__global__ void fill_i32_kernel(int* p, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void add_i32_kernel(int* p, int v) { *p += v; }
int launch_add_i32(int* p, int v, hipStream_t st) {
    hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(1), 0, st, p, v);
    SVA_HIP(hipGetLastError());
    return 0;
}
int launch_fill_i32(int* p, int n, int v, hipStream_t st) {
    hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n, v);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// E1: causal STFT magnitude (modules/vqgan/spectrogram.py:26-65): frame m = samples
// [512 m - 1536, 512 m + 512) of the window (zeros before the window start), periodic Hann,
// 2048-point DFT, sqrt(re^2 + im^2 + 1e-6).  One workgroup per frame; radix-2 DIT in LDS.
// `step_info` = {step pointer, chunk n, add}: oldest sample at ((step+add)*n) % N.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_mag_kernel(const float* __restrict__ ring, const int* step, int n_chunk,
                                                       int add, int N, const float2* __restrict__ tw,
                                                       const float* __restrict__ hann, float* __restrict__ mag,
                                                       int ldm, long mag_bstride, int m0, int seg_a, int m0b, int row_b0) {
    __shared__ float re[2048];
    __shared__ float im[2048];
    // frames [0, seg_a) of the grid: window frames m0 .. -> output rows 0 ..; the rest: window frames m0b .. -> output rows row_b0 ..
    const bool second = (int)blockIdx.x >= seg_a;
    const int m = second ? m0b + ((int)blockIdx.x - seg_a) : m0 + (int)blockIdx.x;
    const int out_row = second ? row_b0 + ((int)blockIdx.x - seg_a) : (int)blockIdx.x;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int start = step ? (int)(((long)(*step + add) * n_chunk) % N) : 0;
    const float* rb = ring + (long)b * N;
    for (int i = tid; i < 2048; i += 256) {
        const int j = 512 * m - 1536 + i;
        float v = 0.f;
        if (j >= 0) {
            int idx = start + j;
            if (idx >= N) idx -= N;
            v = rb[idx] * hann[i];
        }
        const int r = (int)(__brev((unsigned)i) >> 21);
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s <= 11; ++s) {
        const int half = 1 << (s - 1);
        for (int jb = tid; jb < 1024; jb += 256) {
            const int pos = jb & (half - 1);
            const int i0 = ((jb >> (s - 1)) << s) + pos;
            const int i1 = i0 + half;
            const float2 w = tw[pos << (11 - s)];
            const float xr = re[i1], xi = im[i1];
            const float tr = w.x * xr - w.y * xi, ti = w.x * xi + w.y * xr;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr;
            im[i0] = ui + ti;
            re[i1] = ur - tr;
            im[i1] = ui - ti;
        }
        __syncthreads();
    }
    float* out = mag + (long)b * mag_bstride + (long)out_row * ldm;
    for (int k = tid; k < ldm; k += 256) out[k] = k <= 1024 ? sqrtf(re[k] * re[k] + im[k] * im[k] + 1e-6f) : 0.f;
}
int launch_stft_mag_ring(const float* ring, const int* step, int n_chunk, int add, int B, int N, const float2* tw,
                         const float* hann, float* mag, int ldm, long mag_bstride, int m0, int nfr, hipStream_t st) {
    SVA_CHECK(N % 512 == 0 && ldm >= 1025 && m0 >= 0 && m0 + nfr <= N / 512, "stft: bad shape");
    hipLaunchKernelGGL(stft_mag_kernel, dim3(nfr, B), dim3(256), 0, st, ring, step, n_chunk, add, N, tw, hann, mag, ldm, mag_bstride, m0, nfr, 0, 0);
    SVA_HIP(hipGetLastError());
    return 0;
}
// two frame ranges in one launch: frames [m0, m0 + nfr) -> rows [0, nfr), frames [m0b, m0b + nfrb) -> rows [row_b0, row_b0 + nfrb)
int launch_stft_mag_ring2(const float* ring, const int* step, int n_chunk, int add, int B, int N, const float2* tw,
                          const float* hann, float* mag, int ldm, long mag_bstride, int m0, int nfr, int m0b, int nfrb, int row_b0, hipStream_t st) {
    SVA_CHECK(N % 512 == 0 && ldm >= 1025 && m0 >= 0 && m0 + nfr <= N / 512 && m0b >= 0 && m0b + nfrb <= N / 512, "stft: bad shape");
    hipLaunchKernelGGL(stft_mag_kernel, dim3(nfr + nfrb, B), dim3(256), 0, st, ring, step, n_chunk, add, N, tw, hann, mag, ldm, mag_bstride, m0, nfr,
                       m0b, row_b0);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// ConvNeXtBlock prologue: depthwise causal k=7 conv + LayerNorm(eps 1e-6, biased variance)
// (modules/vqgan/modules/firefly.py:421-427, 92-103, 361-365).  One wave per output row.
// ------------------------------------------------------------------------------------------
// NV = float4 groups per lane (C = 256 * NV; C < 256: the upper lanes idle).  All 14 * NV loads of a lane are independent
// 16-byte loads, so one round trip to L2 covers the row.
template <int NV>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, long x_bstride, long x_off, int T,
                                                         int C, int rows, const float* __restrict__ wT,
                                                         const float* __restrict__ bias, const float* __restrict__ lw,
                                                         const float* __restrict__ lb, float eps, float* __restrict__ out,
                                                         long o_bstride, unsigned short* __restrict__ outp, long op_pstride, int op_planes,
                                                         long op_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (blockDim.x >> 6) + wave;
    if (row >= rows) return;
    const int b = row / T, t = row - b * T;
    const float* xr = x + (long)b * x_bstride + x_off + (long)t * C;
    float4 xv[NV][7], wv[NV][7], v[NV];
    bool on[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        on[i] = c < C;
        const int cc = on[i] ? c : 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            xv[i][j] = *reinterpret_cast<const float4*>(xr + (long)j * C + cc);
            wv[i][j] = *reinterpret_cast<const float4*>(wT + j * C + cc);
        }
        v[i] = *reinterpret_cast<const float4*>(bias + cc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {       // same tap order as the scalar formulation: acc = fma(w_j, x_j, acc), j ascending
            v[i].x = fmaf(wv[i][j].x, xv[i][j].x, v[i].x);
            v[i].y = fmaf(wv[i][j].y, xv[i][j].y, v[i].y);
            v[i].z = fmaf(wv[i][j].z, xv[i][j].z, v[i].z);
            v[i].w = fmaf(wv[i][j].w, xv[i][j].w, v[i].w);
        }
        if (on[i]) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (!on[i]) continue;
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float inv = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
    float* o = out + (long)b * o_bstride + (long)t * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (!on[i]) continue;
        const int c = 4 * (lane + 64 * i);
        const float4 g = *reinterpret_cast<const float4*>(lw + c), h = *reinterpret_cast<const float4*>(lb + c);
        float4 r;
        r.x = (v[i].x - mean) * inv * g.x + h.x;
        r.y = (v[i].y - mean) * inv * g.y + h.y;
        r.z = (v[i].z - mean) * inv * g.z + h.z;
        r.w = (v[i].w - mean) * inv * g.w + h.w;
        if (outp) {          // the consumer is a planes GEMM (gemm_planes.hip): fp16 hi (+ lo) parts instead of the fp32 row
            unsigned short hi[4], lo[4];
            h3_split(r.x, hi[0], lo[0]); h3_split(r.y, hi[1], lo[1]); h3_split(r.z, hi[2], lo[2]); h3_split(r.w, hi[3], lo[3]);
            const long po = op_rows > 0 ? plane_off_blocked((long)b * T + t, c, op_rows) : (long)b * o_bstride + (long)t * C + c;
            *reinterpret_cast<uint2*>(outp + po) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
            if (op_planes > 1) *reinterpret_cast<uint2*>(outp + op_pstride + po) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
        } else {
            *reinterpret_cast<float4*>(o + c) = r;
        }
    }
}
int launch_dwconv7_ln(const float* x, long x_bstride, long x_off, int B, int T, int C, const float* wT,
                      const float* bias, const float* ln_w, const float* ln_b, float eps, float* out, long o_bstride,
                      hipStream_t st, unsigned short* outp, long op_pstride, int op_planes, long op_rows) {
    SVA_CHECK(C % 4 == 0 && C <= 512 && x_bstride % 4 == 0 && x_off % 4 == 0 && o_bstride % 4 == 0,
              "dwconv7_ln: C must be a multiple of 4, <= 512, float4-aligned rows");
    const int rows = B * T;
    const int wpb = rows >= 4096 ? 4 : 1;       // few rows: one wave per workgroup so the rows spread over the CUs
    dim3 grid((rows + wpb - 1) / wpb);
    if (C <= 256)
        hipLaunchKernelGGL((dwconv7_ln_kernel<1>), grid, dim3(64 * wpb), 0, st, x, x_bstride, x_off, T, C, rows, wT, bias, ln_w, ln_b, eps, out, o_bstride, outp, op_pstride, op_planes, op_rows);
    else
        hipLaunchKernelGGL((dwconv7_ln_kernel<2>), grid, dim3(64 * wpb), 0, st, x, x_bstride, x_off, T, C, rows, wT, bias, ln_w, ln_b, eps, out, o_bstride, outp, op_pstride, op_planes, op_rows);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// row LayerNorm (firefly.py:361-371) / RMSNorm (dual_ar_stream.py:979-990,
// windowed_transformer.py:248-259: x * rsqrt(mean(x^2) + eps) * w), one wave per row.
// ------------------------------------------------------------------------------------------
template <int NPL, bool RMS>
__global__ __launch_bounds__(256) void norm_rows_kernel(const float* __restrict__ x, long x_bstride, long x_off, int ldx,
                                                        int T, int C, int rows, const float* __restrict__ w,
                                                        const float* __restrict__ bvec, float eps,
                                                        float* __restrict__ out, long o_bstride, long o_off, int ldo, int skip_lo, int skip_hi,
                                                        unsigned short* __restrict__ outp, long op_pstride, int op_planes, long op_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / T, t = row - b * T;
    if (t >= skip_lo && t < skip_hi) return;          // rows the caller keeps (streaming history of the merged encoder pass)
    const float* xr = x + (long)b * x_bstride + x_off + (long)t * ldx;
    float* o = out + (long)b * o_bstride + o_off + (long)t * ldo;
    // planes output (the consumer is a planes GEMM): fp16 hi (+ lo) parts at the fp32 element's index, or K-blocked (planes_split.h)
    auto put = [&](int c, float val) {
        if (outp) {
            unsigned short hi, lo;
            h3_split(val, hi, lo);
            const long po = op_rows > 0 ? plane_off_blocked((long)b * (o_bstride / ldo) + o_off / ldo + t, c, op_rows) : (long)b * o_bstride + o_off + (long)t * ldo + c;
            outp[po] = hi;
            if (op_planes > 1) outp[op_pstride + po] = lo;
        } else {
            o[c] = val;
        }
    };
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        v[i] = xr[lane + 64 * i];
        s += RMS ? v[i] * v[i] : v[i];
    }
    if (RMS) {
        const float inv = 1.f / sqrtf(wave_sum(s) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NPL; ++i) put(lane + 64 * i, v[i] * inv * w[lane + 64 * i]);
    } else {
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const float d = v[i] - mean;
            q = fmaf(d, d, q);
        }
        const float inv = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            put(c, (v[i] - mean) * inv * w[c] + bvec[c]);
        }
    }
}
template <bool RMS>
static int launch_norm_rows(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int C, const float* w,
                            const float* b, float eps, float* out, long o_bstride, long o_off, int ldo, hipStream_t st, int skip_lo = 0, int skip_hi = 0,
                            unsigned short* outp = nullptr, long op_pstride = 0, int op_planes = 0, long op_rows = 0) {
    SVA_CHECK(C % 64 == 0 && C <= 1024, "norm_rows: C must be a multiple of 64, <= 1024");
    SVA_CHECK(!outp || op_rows == 0 || (o_bstride % ldo == 0 && o_off % ldo == 0), "norm_rows: K-blocked planes need whole rows");
    const int rows = B * T;
    dim3 grid((rows + 3) / 4);
#define SVA_NR(N_) hipLaunchKernelGGL((norm_rows_kernel<N_, RMS>), grid, dim3(256), 0, st, x, x_bstride, x_off, ldx, T, C, rows, w, b, eps, out, o_bstride, o_off, ldo, skip_lo, skip_hi, outp, op_pstride, op_planes, op_rows)
    switch (C / 64) {
        case 1: SVA_NR(1); break;
        case 2: SVA_NR(2); break;
        case 3: SVA_NR(3); break;
        case 4: SVA_NR(4); break;
        case 6: SVA_NR(6); break;
        case 8: SVA_NR(8); break;
        case 12: SVA_NR(12); break;
        case 16: SVA_NR(16); break;
        default: set_error("norm_rows: unsupported C"); return -1;
    }
#undef SVA_NR
    SVA_HIP(hipGetLastError());
    return 0;
}
int launch_layernorm_rows(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int C, const float* w,
                          const float* b, float eps, float* out, long o_bstride, long o_off, int ldo, hipStream_t st, int skip_lo, int skip_hi) {
    return launch_norm_rows<false>(x, x_bstride, x_off, ldx, B, T, C, w, b, eps, out, o_bstride, o_off, ldo, st, skip_lo, skip_hi);
}
int launch_rmsnorm_rows(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int C, const float* w,
                        float eps, float* out, long o_bstride, long o_off, int ldo, hipStream_t st, unsigned short* outp, long op_pstride, int op_planes, long op_rows) {
    return launch_norm_rows<true>(x, x_bstride, x_off, ldx, B, T, C, w, nullptr, eps, out, o_bstride, o_off, ldo, st, 0, 0, outp, op_pstride, op_planes, op_rows);
}

// ------------------------------------------------------------------------------------------
// E7 attention (modules/vqgan/windowed_transformer.py:163-194, mask 298-303 = plain causal
// because window 512 >= T): one workgroup per (head, stream); K (RoPE'd) and V in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ rope,
                                                            int T, int H, float* __restrict__ out, int row0) {
    constexpr int HD = 64, LDK = HD + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                    // [T][65]
    float* Vs = Ks + (long)T * LDK;      // [T][65]
    float* qs = Vs + (long)T * LDK;      // [4][64]
    float* ps = qs + 4 * HD;             // [4][T]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int D = H * HD;
    const float* base = qkv + (long)b * T * 3 * D;
    for (int idx = tid; idx < T * (HD / 2); idx += 256) {
        const int t = idx / (HD / 2), p = idx - t * (HD / 2);
        const float* row = base + (long)t * 3 * D;
        const float k0 = row[D + h * HD + 2 * p], k1 = row[D + h * HD + 2 * p + 1];
        const float c = rope[(t * (HD / 2) + p) * 2], s = rope[(t * (HD / 2) + p) * 2 + 1];
        Ks[t * LDK + 2 * p] = k0 * c - k1 * s;
        Ks[t * LDK + 2 * p + 1] = k1 * c + k0 * s;
        Vs[t * LDK + 2 * p] = row[2 * D + h * HD + 2 * p];
        Vs[t * LDK + 2 * p + 1] = row[2 * D + h * HD + 2 * p + 1];
    }
    __syncthreads();
    const float scale = 0.125f;          // 1/sqrt(64)
    // query rows interleaved over (blockIdx.z, wave) so the causal work is balanced; T % (4*gridDim.z) == 0
    // keeps the trip count uniform across the workgroup's waves (barriers inside the loop)
    for (int rb = row0 + blockIdx.z * 4; rb < T; rb += 4 * gridDim.z) {
        const int r = min(rb + wave, T - 1);          // surplus waves of a short tail recompute the last row (benign)
        if (lane < HD / 2) {
            const float* row = base + (long)r * 3 * D + h * HD;
            const float q0 = row[2 * lane], q1 = row[2 * lane + 1];
            const float c = rope[(r * (HD / 2) + lane) * 2], s = rope[(r * (HD / 2) + lane) * 2 + 1];
            qs[wave * HD + 2 * lane] = q0 * c - q1 * s;
            qs[wave * HD + 2 * lane + 1] = q1 * c + q0 * s;
        }
        __syncthreads();
        float mx = -INFINITY;
        for (int j = lane; j <= r; j += 64) {
            float acc = 0.f;
#pragma unroll 16
            for (int d = 0; d < HD; ++d) acc = fmaf(qs[wave * HD + d], Ks[j * LDK + d], acc);
            acc *= scale;
            ps[wave * T + j] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j <= r; j += 64) {
            const float e = expf(ps[wave * T + j] - mx);
            ps[wave * T + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __syncthreads();
        float acc = 0.f;
        for (int j = 0; j <= r; ++j) acc = fmaf(ps[wave * T + j], Vs[j * LDK + lane], acc);
        out[((long)b * T + r) * D + h * HD + lane] = acc / sum;
        __syncthreads();
    }
}
// MFMA formulation for windows of up to 128 tokens (the streaming configurations): S = Q K^T and O = P V run on
// v_mfma_f32_16x16x4_f32 (fp32 operands: the BSQ bits downstream need fp32).  One workgroup per (head, stream, query
// split); K (RoPE applied) and V^T of the head live in LDS with 16-byte-aligned, bank-rotating row strides; each wave owns
// one 16-row query tile at a time: Q fragments come straight from global memory (RoPE in registers), the 16 x T score tile
// stays in accumulators, softmax statistics are reduced over the 16 lanes of an accumulator row with DPP, P goes through a
// wave-private LDS slab into the A operand of the second product.  Causal tiles above the diagonal are skipped.
typedef float attn_f32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ float row16_xor(float v) {       // value of lane ^ OFF inside a 16-lane row
    if constexpr (OFF == 1) return dpp_mov<0xB1>(v);
    else if constexpr (OFF == 2) return dpp_mov<0x4E>(v);
    else if constexpr (OFF == 8) return dpp_mov<0x128>(v);                    // row_ror:8
    else {                                                                    // 4: row_shl / row_shr by 4, picked per lane
        const float up = dpp_mov<0x104>(v), dn = dpp_mov<0x114>(v);
        return (threadIdx.x & 4) ? dn : up;
    }
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, row16_xor<1>(v)); v = fmaxf(v, row16_xor<2>(v)); v = fmaxf(v, row16_xor<4>(v)); v = fmaxf(v, row16_xor<8>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += row16_xor<1>(v); v += row16_xor<2>(v); v += row16_xor<4>(v); v += row16_xor<8>(v);
    return v;
}

// NWV waves per workgroup: 8 when the window has more than four query tiles -- one wave per SIMD (the four-wave form with its 102 KiB of LDS per
// workgroup) leaves the matrix pipe waiting on every LDS read; eight waves take one tile each instead of two and load K / V twice as fast
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void enc_attention_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ rope,
                                                                 int T, int H, float* __restrict__ out, int row0,
                                                                 unsigned short* __restrict__ outp, long op_pstride, int op_planes, long op_rows) {
    constexpr int HD = 64, LK = HD + 4, NTM = 8;          // up to 8 key tiles (T <= 128)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LV = T + 4;
    float* Ks = smem;                                     // [T][68]    K with RoPE
    float* Vt = Ks + (long)T * LK;                        // [64][T+4]  V transposed
    float* Ps = Vt + (long)HD * LV;                       // [NWV waves][16][T+4]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int D = H * HD;
    const float* base = qkv + (long)b * T * 3 * D;
    for (int idx = tid; idx < T * (HD / 2); idx += 64 * NWV) {
        const int t = idx / (HD / 2), p = idx - t * (HD / 2);
        const float* row = base + (long)t * 3 * D + h * HD + 2 * p;
        const float2 kk = *reinterpret_cast<const float2*>(row + D);
        const float2 vv = *reinterpret_cast<const float2*>(row + 2 * D);
        const float2 cs = *reinterpret_cast<const float2*>(rope + ((long)t * (HD / 2) + p) * 2);
        *reinterpret_cast<float2*>(Ks + t * LK + 2 * p) = make_float2(kk.x * cs.x - kk.y * cs.y, kk.y * cs.x + kk.x * cs.y);
        Vt[(2 * p) * LV + t] = vv.x;
        Vt[(2 * p + 1) * LV + t] = vv.y;
    }
    __syncthreads();
    float* Pw = Ps + (long)wave * 16 * LV;
    const int n_tiles = T / 16;
    const int first_tile = row0 / 16;
    // query tiles first_tile .. n_tiles-1, dealt to (query split, wave) round-robin from the heaviest (last) tile down
    for (int u = blockIdx.z * NWV + wave; u < n_tiles - first_tile; u += NWV * gridDim.z) {
        const int rt = n_tiles - 1 - u;
        const int r0 = rt * 16;
        // Q fragments with RoPE: lane (fr, fk) holds dims 16*blk + 4*fk .. +3 of row r0 + fr
        float4 qa[4];
        {
            const float* qrow = base + (long)(r0 + fr) * 3 * D + h * HD + 4 * fk;
            const float* rrow = rope + (long)(r0 + fr) * HD + 4 * fk;        // (cos, sin) pairs: 2 floats per dim pair
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const float4 q = *reinterpret_cast<const float4*>(qrow + 16 * blk);
                const float4 cs = *reinterpret_cast<const float4*>(rrow + 16 * blk);      // c0 s0 c1 s1 of pairs (4fk+16blk)/2, +1
                qa[blk] = make_float4(q.x * cs.x - q.y * cs.y, q.y * cs.x + q.x * cs.y, q.z * cs.z - q.w * cs.w, q.w * cs.z + q.z * cs.w);
            }
        }
        attn_f32x4 s[NTM];
#pragma unroll
        for (int kt = 0; kt < NTM; ++kt) {
            s[kt] = (attn_f32x4){0.f, 0.f, 0.f, 0.f};
            if (kt <= rt) {
                const float* kp = Ks + (kt * 16 + fr) * LK + 4 * fk;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const float4 kb = *reinterpret_cast<const float4*>(kp + 16 * blk);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].x, kb.x, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].y, kb.y, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].z, kb.z, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].w, kb.w, s[kt], 0, 0, 0);
                }
            }
        }
        // accumulator element r of a lane: row 4*fk + r, key 16*kt + fr.  Scale, causal mask on the diagonal tile, softmax.
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NTM; ++kt)
            if (kt <= rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[kt][r] * 0.125f;
                    if (kt == rt && fr > 4 * fk + r) v = -INFINITY;
                    s[kt][r] = v;
                    mx[r] = fmaxf(mx[r], v);
                }
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = row16_max(mx[r]);
#pragma unroll
        for (int kt = 0; kt < NTM; ++kt)
            if (kt <= rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = expf(s[kt][r] - mx[r]);
                    sum[r] += e;
                    Pw[(4 * fk + r) * LV + kt * 16 + fr] = e;
                }
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = row16_sum(sum[r]);
        __builtin_amdgcn_wave_barrier();
        attn_f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (attn_f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt <= rt; ++kt) {
            const float4 pa = *reinterpret_cast<const float4*>(Pw + fr * LV + kt * 16 + 4 * fk);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const float4 vb = *reinterpret_cast<const float4*>(Vt + (dt * 16 + fr) * LV + kt * 16 + 4 * fk);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, vb.x, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, vb.y, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, vb.z, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, vb.w, o[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv = 1.f / sum[r];
            if (outp) {          // the output projection takes its A operand as K-blocked planes (planes_split.h)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    unsigned short hi, lo;
                    h3_split(o[dt][r] * inv, hi, lo);
                    const long po = plane_off_blocked((long)b * T + r0 + 4 * fk + r, h * HD + dt * 16 + fr, op_rows);
                    outp[po] = hi;
                    if (op_planes > 1) outp[op_pstride + po] = lo;
                }
                continue;
            }
            float* orow = out + ((long)b * T + r0 + 4 * fk + r) * D + h * HD + fr;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) orow[dt * 16] = o[dt][r] * inv;
        }
        __builtin_amdgcn_wave_barrier();      // the next tile of this wave overwrites Pw
    }
}

// Tiled (flash-style) formulation for whole-utterance encodes: any T, keys visited in blocks of 64 with an online softmax, so
// LDS holds one K / V block instead of the whole sequence.  Causal + window-limited exactly as WindowLimitedTransformer builds
// its mask (modules/vqgan/windowed_transformer.py:291-304: keys max(0, r - window + 1) .. r; window_size 512 in the tokenizer's
// YAML -- irrelevant for the <= 256-token streaming windows, binding for utterances beyond 512 tokens = 23.8 s).  One workgroup =
// 4 consecutive query rows (one per wave) of one (head, stream); lane = key inside a block for the scores, lane = head
// dimension for P.V.  Exact fp32 (the BSQ bits downstream need it).
__global__ __launch_bounds__(256) void enc_attention_flash_kernel(const float* __restrict__ qkv, const float* __restrict__ rope, int T, int H, int window,
                                                                  float* __restrict__ out, int row0) {
    constexpr int HD = 64, LDK = HD + 1;
    __shared__ float Ks[64 * LDK], Vs[64 * LDK], qs[4 * HD], ps[4 * 64];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = H * HD;
    const float* base = qkv + (long)b * T * 3 * D;
    const int r0 = row0 + blockIdx.z * 4;
    const int r = min(r0 + wave, T - 1);
    if (lane < HD / 2) {
        const float* row = base + (long)r * 3 * D + h * HD;
        const float q0 = row[2 * lane], q1 = row[2 * lane + 1];
        const float c = rope[(r * (HD / 2) + lane) * 2], sn = rope[(r * (HD / 2) + lane) * 2 + 1];
        qs[wave * HD + 2 * lane] = (q0 * c - q1 * sn) * 0.125f;
        qs[wave * HD + 2 * lane + 1] = (q1 * c + q0 * sn) * 0.125f;
    }
    const int klo = max(0, r - window + 1);                       // this row's first key
    const int rmax = min(r0 + 3, T - 1);
    const int blk_lo = max(0, r0 - window + 1) / 64 * 64;
    float m = -INFINITY, l = 0.f, o = 0.f;
    for (int j0 = blk_lo; j0 <= rmax; j0 += 64) {
        __syncthreads();                                          // the previous block is consumed (and qs is written, first pass)
        for (int idx = tid; idx < 64 * (HD / 2); idx += 256) {
            const int jj = idx / (HD / 2), p = idx - jj * (HD / 2);
            const int t = min(j0 + jj, T - 1);
            const float* row = base + (long)t * 3 * D;
            const float k0 = row[D + h * HD + 2 * p], k1 = row[D + h * HD + 2 * p + 1];
            const float c = rope[(t * (HD / 2) + p) * 2], sn = rope[(t * (HD / 2) + p) * 2 + 1];
            Ks[jj * LDK + 2 * p] = k0 * c - k1 * sn;
            Ks[jj * LDK + 2 * p + 1] = k1 * c + k0 * sn;
            Vs[jj * LDK + 2 * p] = row[2 * D + h * HD + 2 * p];
            Vs[jj * LDK + 2 * p + 1] = row[2 * D + h * HD + 2 * p + 1];
        }
        __syncthreads();
        const int j = j0 + lane;
        const bool valid = j >= klo && j <= r;
        float sc = -INFINITY;
        if (valid) {
            float acc = 0.f;
#pragma unroll 16
            for (int d = 0; d < HD; ++d) acc = fmaf(qs[wave * HD + d], Ks[lane * LDK + d], acc);
            sc = acc;
        }
        const float bm = wave_max(sc);
        if (bm > -INFINITY) {                                     // (uniform per wave) at least one key of the block is visible to this row
            const float mn = fmaxf(m, bm);
            const float corr = expf(m - mn), pj = valid ? expf(sc - mn) : 0.f;
            ps[wave * 64 + lane] = pj;
            l = l * corr + wave_sum(pj);
            float acc = 0.f;
#pragma unroll 16
            for (int jj = 0; jj < 64; ++jj) acc = fmaf(ps[wave * 64 + jj], Vs[jj * LDK + lane], acc);
            o = o * corr + acc;
            m = mn;
        }
    }
    if (r0 + wave < T) out[((long)b * T + r) * D + h * HD + lane] = o / l;
}

bool enc_attention_can_write_planes(int T) { return T % 16 == 0 && T <= 128; }
int launch_enc_attention(const float* qkv, const float* rope, int B, int T, int H, int hd, float* out, int row0, hipStream_t st,
                         unsigned short* outp, long op_pstride, int op_planes, long op_rows) {
    SVA_CHECK(hd == 64 && T % 4 == 0, "enc_attention: head_dim must be 64 and T a multiple of 4");
    SVA_CHECK(!outp || enc_attention_can_write_planes(T), "enc_attention: planes output only from the <= 128-token MFMA kernel");
    if (T % 16 == 0 && T <= 128) {
        const int tiles = T / 16 - row0 / 16;
        const int nwv = tiles > 4 ? 8 : 4;
        const size_t sm = ((size_t)T * 68 + 64 * (size_t)(T + 4) + (size_t)nwv * 16 * (size_t)(T + 4)) * sizeof(float);
        static DeviceOnce attr_m;
        if (attr_m.needed()) {
            SVA_HIP(hipFuncSetAttribute((const void*)enc_attention_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            SVA_HIP(hipFuncSetAttribute((const void*)enc_attention_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_m.done();
        }
        int qsplit = 1;                                    // more workgroups per (head, stream) while the chip is under-filled
        while (qsplit * nwv < tiles && (long)H * B * qsplit < 128) qsplit *= 2;
        if (nwv == 8) hipLaunchKernelGGL(enc_attention_mfma_kernel<8>, dim3(H, B, qsplit), dim3(512), sm, st, qkv, rope, T, H, out, row0, outp, op_pstride, op_planes, op_rows);
        else hipLaunchKernelGGL(enc_attention_mfma_kernel<4>, dim3(H, B, qsplit), dim3(256), sm, st, qkv, rope, T, H, out, row0, outp, op_pstride, op_planes, op_rows);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    const size_t smem = ((size_t)T * 65 * 2 + 4 * 64 + 4 * (size_t)T) * sizeof(float);
    if (smem > 160 * 1024 || T > 512) {      // whole-utterance encodes: tiled kernel with the transformer's 512-key causal window
        hipLaunchKernelGGL(enc_attention_flash_kernel, dim3(H, B, (T - row0 + 3) / 4), dim3(256), 0, st, qkv, rope, T, H, 512, out, row0);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    static DeviceOnce attr_set;
    if (attr_set.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)enc_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    // rows row0..T-1 are produced; the loop trip count must be uniform over the workgroup's 4 waves (barriers inside),
    // so a partial row range is walked with one row per workgroup-wave slot (rows >= T fall out of the loop together
    // only when (T - row0) is a multiple of 4 * qs; otherwise use qs = 1 and pad by predication below)
    int qs = 1;
    if (row0 == 0) {
        while (qs < 8 && T % (8 * qs) == 0 && (long)H * B * qs < 256) qs *= 2;
    } else {
        SVA_CHECK((T - row0) <= 4 || (T - row0) % 4 == 0, "enc_attention: partial row range must be <= 4 rows or a multiple of 4");
    }
    hipLaunchKernelGGL(enc_attention_kernel, dim3(H, B, qs), dim3(256), smem, st, qkv, rope, T, H, out, row0);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// E8 BSQ (modules/vqgan/modules/bsq.py:330-369): Linear C->nbits (+bias), L2-normalise,
// bit_d = u_d > 0, index = sum bit_d << (nbits-1-d) (MSB first, mask buffer :230).
// ------------------------------------------------------------------------------------------
// The quantizer's final RMSNorm (pre_module.norm, windowed_transformer.py:248-259) is applied here on the fly
// (norm_w != null): v_c = x_c * rsqrt(mean(x^2) + eps) * w_c, optionally stored to zn_out.
// One wave per row; the row lives in registers (C <= 512) and the nbits dot products are independent.
template <int NPL, int NB>
__global__ __launch_bounds__(256) void bsq_kernel(const float* __restrict__ z, long z_bstride, long z_off, int ldz, int T,
                                                  int rows, const float* __restrict__ norm_w, float eps,
                                                  float* __restrict__ zn_out, const float* __restrict__ W,
                                                  const float* __restrict__ bias, long long* __restrict__ idx_out,
                                                  int idx_bstride, int idx_off, float* __restrict__ u_out) {
    constexpr int C = 64 * NPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / T, t = row - b * T;
    const float* zr = z + (long)b * z_bstride + z_off + (long)t * ldz;
    float v[NPL], w[NB][NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = zr[lane + 64 * i];
#pragma unroll
    for (int d = 0; d < NB; ++d)
#pragma unroll
        for (int i = 0; i < NPL; ++i) w[d][i] = W[d * C + lane + 64 * i];        // all loads in flight before the first use
    if (norm_w) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) ss = fmaf(v[i], v[i], ss);
        const float inv = 1.f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] = v[i] * inv * norm_w[lane + 64 * i];
        if (zn_out) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) zn_out[(long)b * z_bstride + z_off + (long)t * ldz + lane + 64 * i] = v[i];
        }
    }
    float u[NB], nrm = 0.f;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) acc = fmaf(w[d][i], v[i], acc);
        u[d] = wave_sum(acc) + bias[d];
        nrm = fmaf(u[d], u[d], nrm);
    }
    if (lane == 0) {
        long long idx = 0;
        const float inv = 1.f / fmaxf(sqrtf(nrm), 1e-12f);      // F.normalize eps
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            if (u[d] > 0.f) idx |= 1ll << (NB - 1 - d);
            if (u_out) u_out[((long)b * idx_bstride + idx_off + t) * NB + d] = u[d] * inv;
        }
        idx_out[(long)b * idx_bstride + idx_off + t] = idx;
    }
}
int launch_bsq(const float* z, long z_bstride, long z_off, int ldz, int B, int T, int C, const float* norm_w, float eps,
               float* zn_out, const float* W, const float* bias, int nbits, long long* idx_out, int idx_bstride, int idx_off,
               float* u_out, hipStream_t st) {
    SVA_CHECK(nbits == 13 && C == 512, "bsq: built for the reference's 13 bits on 512 channels");
    const int rows = B * T;
    hipLaunchKernelGGL((bsq_kernel<8, 13>), dim3((rows + 3) / 4), dim3(256), 0, st, z, z_bstride, z_off, ldz, T, rows, norm_w, eps, zn_out,
                       W, bias, idx_out, idx_bstride, idx_off, u_out);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// A2: RoPE (adjacent pairs, bf16-rounded table; modules/dual_ar_stream.py:1004-1016) on q,k
// and KV-cache write at kv_pos (KVCache.update :141-150).  One workgroup per row.
// ------------------------------------------------------------------------------------------
template <typename KV> __device__ __forceinline__ KV to_kv(float v);
template <> __device__ __forceinline__ float to_kv<float>(float v) { return v; }
template <> __device__ __forceinline__ __half to_kv<__half>(float v) { return __float2half(v); }
__device__ __forceinline__ float from_kv(float v) { return v; }
__device__ __forceinline__ float from_kv(__half v) { return __half2float(v); }

template <typename KV>
__global__ __launch_bounds__(256) void rope_kvwrite_kernel(float* __restrict__ qkv, int H, int hd, const int* __restrict__ slot,
                                                           const int* __restrict__ pos, const float* __restrict__ rope,
                                                           KV* __restrict__ cache, long slot_stride, int S) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int D = H * hd, hp = hd / 2;
    float* row = qkv + (long)m * 3 * D;
    const int p = pos[m];
    KV* kc = cache + (long)slot[m] * slot_stride;
    KV* vc = kc + (long)H * S * hd;
    for (int i = tid; i < D / 2; i += 256) {
        const int h = i / hp, j = i - h * hp;
        const float c = rope[((long)p * hp + j) * 2], s = rope[((long)p * hp + j) * 2 + 1];
        const float q0 = row[2 * i], q1 = row[2 * i + 1];
        row[2 * i] = q0 * c - q1 * s;
        row[2 * i + 1] = q1 * c + q0 * s;
        const float k0 = row[D + 2 * i], k1 = row[D + 2 * i + 1];
        const long o = ((long)h * S + p) * hd + 2 * j;
        kc[o] = to_kv<KV>(k0 * c - k1 * s);
        kc[o + 1] = to_kv<KV>(k1 * c + k0 * s);
        vc[o] = to_kv<KV>(row[2 * D + 2 * i]);
        vc[o + 1] = to_kv<KV>(row[2 * D + 2 * i + 1]);
    }
}
template <typename KV>
int launch_rope_kvwrite(float* qkv, int M, int H, int hd, const int* slot, const int* pos, const float* rope, KV* cache,
                        long slot_stride, int S, hipStream_t st) {
    hipLaunchKernelGGL((rope_kvwrite_kernel<KV>), dim3(M), dim3(256), 0, st, qkv, H, hd, slot, pos, rope, cache, slot_stride, S);
    SVA_HIP(hipGetLastError());
    return 0;
}
template int launch_rope_kvwrite<float>(float*, int, int, int, const int*, const int*, const float*, float*, long, int, hipStream_t);
template int launch_rope_kvwrite<__half>(float*, int, int, int, const int*, const int*, const float*, __half*, long, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// A2 decode attention (Attention.forward, modules/dual_ar_stream.py:895-936 with the mask
// causal_mask[kv_pos, :max_seq_len] of :333): query row m attends cache slots 0..pos[m].
// One workgroup per (head, row): scores in LDS, block softmax, V reduction split over waves.
// ------------------------------------------------------------------------------------------
template <typename KV>
__global__ __launch_bounds__(256) void ar_attention_kernel(const float* __restrict__ qkv, int H, const int* __restrict__ slot,
                                                           const int* __restrict__ pos, const KV* __restrict__ cache,
                                                           long slot_stride, int S, float* __restrict__ out,
                                                           float* __restrict__ partial) {
    // gridDim.z > 1 (split-key decode, `partial` != null): workgroup z covers keys [z*cl, (z+1)*cl) and writes the
    // UNnormalised P.V, its score maximum and its exp-sum to partial[((m*H + h)*gridDim.z + z)*68 + {0..63, 64, 65}];
    // the consumer (gemv mode 4) merges the splits.
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                 // [S] scores / probabilities
    float* red = sc + S;              // [8]
    float* part = red + 8;            // [16][64] partial P.V sums
    const int h = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int D = H * HD;
    int L = pos[m] + 1;
    const KV* kc = cache + (long)slot[m] * slot_stride + (long)h * S * HD;
    const KV* vc = kc + (long)H * S * HD;
    if (partial) {
        const int cl = (((L + (int)gridDim.z - 1) / (int)gridDim.z) + 15) & ~15;
        const int j_lo = min((int)blockIdx.z * cl, L);
        L = min(L - j_lo, cl);                  // keys of this split, re-based to 0
        kc += (long)j_lo * HD;
        vc += (long)j_lo * HD;
    }
    // scores: 4 lanes per key row (16 dims each) -> a wave reads 16 consecutive rows = 4 KiB contiguous
    const int part4 = tid & 3;
    float q[16];
    {
        const float* qr = qkv + (long)m * 3 * D + h * HD + part4 * 16;
#pragma unroll
        for (int d = 0; d < 16; ++d) q[d] = qr[d];
    }
    float mx = -INFINITY;
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int j = j0 + (tid >> 2);
        float acc = 0.f;
        if (j < L) {
            const KV* kr = kc + (long)j * HD + part4 * 16;
#pragma unroll
            for (int d = 0; d < 16; ++d) acc = fmaf(q[d], from_kv(kr[d]), acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc *= 0.125f;
        if (j < L) {
            if (part4 == 0) sc[j] = acc;
            mx = fmaxf(mx, acc);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < L; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    // P.V: thread = (row group jg of 16, float4 column c4 of 16) -> a wave reads 4 consecutive V rows = 1 KiB
    const int c4 = tid & 15, jg = tid >> 4;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = jg; j < L; j += 16) {
        const KV* vr = vc + (long)j * HD + c4 * 4;
        const float pj = sc[j];
        a4.x = fmaf(pj, from_kv(vr[0]), a4.x);
        a4.y = fmaf(pj, from_kv(vr[1]), a4.y);
        a4.z = fmaf(pj, from_kv(vr[2]), a4.z);
        a4.w = fmaf(pj, from_kv(vr[3]), a4.w);
    }
    *reinterpret_cast<float4*>(&part[jg * HD + c4 * 4]) = a4;
    __syncthreads();
    if (tid < HD) {
        float o = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) o += part[g2 * HD + tid];
        if (partial) {
            float* pw = partial + (((long)m * H + h) * gridDim.z + blockIdx.z) * 68;
            pw[tid] = o;
            if (tid == 0) { pw[64] = mx; pw[65] = sum; }
        } else {
            out[(long)m * D + h * HD + tid] = o / sum;
        }
    }
}
template <typename KV>
int launch_ar_attention(const float* qkv, int M, int H, int hd, const int* slot, const int* pos, const KV* cache,
                        long slot_stride, int S, float* out, hipStream_t st, float* partial, int splits) {
    SVA_CHECK(hd == 64, "ar_attention: head_dim must be 64");
    const size_t smem = ((size_t)S + 8 + 16 * 64) * sizeof(float);
    hipLaunchKernelGGL((ar_attention_kernel<KV>), dim3(H, M, partial ? splits : 1), dim3(256), smem, st, qkv, H, slot, pos, cache, slot_stride,
                       S, out, partial);
    SVA_HIP(hipGetLastError());
    return 0;
}
// The decode frame's slow layers present every stream's TWO new rows (cached audio embedding, content token: positions p, p + 1 of one slot,
// rows 2 s and 2 s + 1).  One workgroup per (head, stream) serves both from ONE pass over the slot's K / V rows -- the per-row kernel above read
// them twice (PMC, 64 streams: 289 MB per launch against 118 MB of cached keys and values).  Each row's arithmetic -- the thread that owns a
// key, the order of every sum -- is the per-row kernel's, so the results are bit-identical to it.
template <typename KV>
__global__ __launch_bounds__(256) void ar_attention_pair_kernel(const float* __restrict__ qkv, int H, const int* __restrict__ slot,
                                                                const int* __restrict__ pos, const KV* __restrict__ cache,
                                                                long slot_stride, int S, float* __restrict__ out) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc0 = smem;                // [S] scores / probabilities of row 0
    float* sc1 = sc0 + S;             // [S] ... of row 1
    float* red = sc1 + S;             // [16]
    float* part = red + 16;           // [2][16][64] partial P.V sums
    const int h = blockIdx.x, m0 = 2 * blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int D = H * HD;
    const int L0 = pos[m0] + 1, L1 = pos[m0 + 1] + 1;          // L1 = L0 + 1 (the launcher's contract; the loops below only need L1 >= L0)
    const KV* kc = cache + (long)slot[m0] * slot_stride + (long)h * S * HD;
    const KV* vc = kc + (long)H * S * HD;
    const int part4 = tid & 3;
    float q0[16], q1[16];
    {
        const float* qr = qkv + (long)m0 * 3 * D + h * HD + part4 * 16;
#pragma unroll
        for (int d = 0; d < 16; ++d) { q0[d] = qr[d]; q1[d] = qr[3 * D + d]; }
    }
    float mx0 = -INFINITY, mx1 = -INFINITY;
    for (int j0 = 0; j0 < L1; j0 += 64) {
        const int j = j0 + (tid >> 2);
        float a0 = 0.f, a1 = 0.f;
        if (j < L1) {
            const KV* kr = kc + (long)j * HD + part4 * 16;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                // Two explicit v_fmac_f32 per key element.  Written as fmaf() pairs, the compiler packs the two rows into one chain of v_pk_fma_f32
                // whose odd terms read the key through op_sel:[0,1,0]; builds of THIS kernel with that chain (compiler-made or hand-written) returned
                // wrong low-lane sums in lanes 48..63 of a wave for ~1 % of the (head, stream) workgroups of a loaded 64-stream decode -- never in an
                // isolated launch, and not in a stand-alone probe of the instruction form (tools/probes/pk_opsel_probe.hip) -- while the packed chain
                // without the op_sel read and this form are exact in the same place (the in-situ bisection).
                const float kv = from_kv(kr[d]);
                asm("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(q0[d]), "v"(kv));
                asm("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(q1[d]), "v"(kv));
            }
        }
        a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
        a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64);
        a0 *= 0.125f; a1 *= 0.125f;
        if (j < L0) {
            if (part4 == 0) sc0[j] = a0;
            mx0 = fmaxf(mx0, a0);
        }
        if (j < L1) {
            if (part4 == 0) sc1[j] = a1;
            mx1 = fmaxf(mx1, a1);
        }
    }
    mx0 = wave_max(mx0); mx1 = wave_max(mx1);
    if (lane == 0) { red[wave] = mx0; red[8 + wave] = mx1; }
    __syncthreads();
    mx0 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    mx1 = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
    float sum0 = 0.f, sum1 = 0.f;
    for (int j = tid; j < L1; j += 256) {
        if (j < L0) { const float e = expf(sc0[j] - mx0); sc0[j] = e; sum0 += e; }
        const float e1 = expf(sc1[j] - mx1); sc1[j] = e1; sum1 += e1;
    }
    sum0 = wave_sum(sum0); sum1 = wave_sum(sum1);
    if (lane == 0) { red[4 + wave] = sum0; red[12 + wave] = sum1; }
    __syncthreads();
    sum0 = red[4] + red[5] + red[6] + red[7];
    sum1 = red[12] + red[13] + red[14] + red[15];
    const int c4 = tid & 15, jg = tid >> 4;
    float4 u4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = jg; j < L1; j += 16) {
        const KV* vr = vc + (long)j * HD + c4 * 4;
        const float v0 = from_kv(vr[0]), v1 = from_kv(vr[1]), v2 = from_kv(vr[2]), v3 = from_kv(vr[3]);
        if (j < L0) {
            const float pj = sc0[j];
            u4.x = fmaf(pj, v0, u4.x); u4.y = fmaf(pj, v1, u4.y); u4.z = fmaf(pj, v2, u4.z); u4.w = fmaf(pj, v3, u4.w);
        }
        const float pk = sc1[j];
        w4.x = fmaf(pk, v0, w4.x); w4.y = fmaf(pk, v1, w4.y); w4.z = fmaf(pk, v2, w4.z); w4.w = fmaf(pk, v3, w4.w);
    }
    *reinterpret_cast<float4*>(&part[jg * HD + c4 * 4]) = u4;
    *reinterpret_cast<float4*>(&part[16 * HD + jg * HD + c4 * 4]) = w4;
    __syncthreads();
    if (tid < 2 * HD) {
        const int r = tid >> 6, t = tid & 63;
        float o = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) o += part[r * 16 * HD + g2 * HD + t];
        out[(long)(m0 + r) * D + h * HD + t] = o / (r ? sum1 : sum0);
    }
}
// rows (2 s, 2 s + 1) = positions (p, p + 1) of slot s for every s (M even): the decode frame's slow layers (stages.hip)
template <typename KV>
int launch_ar_attention_pairs(const float* qkv, int M, int H, int hd, const int* slot, const int* pos, const KV* cache, long slot_stride, int S, float* out,
                              hipStream_t st) {
    SVA_CHECK(hd == 64 && M % 2 == 0, "ar_attention_pairs: head_dim 64, an even number of rows");
    const size_t smem = ((size_t)2 * S + 16 + 2 * 16 * 64) * sizeof(float);
    static DeviceOnce attr;
    if (attr.needed() && smem > 48 * 1024) {
        SVA_HIP(hipFuncSetAttribute((const void*)ar_attention_pair_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        SVA_HIP(hipFuncSetAttribute((const void*)ar_attention_pair_kernel<__half>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr.done();
    }
    hipLaunchKernelGGL((ar_attention_pair_kernel<KV>), dim3(H, M / 2), dim3(256), smem, st, qkv, H, slot, pos, cache, slot_stride, S, out);
    SVA_HIP(hipGetLastError());
    return 0;
}
template int launch_ar_attention_pairs<float>(const float*, int, int, int, const int*, const int*, const float*, long, int, float*, hipStream_t);
template int launch_ar_attention_pairs<__half>(const float*, int, int, int, const int*, const int*, const __half*, long, int, float*, hipStream_t);
template int launch_ar_attention<float>(const float*, int, int, int, const int*, const int*, const float*, long, int, float*, hipStream_t, float*, int);
template int launch_ar_attention<__half>(const float*, int, int, int, const int*, const int*, const __half*, long, int, float*, hipStream_t, float*, int);

// ------------------------------------------------------------------------------------------
// Prefill attention (prompt prefill / re-prefill / offline generate: hundreds of query rows at CONSECUTIVE positions of ONE slot,
// modules/dual_ar_stream.py:764-796 -> forward_generate :338-356 with causal_mask[kv_pos]): flash-style on the matrix pipes.
// One workgroup = ONE 16-row query tile of one head (12 heads x M / 16 tiles: 240 workgroups at M = 314); its four waves split
// the tile's KEY blocks of 64 round-robin (flash-decoding style), each with its own online softmax, and merge their (max, sum,
// P.V) triples through LDS at the end.  Scores Q.K^T and P.V are v_mfma_f32_16x16x4_f32 (fp32 operands and accumulation, the
// arithmetic class of the per-row kernel above); K and V fragments come straight from the slot's cache in global memory (the
// cache rows are the B operands: no staging, no workgroup barrier in the key loop), row statistics are reduced over the 16
// lanes of an accumulator row with DPP, P goes through a wave-private LDS slab into the A operand of the second product.
// (A first version with 64 query rows per workgroup and LDS-staged K / V^T was no faster than the per-row kernel below M = 400:
// 60 workgroups, every block a load -> barrier -> compute round trip.)
// ------------------------------------------------------------------------------------------
template <typename KV> __device__ __forceinline__ float4 ld_kv_f4(const KV* p);
template <> __device__ __forceinline__ float4 ld_kv_f4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld_kv_f4<__half>(const __half* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
}

template <typename KV>
__global__ __launch_bounds__(256) void ar_prefill_attention_kernel(const float* __restrict__ qkv, int M, int H, int pos0,
                                                                   const KV* __restrict__ cache_slot, int S, float* __restrict__ out) {
    constexpr int HD = 64, KB = 64, LV = KB + 4;
    __shared__ __attribute__((aligned(16))) float Ps[4 * 16 * LV];        // per wave [row][key]
    __shared__ float Ms[4 * 16], Ls[4 * 16];                              // per wave row max / exp-sum
    __shared__ __attribute__((aligned(16))) float Os[4 * 16 * (HD + 4)];  // per wave unnormalised P.V
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int D = H * HD;
    const KV* kc = cache_slot + (long)h * S * HD;
    const KV* vc = kc + (long)H * S * HD;
    const int q0 = blockIdx.y * 16;
    const int L = pos0 + M;                                               // keys of this pass: 0 .. L-1
    const int last_key = min(q0 + 15, M - 1) + pos0;                      // last key any row of the tile may see
    // Q fragments: lane (fr, fk) holds dims 16 blk + 4 fk .. +3 of row q0 + fr (RoPE already applied in place)
    float4 qa[4];
    {
        const int m = min(q0 + fr, M - 1);
        const float* qrow = qkv + (long)m * 3 * D + h * HD + 4 * fk;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) qa[blk] = *reinterpret_cast<const float4*>(qrow + 16 * blk);
    }
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f};
    attn_f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (attn_f32x4){0.f, 0.f, 0.f, 0.f};
    float* Pw = Ps + wave * 16 * LV;
    for (int j0 = wave * KB; j0 <= last_key; j0 += 4 * KB) {
        // K fragments of the block's four key tiles, then V: everything in flight before the first MFMA
        float4 kf[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const long j = min(j0 + kt * 16 + fr, L - 1);
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) kf[kt][blk] = ld_kv_f4<KV>(kc + j * HD + 16 * blk + 4 * fk);
        }
        float vf[4][4][4];                                                // [key tile][dim tile][key 4 fk + c]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long j = min(j0 + kt * 16 + 4 * fk + c, L - 1);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[kt][dt][c] = from_kv(vc[j * HD + dt * 16 + fr]);
            }
        attn_f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (attn_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].x, kf[kt][blk].x, s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].y, kf[kt][blk].y, s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].z, kf[kt][blk].z, s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[blk].w, kf[kt][blk].w, s[kt], 0, 0, 0);
            }
        }
        // accumulator element r of a lane: query row q0 + 4 fk + r (position pos0 + that), key j0 + 16 kt + fr
        float bm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][r] * 0.125f;
                if (j0 + kt * 16 + fr > pos0 + q0 + 4 * fk + r) v = -INFINITY;
                s[kt][r] = v;
                bm[r] = fmaxf(bm[r], v);
            }
        float corr[4], bs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mn = fmaxf(mx[r], row16_max(bm[r]));
            // a row may see none of this wave's keys so far (its diagonal lies below the block): keep exp() away from inf - inf
            corr[r] = mn == -INFINITY ? 1.f : expf(mx[r] - mn);
            mx[r] = mn;
            bs[r] = 0.f;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = mx[r] == -INFINITY ? 0.f : expf(s[kt][r] - mx[r]);
                bs[r] += e;
                Pw[(4 * fk + r) * LV + kt * 16 + fr] = e;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = sum[r] * corr[r] + row16_sum(bs[r]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= corr[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float4 pa = *reinterpret_cast<const float4*>(Pw + fr * LV + kt * 16 + 4 * fk);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, vf[kt][dt][0], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, vf[kt][dt][1], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, vf[kt][dt][2], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, vf[kt][dt][3], o[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                                  // Pw is rewritten by this wave's next block
    }
    // merge the four waves' partial softmaxes (wave 0 always holds block 0, whose key 0 every row sees: the merged max is finite)
    if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { Ms[wave * 16 + 4 * fk + r] = mx[r]; Ls[wave * 16 + 4 * fk + r] = sum[r]; }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Os[(wave * 16 + 4 * fk + r) * (HD + 4) + dt * 16 + fr] = o[dt][r];
    __syncthreads();
    for (int idx = tid; idx < 16 * (HD / 4); idx += 256) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        const int m = q0 + row;
        if (m >= M) continue;
        const float m0 = Ms[row], m1 = Ms[16 + row], m2 = Ms[32 + row], m3 = Ms[48 + row];
        const float mm = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float w0 = expf(m0 - mm), w1 = m1 == -INFINITY ? 0.f : expf(m1 - mm), w2 = m2 == -INFINITY ? 0.f : expf(m2 - mm),
                    w3 = m3 == -INFINITY ? 0.f : expf(m3 - mm);
        const float inv = 1.f / (Ls[row] * w0 + Ls[16 + row] * w1 + Ls[32 + row] * w2 + Ls[48 + row] * w3);
        const float4 a = *reinterpret_cast<const float4*>(Os + row * (HD + 4) + c4);
        const float4 b = *reinterpret_cast<const float4*>(Os + (16 + row) * (HD + 4) + c4);
        const float4 c = *reinterpret_cast<const float4*>(Os + (32 + row) * (HD + 4) + c4);
        const float4 e = *reinterpret_cast<const float4*>(Os + (48 + row) * (HD + 4) + c4);
        float4 r4;
        r4.x = (a.x * w0 + b.x * w1 + c.x * w2 + e.x * w3) * inv;
        r4.y = (a.y * w0 + b.y * w1 + c.y * w2 + e.y * w3) * inv;
        r4.z = (a.z * w0 + b.z * w1 + c.z * w2 + e.z * w3) * inv;
        r4.w = (a.w * w0 + b.w * w1 + c.w * w2 + e.w * w3) * inv;
        *reinterpret_cast<float4*>(out + (long)m * D + h * HD + c4) = r4;
    }
}
template <typename KV>
int launch_ar_prefill_attention(const float* qkv, int M, int H, int hd, int slot0, int pos0, const KV* cache, long slot_stride, int S, float* out,
                                hipStream_t st) {
    SVA_CHECK(hd == 64 && M >= 1 && pos0 >= 0 && pos0 + M <= S, "ar_prefill_attention: head_dim 64, rows inside the cache");
    hipLaunchKernelGGL((ar_prefill_attention_kernel<KV>), dim3(H, (M + 15) / 16), dim3(256), 0, st, qkv, M, H, pos0, cache + (long)slot0 * slot_stride, S, out);
    SVA_HIP(hipGetLastError());
    return 0;
}
template int launch_ar_prefill_attention<float>(const float*, int, int, int, int, int, const float*, long, int, float*, hipStream_t);
template int launch_ar_prefill_attention<__half>(const float*, int, int, int, int, int, const __half*, long, int, float*, hipStream_t);

// ------------------------------------------------------------------------------------------
// Fast-AR attention for batched decode (cache of <= 8 codebook positions): RoPE on q / k, the KV-cache write and the
// attention itself in one launch, one WAVE per (row, head), lane = head dimension.  Replaces rope_kvwrite + ar_attention
// (two launches of 256-thread workgroups with LDS and barriers for at most 8 keys).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ar_fast_attention_kernel(const float* __restrict__ qkv, int H, int M, const int* __restrict__ slot,
                                                                const int* __restrict__ pos, const float* __restrict__ rope,
                                                                float* __restrict__ cache, long slot_stride, int S, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= M * H) return;
    const int m = unit / H, h = unit - m * H;
    const int D = H * 64, p = pos[m];
    const float* row = qkv + (long)m * 3 * D + h * 64;
    float q = row[lane], k = row[D + lane];
    const float v = row[2 * D + lane];
    {   // adjacent-pair rotation (dual_ar_stream.py:1004-1016): even lane holds x0, odd lane x1 of the pair
        const float c = rope[((long)p * 32 + (lane >> 1)) * 2], sn = rope[((long)p * 32 + (lane >> 1)) * 2 + 1];
        const float qo = dpp_mov<0xB1>(q), ko = dpp_mov<0xB1>(k);           // partner of the pair (lane ^ 1)
        q = (lane & 1) ? q * c + qo * sn : q * c - qo * sn;
        k = (lane & 1) ? k * c + ko * sn : k * c - ko * sn;
    }
    float* kc = cache + (long)slot[m] * slot_stride + (long)h * S * 64;
    float* vc = kc + (long)H * S * 64;
    kc[(long)p * 64 + lane] = k;
    vc[(long)p * 64 + lane] = v;
    float kk[8], vv[8], sc[8], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int jj = j < p ? j : 0;
        kk[j] = j < p ? kc[jj * 64 + lane] : k;           // j == p: this token (just written); j > p: masked below
        vv[j] = j < p ? vc[jj * 64 + lane] : v;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float d = wave_sum(q * kk[j]) * 0.125f;
        sc[j] = j <= p ? d : -INFINITY;
        mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float e = j <= p ? expf(sc[j] - mx) : 0.f;
        sum += e;
        o = fmaf(e, vv[j], o);
    }
    out[(long)m * D + h * 64 + lane] = o / sum;
}
int launch_ar_fast_attention(const float* qkv, int M, int H, const int* slot, const int* pos, const float* rope, float* cache,
                             long slot_stride, int S, float* out, hipStream_t st) {
    SVA_CHECK(S <= 8, "ar_fast_attention: cache of at most 8 positions");
    hipLaunchKernelGGL(ar_fast_attention_kernel, dim3((M * H + 3) / 4), dim3(256), 0, st, qkv, H, M, slot, pos, rope, cache, slot_stride, S, out);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// A5 sampler (modules/dual_ar_stream.py:1092-1132, defaults T = 0.7, top_p = 0.7, no repetition
// penalty): sort descending, inclusive cumsum of softmax, drop every sorted entry with
// cum > top_p except rank 0 (no right shift), divide by max(T, 1e-5), softmax, argmax(p / q),
// q ~ Exp(1).  One 1024-thread workgroup per row; bitonic sort in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sampler_kernel(const float* __restrict__ logits, int V, int ldl, int P,
                                                       const float* __restrict__ noise, int ldn,
                                                       const unsigned long long* __restrict__ seed, const int* __restrict__ frame,
                                                       int kind, int noise_elem_off, float inv_temp, float top_p,
                                                       int* __restrict__ tok_out, int tok_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* val = smem;                                  // [P]
    unsigned short* ids = (unsigned short*)(val + P);   // [P]
    double* dred = (double*)(ids + P);                  // [16] wave partials (P*6 bytes is 8-aligned: P % 4 == 0)
    float* fred = (float*)(dred + 16);                  // [16]
    int* ired = (int*)(fred + 16);                      // [16]
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = logits + (long)row * ldl;
    for (int i = tid; i < P; i += 1024) {
        val[i] = i < V ? lg[i] : -INFINITY;
        ids[i] = (unsigned short)i;
    }
    __syncthreads();
    // bitonic sort, descending (ties: smaller id first)
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += 1024) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const int l = i + j;
                const bool desc = (i & k) == 0;
                const float a = val[i], b = val[l];
                const unsigned short ia = ids[i], ib = ids[l];
                const bool a_first = (a > b) || (a == b && ia < ib);   // a should precede b in descending order
                if (desc ? !a_first : a_first) {
                    val[i] = b; val[l] = a;
                    ids[i] = ib; ids[l] = ia;
                }
            }
            __syncthreads();
        }
    }
    const float mx = val[0];
    // softmax denominator
    float s = 0.f;
    for (int i = tid; i < V; i += 1024) s += expf(val[i] - mx);
    s = wave_sum(s);
    if (lane == 0) fred[wave] = s;
    __syncthreads();
    float denom = 0.f;
    for (int w = 0; w < 16; ++w) denom += fred[w];
    __syncthreads();
    // inclusive cumulative sum in sorted order (double accumulation: torch's CPU cumsum uses a
    // double accumulator for float inputs), each thread owns PER consecutive entries
    const int PER = P / 1024;
    double local = 0.0;
    for (int e = 0; e < PER; ++e) {
        const int i = tid * PER + e;
        if (i < V) local += (double)(expf(val[i] - mx) / denom);
    }
    double incl = local;                                 // wave inclusive scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) dred[wave] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += dred[w];
    double run = base + incl - local;                    // exclusive prefix of this thread
    int first_rm = P;                                    // first sorted rank (>= 1) with cum > top_p
    for (int e = 0; e < PER; ++e) {
        const int i = tid * PER + e;
        if (i < V) {
            run += (double)(expf(val[i] - mx) / denom);
            if (i >= 1 && (float)run > top_p && i < first_rm) first_rm = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first_rm = min(first_rm, __shfl_xor(first_rm, o, 64));
    if (lane == 0) ired[wave] = first_rm;
    __syncthreads();
    int ncut = P;
    for (int w = 0; w < 16; ++w) ncut = min(ncut, ired[w]);
    if (ncut > V) ncut = V;
    __syncthreads();
    // temperature softmax over the kept prefix, then argmax(p / q)
    const float m2 = mx * inv_temp;
    float s2 = 0.f;
    for (int i = tid; i < ncut; i += 1024) s2 += expf(val[i] * inv_temp - m2);
    s2 = wave_sum(s2);
    if (lane == 0) fred[wave] = s2;
    __syncthreads();
    float denom2 = 0.f;
    for (int w = 0; w < 16; ++w) denom2 += fred[w];
    __syncthreads();
    float best = -1.f;
    int best_id = 0x7fffffff;
    for (int i = tid; i < ncut; i += 1024) {
        const int id = ids[i];
        const float p = expf(val[i] * inv_temp - m2) / denom2;
        const float q = noise ? noise[(long)row * ldn + id]
                              : exp1_noise_dev(seed[row], frame[row], kind, (unsigned)(noise_elem_off + id));
        const float r = p / q;
        if (r > best || (r == best && id < best_id)) { best = r; best_id = id; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_id, o, 64);
        if (ob > best || (ob == best && oi < best_id)) { best = ob; best_id = oi; }
    }
    if (lane == 0) { fred[wave] = best; ired[wave] = best_id; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < best_id)) { best = fred[w]; best_id = ired[w]; }
        tok_out[(long)row * tok_stride] = best_id;
    }
}
// Register-resident variant for P = 1024 * PER entries (the 8192-way semantic head): thread t owns PER consecutive
// positions, so of the log2(P)(log2(P)+1)/2 compare-exchange stages only those with partner distance >= 64 * PER go
// through LDS (10 of 91 for P = 8192); distances < PER stay inside a thread, the rest are wave shuffles.  Same order
// (descending, ties by smaller id), cumulative sum, cut and argmax rules as sampler_kernel.
template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void sampler_reg_kernel(const float* __restrict__ logits, int V, int ldl,
                                                           const float* __restrict__ noise, int ldn,
                                                           const unsigned long long* __restrict__ seed, const int* __restrict__ frame,
                                                           int kind, int noise_elem_off, float inv_temp, float top_p,
                                                           int* __restrict__ tok_out, int tok_stride,
                                                           // optional fusion (fast-AR heads): teacher forcing + gather of the next input embedding
                                                           int* __restrict__ tok, const int* __restrict__ forced, int forced_stride,
                                                           const int* __restrict__ use_forced, const float* __restrict__ emb_table, int D,
                                                           float* __restrict__ emb_out, int ldo) {
    constexpr int P = THREADS * PER, NW = THREADS / 64;
    __shared__ float xv[P];
    __shared__ unsigned short xi[P];
    __shared__ double dred[16];
    __shared__ float fred[16];
    __shared__ int ired[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = logits + (long)row * ldl;
    float v[PER];
    int id[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int e = tid * PER + r;
        v[r] = e < V ? lg[e] : -INFINITY;
        id[r] = e;
    }
    // my element precedes the other in descending order?
    auto first = [](float a, int ia, float b, int ib) { return (a > b) || (a == b && ia < ib); };
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * PER) {                                   // partner in another wave
                const int pt = tid ^ (j / PER);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < PER; ++r) { xv[r * THREADS + tid] = v[r]; xi[r * THREADS + tid] = (unsigned short)id[r]; }
                __syncthreads();
                const bool lower = (tid & (j / PER)) == 0;
                const bool desc = ((tid * PER) & k) == 0;
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    const float ov = xv[r * THREADS + pt];
                    const int oi = xi[r * THREADS + pt];
                    if (first(v[r], id[r], ov, oi) != (desc == lower)) { v[r] = ov; id[r] = oi; }
                }
            } else if (j >= PER) {                                 // partner lane in this wave
                const int dl = j / PER;
                const bool lower = (lane & dl) == 0;
                const bool desc = ((tid * PER) & k) == 0;
#define SVA_XSTAGE(DL_)                                                                              \
    _Pragma("unroll") for (int r = 0; r < PER; ++r) {                                                \
        const float ov = lane_xor_f<DL_>(v[r]);                                                      \
        const int oi = lane_xor_i<DL_>(id[r]);                                                       \
        if (first(v[r], id[r], ov, oi) != (desc == lower)) { v[r] = ov; id[r] = oi; }                \
    }
                switch (dl) {
                    case 1: SVA_XSTAGE(1) break;
                    case 2: SVA_XSTAGE(2) break;
                    case 4: SVA_XSTAGE(4) break;
                    case 8: SVA_XSTAGE(8) break;
                    case 16: SVA_XSTAGE(16) break;
                    default: SVA_XSTAGE(32) break;
                }
#undef SVA_XSTAGE
            } else {                                               // partner inside the thread: static register indices
#pragma unroll
                for (int jj = PER / 2; jj > 0; jj >>= 1) {
                    if (jj != j) continue;
#pragma unroll
                    for (int r = 0; r < PER; ++r) {
                        if (r & jj) continue;
                        const int l = r | jj;
                        const bool desc = (((tid * PER) + r) & k) == 0;
                        const bool a_first = first(v[r], id[r], v[l], id[l]);
                        if (desc ? !a_first : a_first) {
                            const float tv = v[r]; v[r] = v[l]; v[l] = tv;
                            const int ti = id[r]; id[r] = id[l]; id[l] = ti;
                        }
                    }
                }
            }
        }
    }
    // thread t holds ranks t*PER .. t*PER+PER-1
    __syncthreads();
    if (tid == 0) fred[0] = v[0];
    __syncthreads();
    const float mx = fred[0];
    __syncthreads();
    float ev[PER];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        ev[r] = (tid * PER + r) < V ? expf(v[r] - mx) : 0.f;
        s += ev[r];
    }
    s = wave_sum(s);
    if (lane == 0) fred[wave] = s;
    __syncthreads();
    float denom = 0.f;
    for (int w = 0; w < NW; ++w) denom += fred[w];
    __syncthreads();
    double local = 0.0;
#pragma unroll
    for (int r = 0; r < PER; ++r) local += (double)(ev[r] / denom);
    double incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) dred[wave] = incl;
    __syncthreads();
    double run = incl - local;
    for (int w = 0; w < wave; ++w) run += dred[w];
    int first_rm = P;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int i = tid * PER + r;
        if (i < V) {
            run += (double)(ev[r] / denom);
            if (i >= 1 && (float)run > top_p && i < first_rm) first_rm = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first_rm = min(first_rm, __shfl_xor(first_rm, o, 64));
    if (lane == 0) ired[wave] = first_rm;
    __syncthreads();
    int ncut = P;
    for (int w = 0; w < NW; ++w) ncut = min(ncut, ired[w]);
    if (ncut > V) ncut = V;
    __syncthreads();
    const float m2 = mx * inv_temp;
    float e2[PER];
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        e2[r] = (tid * PER + r) < ncut ? expf(v[r] * inv_temp - m2) : 0.f;
        s2 += e2[r];
    }
    s2 = wave_sum(s2);
    if (lane == 0) fred[wave] = s2;
    __syncthreads();
    float denom2 = 0.f;
    for (int w = 0; w < NW; ++w) denom2 += fred[w];
    __syncthreads();
    float best = -1.f;
    int best_id = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        if ((tid * PER + r) >= ncut) continue;
        const float pr = e2[r] / denom2;
        const float q = noise ? noise[(long)row * ldn + id[r]]
                              : exp1_noise_dev(seed[row], frame[row], kind, (unsigned)(noise_elem_off + id[r]));
        const float rr = pr / q;
        if (rr > best || (rr == best && id[r] < best_id)) { best = rr; best_id = id[r]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_id, o, 64);
        if (ob > best || (ob == best && oi < best_id)) { best = ob; best_id = oi; }
    }
    if (lane == 0) { fred[wave] = best; ired[wave] = best_id; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < best_id)) { best = fred[w]; best_id = ired[w]; }
        tok_out[(long)row * tok_stride] = best_id;
        if (tok) {
            int t = best_id;
            if (forced && *use_forced) t = forced[(long)row * forced_stride];
            tok[(long)row * tok_stride] = t;
            ired[0] = t;
        }
    }
    if (emb_table) {
        __syncthreads();
        const int t = ired[0];
        for (int c = tid; c < D; c += THREADS) emb_out[(long)row * ldo + c] = emb_table[(long)t * D + c];
    }
}

// ---------------------------------------------------------------------------------------
// Sampler edits (previous_tokens / repetition_penalty / suppress_tokens of logits_to_probs, modules/dual_ar_stream.py:1099-1117)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logit_edit_kernel(float* __restrict__ logits, int V, int ldl, const int* __restrict__ prev, int prev_cap,
                                                         const int* __restrict__ suppress, const int* __restrict__ params) {
    constexpr int PER = 16;                                  // 256 x 16 = 4096 listed tokens at most
    float* row = logits + (long)blockIdx.x * ldl;
    const int W = min(params[0], prev_cap), ns = params[1];
    const float penalty = __int_as_float(params[2]);
    // phase 1: every listed token's ORIGINAL score (torch.gather before scatter_: duplicates all read the unedited value)
    float sc[PER];
    int id[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * 256;
        id[k] = i < W ? prev[i] : -1;
        if (id[k] >= V) id[k] = -1;
        sc[k] = id[k] >= 0 ? row[id[k]] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (id[k] >= 0) row[id[k]] = sc[k] < 0.f ? sc[k] * penalty : sc[k] / penalty;
    __syncthreads();
    for (int i = threadIdx.x; i < ns; i += 256) {
        const int t = suppress[i];
        if (t >= 0 && t < V) row[t] = -INFINITY;
    }
}

int launch_logit_edits(float* logits, int rows, int V, int ldl, const int* prev, int prev_cap, const int* suppress, const int* params, hipStream_t st) {
    SVA_CHECK(prev_cap <= 4096, "logit edits: at most 4096 previous tokens per head");
    hipLaunchKernelGGL(logit_edit_kernel, dim3(rows), dim3(256), 0, st, logits, V, ldl, prev, prev_cap, suppress, params);
    SVA_HIP(hipGetLastError());
    return 0;
}


// Sort-free variant.  The nucleus rule only needs to know, for each entry, whether the probability mass of the entries
// that precede it in descending order (itself included) exceeds top_p; without ties that mass is F(p_i) = sum of all
// p_j >= p_i, a non-increasing step function of the threshold.  So instead of sorting, bisect the threshold over the
// float bit pattern of p (monotone for p >= 0): kb = the largest key whose F (double accumulator, compared after the
// cast to float like the sorted scan) exceeds top_p.  Entries with key > kb are kept, entries below are cut, the
// entries AT kb are resolved in id order (the sort's tie rule) from F(kb + 1) by repeated addition, and the top entry
// is always kept.  ~30 block reductions of one double instead of 91 compare-exchange stages; SEARCH > 0 places the probes by
// interpolation and snaps the bracket onto present keys (~10 probes, same result).
template <int THREADS, int PER, int SEARCH = 0>
__global__ __launch_bounds__(THREADS) void sampler_bisect_kernel(const float* __restrict__ logits, int V, int ldl,
                                                              const float* __restrict__ noise, int ldn,
                                                              const unsigned long long* __restrict__ seed, const int* __restrict__ frame,
                                                              int kind, int noise_elem_off, float inv_temp, float top_p,
                                                              int* __restrict__ tok_out, int tok_stride,
                                                              int* __restrict__ tok, const int* __restrict__ forced, int forced_stride,
                                                              const int* __restrict__ use_forced, const float* __restrict__ emb_table, int D,
                                                              float* __restrict__ emb_out, int ldo) {
    constexpr int NW = THREADS / 64;
    __shared__ double dred[2][NW];
    __shared__ unsigned ured[2][2][NW];
    __shared__ float fred[2][NW];
    __shared__ int ired[NW + 1];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = logits + (long)row * ldl;
    int slot = 0;
    auto block_sum_d = [&](double x) {
        x = wave_sum_d(x);
        if (lane == 0) dred[slot][wave] = x;
        __syncthreads();
        // second level: lane i takes the partial of wave i % NW, the first log2(NW) DPP steps sum each aligned group of NW lanes
        double t = dred[slot][lane & (NW - 1)];
        if constexpr (NW >= 2) t += dpp_mov_d<0xB1>(t);
        if constexpr (NW >= 4) t += dpp_mov_d<0x4E>(t);
        if constexpr (NW >= 8) t += dpp_mov_d<0x141>(t);
        if constexpr (NW >= 16) t += dpp_mov_d<0x140>(t);
        slot ^= 1;
        return t;
    };
    auto block_sum_f = [&](float x) {
        x = wave_sum(x);
        if (lane == 0) fred[slot][wave] = x;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += fred[slot][w];
        slot ^= 1;
        return t;
    };
    float l[PER];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int e = tid + r * THREADS;
        l[r] = e < V ? lg[e] : -INFINITY;
        m = fmaxf(m, l[r]);
    }
    m = wave_max(m);
    if (lane == 0) fred[slot][wave] = m;
    __syncthreads();
    float mx = fred[slot][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, fred[slot][w]);
    slot ^= 1;
    float p[PER];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        p[r] = (tid + r * THREADS) < V ? expf(l[r] - mx) : 0.f;
        s += p[r];
    }
    const float denom = block_sum_f(s);
    unsigned key[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        p[r] = p[r] / denom;
        key[r] = __builtin_bit_cast(unsigned, p[r]);
    }
    auto mass_from = [&](unsigned k) {          // F(k) = sum of p with key >= k
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < PER; ++r) t += key[r] >= k ? (double)p[r] : 0.0;
        return block_sum_d(t);
    };
    bool keep[PER];
    if (!((float)mass_from(0u) > top_p)) {      // nothing exceeds top_p: the whole distribution is kept
#pragma unroll
        for (int r = 0; r < PER; ++r) keep[r] = (tid + r * THREADS) < V;
    } else {
        unsigned lo = 0u, hi = __builtin_bit_cast(unsigned, 1.0f / denom) + 2u;      // every key < hi (p <= 1 / denom up to rounding)
        if (hi > 0x3F800001u) hi = 0x3F800001u;
        if constexpr (SEARCH == 0) {
            while (hi - lo > 1u) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                if ((float)mass_from(mid) > top_p) lo = mid; else hi = mid;
            }
        } else {
            // Same bracket invariant (mass(lo) exceeds top_p, mass(hi) does not), so the same kb whatever probes are used -- but
            // every probe also returns the nearest PRESENT keys on both sides, which snap the bracket onto them (the mass is
            // constant between present keys), and every other probe is placed by linear interpolation of the mass instead
            // of at the midpoint: ~10 probes instead of ~30.
            auto umax_wave = [&](unsigned v) {
                v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
                v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
                v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
                v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
                v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
                v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
                return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
            };
            double f_lo = 1.0, f_hi = 0.0;
            int it = 0;
            while (hi - lo > 1u) {
                unsigned mid = lo + ((hi - lo) >> 1);
                if (SEARCH == 1 ? !(it & 1) : (it % 3) != 2) {
                    const double t = (f_lo - (double)top_p) / (f_lo - f_hi);
                    const double off = (double)(hi - lo) * (t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t));
                    unsigned m2 = lo + (unsigned)off;
                    if (m2 <= lo) m2 = lo + 1u;
                    if (m2 >= hi) m2 = hi - 1u;
                    mid = m2;
                }
                ++it;
                double t = 0.0;
                unsigned below = 0u, above_inv = 0u;           // max key < mid; ~(min key >= mid)
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    const bool ge = key[r] >= mid;
                    t += ge ? (double)p[r] : 0.0;
                    above_inv = max(above_inv, ge ? ~key[r] : 0u);
                    below = max(below, ge ? 0u : key[r]);
                }
                t = wave_sum_d(t);
                below = umax_wave(below);
                above_inv = umax_wave(above_inv);
                if (lane == 0) { dred[slot][wave] = t; ured[slot][0][wave] = below; ured[slot][1][wave] = above_inv; }
                __syncthreads();
                double fm = 0.0;
                unsigned kl = 0u, kgi = 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w) { fm += dred[slot][w]; kl = max(kl, ured[slot][0][w]); kgi = max(kgi, ured[slot][1][w]); }
                slot ^= 1;
                if ((float)fm > top_p) { lo = ~kgi; f_lo = fm; }        // smallest present key >= mid: same mass
                else { hi = kl + 1u; f_hi = fm; }                          // just above the largest present key < mid: same mass
            }
        }
        const unsigned kb = lo;
        const float pb = __builtin_bit_cast(float, kb);
        double above = 0.0;
        float ties = 0.f;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            above += key[r] > kb ? (double)p[r] : 0.0;
            ties += (key[r] == kb && (tid + r * THREADS) < V) ? 1.f : 0.f;
        }
        const double base = block_sum_d(above);
        const int cnt = (int)block_sum_f(ties);
        int nk = 0;                                       // ties kept, in id order: cumulative mass by repeated addition like the scan
        double run = base;
        for (int j = 0; j < cnt; ++j) {
            run += (double)pb;
            if ((float)run > top_p) break;
            ++nk;
        }
        if (base == 0.0 && nk == 0) nk = 1;              // the top entry is never cut
        int id_cut = -1;
        if (nk >= cnt) id_cut = 0x7fffffff;
        else if (nk > 0) {                                // the nk-th smallest id among the ties
            int ilo = -1, ihi = V - 1;
            while (ihi - ilo > 1) {
                const int mid = ilo + ((ihi - ilo) >> 1);
                float c = 0.f;
#pragma unroll
                for (int r = 0; r < PER; ++r) c += (key[r] == kb && (tid + r * THREADS) <= mid) ? 1.f : 0.f;
                if ((int)block_sum_f(c) >= nk) ihi = mid; else ilo = mid;
            }
            id_cut = ihi;
        }
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int e = tid + r * THREADS;
            keep[r] = e < V && (key[r] > kb || (key[r] == kb && e <= id_cut));
        }
    }
    const float m2 = mx * inv_temp;
    float e2[PER];
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        e2[r] = keep[r] ? expf(l[r] * inv_temp - m2) : 0.f;
        s2 += e2[r];
    }
    const float denom2 = block_sum_f(s2);
    float best = -1.f;
    int best_id = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        if (!keep[r]) continue;
        const int e = tid + r * THREADS;
        const float pr = e2[r] / denom2;
        const float q = noise ? noise[(long)row * ldn + e] : exp1_noise_dev(seed[row], frame[row], kind, (unsigned)(noise_elem_off + e));
        const float rr = pr / q;
        if (rr > best || (rr == best && e < best_id)) { best = rr; best_id = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_id, o, 64);
        if (ob > best || (ob == best && oi < best_id)) { best = ob; best_id = oi; }
    }
    if (lane == 0) { fred[slot][wave] = best; ired[wave] = best_id; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w)
            if (fred[slot][w] > best || (fred[slot][w] == best && ired[w] < best_id)) { best = fred[slot][w]; best_id = ired[w]; }
        tok_out[(long)row * tok_stride] = best_id;
        int t = best_id;
        if (tok) {
            if (forced && *use_forced) t = forced[(long)row * forced_stride];
            tok[(long)row * tok_stride] = t;
        }
        ired[NW] = t;
    }
    if (emb_table) {
        __syncthreads();
        const int t = ired[NW];
        for (int c = tid; c < D; c += THREADS) emb_out[(long)row * ldo + c] = emb_table[(long)t * D + c];
    }
}
// The production sampler is the sort-free threshold search with interpolated, key-snapped probes; the sorting kernels and the other
// search shapes below stay reachable through the unit-test hook only (sva_test_sampler variants 1..7, tests compare them all).
static constexpr int sampler_mode() { return 0; }

int launch_sampler(const float* logits, int rows, int V, int ldl, const float* noise, int ldn,
                   const unsigned long long* seed, const int* frame, int kind, int noise_elem_off, float temperature,
                   float top_p, int* tok_out, int tok_stride, hipStream_t st) {
    int P = 1024;
    while (P < V) P <<= 1;
    SVA_CHECK(P <= 16384, "sampler: vocabulary too large");
    const size_t smem = (size_t)P * 6 + 16 * 8 + 16 * 4 + 16 * 4;
    static DeviceOnce attr_set;
    if (attr_set.needed()) {
        SVA_HIP(hipFuncSetAttribute((const void*)sampler_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set.done();
    }
    const float tclamp = temperature > 1e-5f ? temperature : 1e-5f;
    constexpr bool legacy = false;
    if (P == 8192 && !legacy && !sampler_mode()) {
        constexpr int bv = 1, srch = 2;      // 512 x 16 (24.8 us) beats 1024 x 8 (28.2 us); interpolated probes
        if (bv == 1 && srch)
            hipLaunchKernelGGL((sampler_bisect_kernel<512, 16, 2>), dim3(rows), dim3(512), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                               noise_elem_off, 1.0f / tclamp, top_p, tok_out, tok_stride, (int*)nullptr, (const int*)nullptr, 0,
                               (const int*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0);
        else if (bv == 1)
            hipLaunchKernelGGL((sampler_bisect_kernel<512, 16>), dim3(rows), dim3(512), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                               noise_elem_off, 1.0f / tclamp, top_p, tok_out, tok_stride, (int*)nullptr, (const int*)nullptr, 0,
                               (const int*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0);
        else
            hipLaunchKernelGGL((sampler_bisect_kernel<1024, 8>), dim3(rows), dim3(1024), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                               noise_elem_off, 1.0f / tclamp, top_p, tok_out, tok_stride, (int*)nullptr, (const int*)nullptr, 0,
                               (const int*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (P == 8192 && !legacy) {
        hipLaunchKernelGGL((sampler_reg_kernel<1024, 8>), dim3(rows), dim3(1024), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                           noise_elem_off, 1.0f / tclamp, top_p, tok_out, tok_stride, (int*)nullptr, (const int*)nullptr, 0,
                           (const int*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(sampler_kernel, dim3(rows), dim3(1024), smem, st, logits, V, ldl, P, noise, ldn, seed, frame, kind,
                       noise_elem_off, 1.0f / tclamp, top_p, tok_out, tok_stride);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// embeddings
// ------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ table, const int* __restrict__ idx, int idx_stride,
                                   int idx_offset, int D, float* __restrict__ out, int ldo) {
    const int r = blockIdx.x;
    const long src = (long)(idx[(long)r * idx_stride] + idx_offset) * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) out[(long)r * ldo + c] = table[src + c];
}
int launch_gather_rows(const float* table, const int* idx, int idx_stride, int idx_offset, int rows, int D, float* out,
                       int ldo, hipStream_t st) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, st, table, idx, idx_stride, idx_offset, D, out, ldo);
    SVA_HIP(hipGetLastError());
    return 0;
}
// BaseTransformer.embed (modules/dual_ar_stream.py:245-255): sum over codebooks of
// codebook_embeddings[code_i + i * codebook_size]
__global__ void audio_embed_kernel(const float* __restrict__ table, const int* __restrict__ codes, int code_stride,
                                   int cb_stride, int ncb, int codebook_size, int D, float* __restrict__ out, int ldo) {
    // one output element per thread (grid.y covers D): the ncb gathers of a thread are independent loads
    const int r = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float acc = 0.f;
    for (int i = 0; i < ncb; ++i) {
        const int code = codes[(long)r * code_stride + (long)i * cb_stride];
        acc += table[((long)code + (long)i * codebook_size) * D + c];
    }
    out[(long)r * ldo + c] = acc;
}
int launch_audio_embed(const float* table, const int* codes, int code_stride, int cb_stride, int rows, int ncb,
                       int codebook_size, int D, float* out, int ldo, hipStream_t st) {
    hipLaunchKernelGGL(audio_embed_kernel, dim3(rows, (D + 255) / 256), dim3(256), 0, st, table, codes, code_stride, cb_stride, ncb,
                       codebook_size, D, out, ldo);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// FSQ quantize of firefly.encode (modules/vqgan/modules/fsq.py:106-110 -> GroupedResidualFSQ, one quantizer per
// group; arithmetic twin at modules/bicodec_speaker_encoder/fsq/finite_scalar_quantization.py:127-154):
// z_g = project_in_g(x[g*64:(g+1)*64]); bounded = tanh(z + shift) * half_l - offset; digit = round(bounded) + L//2;
// index = sum digit_d * [1, 8, 40, 200].  One wave per (b, t, g), lane = channel of the group.
// ------------------------------------------------------------------------------------------
struct FsqConst { float half_l[4], offset[4], shift[4]; };
__global__ void fsq_encode_kernel(const float* __restrict__ x, long x_bstride, long x_off, int ldx, int T, int G, int gdim,
                                  const float* __restrict__ Win, const float* __restrict__ bin, FsqConst K,
                                  int* __restrict__ codes, long c_bstride, long c_gstride) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63;
    const int g = blockIdx.z * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= G) return;
    const float* row = x + (long)b * x_bstride + x_off + (long)t * ldx + g * gdim;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < gdim; j += 64) {
        const float v = row[j];
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] = fmaf(v, Win[((long)g * 4 + d) * gdim + j], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[d] += __shfl_xor(acc[d], o, 64);
    if (lane == 0) {
        const int halfw[4] = {4, 2, 2, 2}, basis[4] = {1, 8, 40, 200};
        int idx = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float z = acc[d] + bin[g * 4 + d];
            const float bd = tanhf(z + K.shift[d]) * K.half_l[d] - K.offset[d];
            idx += ((int)rintf(bd) + halfw[d]) * basis[d];      // torch.round = half to even
        }
        codes[(long)b * c_bstride + (long)g * c_gstride + t] = idx;
    }
}
int launch_fsq_encode(const float* x, long x_bstride, long x_off, int ldx, int B, int T, int G, int gdim, const float* Win,
                      const float* bin, int* codes, long c_bstride, long c_gstride, hipStream_t st) {
    FsqConst K;
    const int levels[4] = {8, 5, 5, 5};
    for (int d = 0; d < 4; ++d) {     // FSQ.bound, eps = 1e-3, evaluated in fp32 like the reference's buffers
        K.half_l[d] = (float)(levels[d] - 1) * (float)(1.0 + 1e-3) / 2.f;
        K.offset[d] = levels[d] % 2 == 0 ? 0.5f : 0.f;
        K.shift[d] = atanhf(K.offset[d] / K.half_l[d]);
    }
    hipLaunchKernelGGL(fsq_encode_kernel, dim3(T, B, (G + 3) / 4), dim3(256), 0, st, x, x_bstride, x_off, ldx, T, G, gdim, Win,
                       bin, K, codes, c_bstride, c_gstride);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// V1 FSQ decode (modules/vqgan/modules/fsq.py:112-114; arithmetic of vector_quantize_pytorch
// 1.14.24, twin at modules/bicodec_speaker_encoder/fsq/finite_scalar_quantization.py:143-162):
// digits = (idx // [1,8,40,200]) % [8,5,5,5]; code = (digit - half) / half, half = [4,2,2,2];
// per-group Linear 4 -> gdim.
// ------------------------------------------------------------------------------------------
__global__ void fsq_decode_kernel(const int* __restrict__ codes, long c_bstride, long c_gstride, int T, int G, int gdim,
                                  const float* __restrict__ Wout, const float* __restrict__ bout, float* __restrict__ out,
                                  long o_bstride, long o_off, int ldo) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int e = threadIdx.x; e < G * gdim; e += blockDim.x) {
        const int g = e / gdim, j = e - g * gdim;
        const int idx = codes[(long)b * c_bstride + (long)g * c_gstride + t];
        const float c0 = (float)((idx % 8) - 4) / 4.f;
        const float c1 = (float)(((idx / 8) % 5) - 2) / 2.f;
        const float c2 = (float)(((idx / 40) % 5) - 2) / 2.f;
        const float c3 = (float)(((idx / 200) % 5) - 2) / 2.f;
        const float* w = Wout + ((long)g * gdim + j) * 4;
        out[(long)b * o_bstride + o_off + (long)t * ldo + e] = bout[g * gdim + j] + w[0] * c0 + w[1] * c1 + w[2] * c2 + w[3] * c3;
    }
}
int launch_fsq_decode(const int* codes, long c_bstride, long c_gstride, int B, int T, int G, int gdim, const float* Wout,
                      const float* bout, float* out, long o_bstride, long o_off, int ldo, hipStream_t st) {
    hipLaunchKernelGGL(fsq_decode_kernel, dim3(T, B), dim3(256), 0, st, codes, c_bstride, c_gstride, T, G, gdim, Wout, bout,
                       out, o_bstride, o_off, ldo);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// V3 tail: SiLU -> conv_post (C -> 1, k taps, causal) -> tanh (firefly.py:289-291).
// 256 outputs per workgroup, silu'd input rows staged once in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const float* __restrict__ x, long x_bstride, long x_off, int T,
                                                             int C, int k, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ pcm,
                                                             long p_bstride, long p_off) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const int nrows = min(256, T - t0) + k - 1;
    const int LD = C + 1;                // odd row stride: thread t reads row t + j, so a stride of C = 16 would put 64 lanes on 2 banks
    const float* xb = x + (long)b * x_bstride + x_off + (long)t0 * C;
    for (int i = tid; i < nrows * C; i += 256) smem[(i / C) * LD + (i % C)] = silu_acc(xb[i]);
    float* ws = smem + (256 + k - 1) * LD;
    for (int i = tid; i < k * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int t = t0 + tid;
    if (t < T) {
        float acc = bias[0];
        for (int j = 0; j < k; ++j) {    // same accumulation order as before: taps outer, channels inner
            const float* r = smem + (tid + j) * LD;
            for (int c = 0; c < C; ++c) acc = fmaf(ws[j * C + c], r[c], acc);
        }
        pcm[(long)b * p_bstride + p_off + t] = tanhf(acc);
    }
}
int launch_conv_post_tanh(const float* x, long x_bstride, long x_off, int B, int T, int C, int k, const float* w,
                          const float* bias, float* pcm, long p_bstride, long p_off, hipStream_t st) {
    const size_t smem = ((size_t)(256 + k - 1) * (C + 1) + (size_t)k * C) * sizeof(float);
    SVA_CHECK(smem <= 64 * 1024, "conv_post: tile too large");
    hipLaunchKernelGGL(conv_post_tanh_kernel, dim3((T + 255) / 256, B), dim3(256), smem, st, x, x_bstride, x_off, T, C, k, w,
                       bias, pcm, p_bstride, p_off);
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// streaming state: every conv input tensor keeps H history rows in front of its T new rows;
// after a step rows [T, T+H) become the next step's history [0, H).  One workgroup per
// (tensor, stream) walks the rows in increasing address order (dst < src always), staging
// each 256-element chunk in registers across a barrier so overlapping ranges are safe.
// ------------------------------------------------------------------------------------------
// gridDim.z column slices per tensor: element (r, c) moves to (r - T, c), so slices of the channel axis are independent and a long
// tensor (the encoder's token cache: 88 rows x 512) is not one workgroup's serial loop
__global__ __launch_bounds__(256) void shift_history_kernel(const ShiftDesc* __restrict__ descs, int* counter, int counter_add) {
    // (the step counter of the chain that ends with this launch: every reader of it precedes this kernel in stream order)
    if (counter && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *counter += counter_add;
    const ShiftDesc d = descs[blockIdx.x];
    float* base = d.ptr + (long)blockIdx.y * d.bstride;
    const int zs = gridDim.z;
    const int cs = (d.C % (4 * zs) == 0) ? d.C / zs : d.C;           // slice width (whole tensor in slice 0 if C does not split)
    if (cs == d.C && blockIdx.z > 0) return;
    const int c0 = cs == d.C ? 0 : (int)blockIdx.z * cs;
    const long n = (long)d.H * cs, delta = (long)d.T * d.C;
    for (long i0 = 0; i0 < n; i0 += 4096) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long i = i0 + threadIdx.x + e * 256;
            const long r = i / cs, c = i - r * cs;
            v[e] = i < n ? base[r * d.C + c0 + c + delta] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long i = i0 + threadIdx.x + e * 256;
            const long r = i / cs, c = i - r * cs;
            if (i < n) base[r * d.C + c0 + c] = v[e];
        }
        __syncthreads();
    }
}
int launch_shift_history(const ShiftDesc* descs_dev, int n_desc, int B, hipStream_t st, int col_slices, int* counter, int counter_add) {
    if (n_desc == 0) return counter ? launch_add_i32(counter, counter_add, st) : 0;
    hipLaunchKernelGGL(shift_history_kernel, dim3(n_desc, B, col_slices < 1 ? 1 : col_slices), dim3(256), 0, st, descs_dev, counter, counter_add);
    SVA_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------------------------------
// Decode GEMV for M <= 4 token rows (B = 1..2 streams): the AR's Linear layers at batch 1 are pure weight
// streaming.  One wave owns two adjacent output columns (a RoPE pair); it reads their W rows as whole contiguous
// 1-KiB wave loads (lane l takes float4 #l, #l+64, ...), keeps the M input rows in registers, and reduces with
// wave shuffles -- no LDS, no barrier, every load issued before the first FMA.  Fusions (each one removes a
// dependent launch from a ~350-kernel chain): RMSNorm of the input rows (the wave sees the whole row, so the
// statistics cost one extra FMA per element), SwiGLU on the interleaved w1|w3 weight, and for the QKV projection
// the adjacent-pair RoPE plus the KV-cache write (modules/dual_ar_stream.py:985-990, 967-976, 1004-1016, 141-150).
// ------------------------------------------------------------------------------------------
// ATT (mode 3, fast-AR wo projection): the input row is not loaded but computed -- the decode attention of query row m
// over its <= 8 cached keys (codebook positions), recomputed by every wave (16 lanes per head cooperate on a score),
// which removes the separate attention launch from the 8 x 4 fast-layer chain.
template <int MR, int KI, int ATT>
__global__ __launch_bounds__(256) void gemv_kernel(const Gemv g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * 4 + wave;                 // pair index
    const int n_units = g.mode == 1 ? g.N / 4 : g.N / 2;
    if (unit >= n_units) return;
    // W rows of this wave: plain / rope: rows 2u, 2u+1; SwiGLU: features f = 2u, 2u+1 -> rows r1(f), r1(f)+16
    int rows[4];
    int nrows = 2;
    if (g.mode == 1) {
        nrows = 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = 2 * unit + q;
            const int r1 = (f >> 4) * 32 + (f & 15);
            rows[2 * q] = r1;
            rows[2 * q + 1] = r1 + 16;
        }
    } else {
        rows[0] = 2 * unit; rows[1] = 2 * unit + 1; rows[2] = rows[3] = 0;
    }
    float4 w[4][KI];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r < nrows) {
            const float4* wr = reinterpret_cast<const float4*>(g.W + (long)rows[r] * g.K);
#pragma unroll
            for (int i = 0; i < KI; ++i) w[r][i] = wr[lane + 64 * i];
        }
    float4 x[MR][KI];
    float ss[MR];
    if constexpr (ATT == 2) {
        // mode 4 (slow-AR wo projection): merge the split-key attention partials of row m (ar_attention_kernel, gridDim.z = S
        // splits): out = sum_s exp(mx_s - mx) o_s / sum_s exp(mx_s - mx) l_s
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int mm = m < g.M ? m : 0;
            ss[m] = 0.f;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int q4 = lane + 64 * i, h = q4 >> 4, d0 = (q4 & 15) * 4;
                const float* pw = g.X + ((long)mm * g.H + h) * g.S * 68;
                float mxs[8], ls[8], mx = -INFINITY;
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    const int s2 = sp < g.S ? sp : g.S - 1;
                    mxs[sp] = pw[s2 * 68 + 64];
                    ls[sp] = sp < g.S ? pw[s2 * 68 + 65] : 0.f;
                    if (ls[sp] > 0.f) mx = fmaxf(mx, mxs[sp]);
                }
                float den = 0.f;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    const int s2 = sp < g.S ? sp : g.S - 1;
                    const float4 o = *reinterpret_cast<const float4*>(pw + s2 * 68 + d0);
                    const float wgt = ls[sp] > 0.f ? expf(mxs[sp] - mx) : 0.f;
                    den = fmaf(wgt, ls[sp], den);
                    a.x = fmaf(wgt, o.x, a.x); a.y = fmaf(wgt, o.y, a.y); a.z = fmaf(wgt, o.z, a.z); a.w = fmaf(wgt, o.w, a.w);
                }
                const float inv = 1.f / den;
                x[m][i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
            }
        }
    } else if constexpr (ATT == 1) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int mm = m < g.M ? m : 0;
            const int L = g.pos[mm] + 1;                              // keys 0..pos (Attention.forward mask, dual_ar_stream.py:333)
            const float* kv_m = g.kv + (long)g.slot[mm] * g.kv_slot_stride;
            ss[m] = 0.f;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int q4 = lane + 64 * i, h = q4 >> 4, d0 = (q4 & 15) * 4;
                const float4 qv = *reinterpret_cast<const float4*>(g.X + (long)mm * g.ldx + h * 64 + d0);
                const float* kc = kv_m + (long)h * g.S * 64 + d0;
                const float* vc = kc + (long)g.H * g.S * 64;
                float4 kk[8], vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int jj = j < L ? j : L - 1;
                    kk[j] = *reinterpret_cast<const float4*>(kc + jj * 64);
                    vv[j] = *reinterpret_cast<const float4*>(vc + jj * 64);
                }
                float sc[8], mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = qv.x * kk[j].x + qv.y * kk[j].y + qv.z * kk[j].z + qv.w * kk[j].w;
                    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
                    sc[j] = j < L ? a * 0.125f : -INFINITY;
                    mx = fmaxf(mx, sc[j]);
                }
                float sum = 0.f;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float e = j < L ? expf(sc[j] - mx) : 0.f;
                    sum += e;
                    o.x = fmaf(e, vv[j].x, o.x); o.y = fmaf(e, vv[j].y, o.y); o.z = fmaf(e, vv[j].z, o.z); o.w = fmaf(e, vv[j].w, o.w);
                }
                const float inv = 1.f / sum;
                x[m][i] = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
            }
        }
    } else {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float4* xr = reinterpret_cast<const float4*>(g.X + (long)(m < g.M ? m : 0) * g.ldx);
        ss[m] = 0.f;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            float4 v = xr[lane + 64 * i];
            ss[m] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            if (g.norm_w) {
                const float4 nw = reinterpret_cast<const float4*>(g.norm_w)[lane + 64 * i];
                v.x *= nw.x; v.y *= nw.y; v.z *= nw.z; v.w *= nw.w;
            }
            x[m][i] = v;
        }
    }
    }
    float acc[MR][4];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.f;
            if (r < nrows) {
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    a = fmaf(x[m][i].x, w[r][i].x, a);
                    a = fmaf(x[m][i].y, w[r][i].y, a);
                    a = fmaf(x[m][i].z, w[r][i].z, a);
                    a = fmaf(x[m][i].w, w[r][i].w, a);
                }
            }
            acc[m][r] = a;
        }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nrows) acc[m][r] = wave_sum(acc[m][r]);
        if (g.norm_w) {
            const float inv = 1.f / sqrtf(wave_sum(ss[m]) / (float)g.K + g.eps);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][r] *= inv;
        }
    }
    if (lane != 0) return;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m >= g.M) break;
        if (g.mode == 1) {
            const float o0 = (acc[m][0] / (1.f + expf(-acc[m][0]))) * acc[m][1];
            const float o1 = (acc[m][2] / (1.f + expf(-acc[m][2]))) * acc[m][3];
            g.Y[(long)m * g.ldy + 2 * unit] = o0;
            g.Y[(long)m * g.ldy + 2 * unit + 1] = o1;
        } else if (g.mode == 2) {
            const int n = 2 * unit, D = g.H * 64;
            float a = acc[m][0], b2 = acc[m][1];
            if (n < 2 * D) {                       // q or k: rotate the adjacent pair
                const int d = n & 63;
                const int p = g.pos[m];
                const float c = g.rope[((long)p * 32 + (d >> 1)) * 2], sn = g.rope[((long)p * 32 + (d >> 1)) * 2 + 1];
                const float ra = a * c - b2 * sn, rb = b2 * c + a * sn;
                a = ra; b2 = rb;
            }
            if (n < D) {
                g.Y[(long)m * g.ldy + n] = a;
                g.Y[(long)m * g.ldy + n + 1] = b2;
            } else {
                const int kvsel = n < 2 * D ? 0 : 1;
                const int nn = n - (kvsel ? 2 * D : D);
                const int h = nn >> 6, d = nn & 63;
                float* dst = g.kv + (long)g.slot[m] * g.kv_slot_stride + ((long)kvsel * g.H + h) * (long)g.S * 64 + (long)g.pos[m] * 64 + d;
                dst[0] = a;
                dst[1] = b2;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int n = 2 * unit + r;
                float v = acc[m][r];
                if (g.bias) v += g.bias[n];
                if (g.res) v += g.res[(long)m * g.ldr + n];
                g.Y[(long)m * g.ldy + n] = v;
            }
        }
    }
}

int launch_gemv(const Gemv& g, hipStream_t st) {
    SVA_CHECK(g.M >= 1 && g.M <= 4, "gemv: M must be 1..4");
    SVA_CHECK(g.K % 256 == 0 && g.ldx % 4 == 0, "gemv: K must be a multiple of 256");
    SVA_CHECK(g.N % 4 == 0, "gemv: N must be a multiple of 4");
    const int n_units = g.mode == 1 ? g.N / 4 : g.N / 2;
    dim3 grid((n_units + 3) / 4);
    const int ki = g.K / 256;
    if (g.mode == 3) {
        SVA_CHECK(ki == 3 && g.H * 64 == g.K && g.S <= 8 && g.M <= 2 && !g.norm_w, "gemv: fused attention needs K = H*64 = 768, S <= 8, M <= 2");
        if (g.M == 1) hipLaunchKernelGGL((gemv_kernel<1, 3, 1>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemv_kernel<2, 3, 1>), grid, dim3(256), 0, st, g);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (g.mode == 4) {        // X = split-key attention partials [M][H][S = splits][68]
        SVA_CHECK(ki == 3 && g.H * 64 == g.K && g.S >= 1 && g.S <= 8 && g.M <= 2 && !g.norm_w, "gemv: partial merge needs K = H*64 = 768, <= 8 splits, M <= 2");
        if (g.M == 1) hipLaunchKernelGGL((gemv_kernel<1, 3, 2>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemv_kernel<2, 3, 2>), grid, dim3(256), 0, st, g);
        SVA_HIP(hipGetLastError());
        return 0;
    }
#define SVA_GV(MR_, KI_) hipLaunchKernelGGL((gemv_kernel<MR_, KI_, 0>), grid, dim3(256), 0, st, g)
    if (ki == 3) {
        if (g.M == 1) SVA_GV(1, 3); else if (g.M == 2) SVA_GV(2, 3); else SVA_GV(4, 3);
    } else if (ki == 9) {
        if (g.M == 1) SVA_GV(1, 9); else if (g.M == 2) SVA_GV(2, 9); else SVA_GV(4, 9);
    } else {
        set_error("gemv: unsupported K (768 or 2304)");
        return -1;
    }
#undef SVA_GV
    SVA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// A5 sampler for V <= 1024: one token per thread, bitonic sort (descending, ties by index) with the 45 intra-wave
// compare-exchange stages done by wave shuffles in registers and only the 10 cross-wave stages through LDS, then
// softmax, inclusive scan, nucleus cut, temperature softmax and the Exp(1) argmax as in sampler_kernel.  Fused with
// teacher forcing and the gather of the next fast-AR input embedding.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sampler_small_kernel(const float* __restrict__ logits, int V, int ldl,
                                                             const float* __restrict__ noise, int ldn,
                                                             const unsigned long long* __restrict__ seed, const int* __restrict__ frame,
                                                             int kind, int noise_elem_off, float inv_temp, float top_p,
                                                             int* __restrict__ tok_raw, int* __restrict__ tok, int tok_stride,
                                                             const int* __restrict__ forced, int forced_stride,
                                                             const int* __restrict__ use_forced, const float* __restrict__ emb_table,
                                                             int D, float* __restrict__ emb_out, int ldo) {
    __shared__ float xv[1024];
    __shared__ int xi[1024];
    __shared__ float fred[16];
    __shared__ double dred[16];
    __shared__ int ired[16];
    __shared__ int chosen;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v = tid < V ? logits[(long)row * ldl + tid] : -INFINITY;
    int id = tid;
    for (int k = 2; k <= 1024; k <<= 1) {
        const bool desc = (tid & k) == 0;
        for (int j = k >> 1; j > 0; j >>= 1) {
            float ov;
            int oi;
            if (j >= 64) {
                __syncthreads();
                xv[tid] = v; xi[tid] = id;
                __syncthreads();
                ov = xv[tid ^ j]; oi = xi[tid ^ j];
            } else {
                switch (j) {        // DPP inside a 16-lane row, ds_bpermute beyond
                    case 1: ov = lane_xor_f<1>(v); oi = lane_xor_i<1>(id); break;
                    case 2: ov = lane_xor_f<2>(v); oi = lane_xor_i<2>(id); break;
                    case 4: ov = lane_xor_f<4>(v); oi = lane_xor_i<4>(id); break;
                    case 8: ov = lane_xor_f<8>(v); oi = lane_xor_i<8>(id); break;
                    case 16: ov = lane_xor_f<16>(v); oi = lane_xor_i<16>(id); break;
                    default: ov = lane_xor_f<32>(v); oi = lane_xor_i<32>(id); break;
                }
            }
            const bool lower = (tid & j) == 0;
            const bool mine_first = (v > ov) || (v == ov && id < oi);       // my element precedes the partner's in descending order
            const bool keep_first = desc == lower;                          // this position keeps the element that precedes
            if (mine_first != keep_first) { v = ov; id = oi; }
        }
    }
    // thread t now holds the rank-t element
    if (tid == 0) fred[0] = v;
    __syncthreads();
    const float mx = fred[0];
    __syncthreads();
    const float e = tid < V ? expf(v - mx) : 0.f;
    float s = wave_sum(e);
    if (lane == 0) fred[wave] = s;
    __syncthreads();
    float denom = 0.f;
    for (int w = 0; w < 16; ++w) denom += fred[w];
    const double p = (double)(e / denom);
    double incl = p;                                     // inclusive scan in rank order (double, like torch's CPU cumsum)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) dred[wave] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += dred[w];
    const float cum = (float)(base + incl);
    const bool keep = tid < V && (tid == 0 || !(cum > top_p));
    const float m2 = mx * inv_temp;
    const float e2 = keep ? expf(v * inv_temp - m2) : 0.f;
    float s2 = wave_sum(e2);
    __syncthreads();
    if (lane == 0) fred[wave] = s2;
    __syncthreads();
    float denom2 = 0.f;
    for (int w = 0; w < 16; ++w) denom2 += fred[w];
    float best = -1.f;
    int best_id = 0x7fffffff;
    if (keep) {
        const float q = noise ? noise[(long)row * ldn + id] : exp1_noise_dev(seed[row], frame[row], kind, (unsigned)(noise_elem_off + id));
        best = (e2 / denom2) / q;
        best_id = id;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi2 = __shfl_xor(best_id, o, 64);
        if (ob > best || (ob == best && oi2 < best_id)) { best = ob; best_id = oi2; }
    }
    __syncthreads();
    if (lane == 0) { fred[wave] = best; ired[wave] = best_id; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < best_id)) { best = fred[w]; best_id = ired[w]; }
        tok_raw[(long)row * tok_stride] = best_id;
        int t = best_id;
        if (forced && *use_forced) t = forced[(long)row * forced_stride];
        tok[(long)row * tok_stride] = t;
        chosen = t;
    }
    __syncthreads();
    if (emb_table) {
        const int t = chosen;
        for (int c = tid; c < D; c += 1024) emb_out[(long)row * ldo + c] = emb_table[(long)t * D + c];
    }
}
int launch_sampler_small(const float* logits, int rows, int V, int ldl, const float* noise, int ldn,
                         const unsigned long long* seed, const int* frame, int kind, int noise_elem_off, float temperature,
                         float top_p, int* tok_raw, int* tok, int tok_stride, const int* forced, int forced_stride,
                         const int* use_forced, const float* emb_table, int D, float* emb_out, int ldo, hipStream_t st) {
    SVA_CHECK(V <= 1024, "sampler_small: V <= 1024");
    const float tclamp = temperature > 1e-5f ? temperature : 1e-5f;
    constexpr int variant = 0, sbv = 1;
    if (variant == 0 && !sampler_mode() && sbv) {
#define SVA_BIS(T_, P_)                                                                                                          \
    hipLaunchKernelGGL((sampler_bisect_kernel<T_, P_>), dim3(rows), dim3(T_), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind, \
                       noise_elem_off, 1.0f / tclamp, top_p, tok_raw, tok_stride, tok, forced, forced_stride, use_forced, emb_table, D, \
                       emb_out, ldo)
        constexpr int srch = 2;
        if (sbv == 2) SVA_BIS(128, 8);
        else if (sbv == 3) SVA_BIS(64, 16);
        else if (sbv == 4) SVA_BIS(512, 2);
        else if (!srch) SVA_BIS(256, 4);
        else
            hipLaunchKernelGGL((sampler_bisect_kernel<256, 4, 2>), dim3(rows), dim3(256), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                               noise_elem_off, 1.0f / tclamp, top_p, tok_raw, tok_stride, tok, forced, forced_stride, use_forced, emb_table, D,
                               emb_out, ldo);
#undef SVA_BIS
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (variant == 2) {
        hipLaunchKernelGGL((sampler_reg_kernel<512, 2>), dim3(rows), dim3(512), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                           noise_elem_off, 1.0f / tclamp, top_p, tok_raw, tok_stride, tok, forced, forced_stride, use_forced, emb_table, D,
                           emb_out, ldo);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (variant == 3) {
        hipLaunchKernelGGL((sampler_reg_kernel<1024, 1>), dim3(rows), dim3(1024), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                           noise_elem_off, 1.0f / tclamp, top_p, tok_raw, tok_stride, tok, forced, forced_stride, use_forced, emb_table, D,
                           emb_out, ldo);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    if (variant == 1) {     // 256 threads x 4 entries: 3 LDS exchange stages instead of 10, 4 waves at the barriers instead of 16
        hipLaunchKernelGGL((sampler_reg_kernel<256, 4>), dim3(rows), dim3(256), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind,
                           noise_elem_off, 1.0f / tclamp, top_p, tok_raw, tok_stride, tok, forced, forced_stride, use_forced, emb_table, D,
                           emb_out, ldo);
        SVA_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(sampler_small_kernel, dim3(rows), dim3(1024), 0, st, logits, V, ldl, noise, ldn, seed, frame, kind, noise_elem_off,
                       1.0f / tclamp, top_p, tok_raw, tok, tok_stride, forced, forced_stride, use_forced, emb_table, D, emb_out, ldo);
    SVA_HIP(hipGetLastError());
    return 0;
}

// unit-test / microbenchmark entry: one explicit implementation of the A5 sampler.
//   1 LDS bitonic sort, 2 register sort, 3 / 4 / 5 threshold bisection (1024x8 | 256x4, 512x16 | 128x8, 64x16 for V <= 1024)
int launch_sampler_variant(int variant, const float* logits, int rows, int V, int ldl, const float* noise, int ldn,
                           const unsigned long long* seed, const int* frame, float temperature, float top_p, int* tok_out, hipStream_t st) {
    const float it = 1.0f / (temperature > 1e-5f ? temperature : 1e-5f);
    int P = 1024;
    while (P < V) P <<= 1;
    SVA_CHECK(P <= 16384, "sampler: vocabulary too large");
#define SVA_ARGS logits, V, ldl, noise, ldn, seed, frame, 0, 0, it, top_p, tok_out, 1
#define SVA_NOFUSE (int*)nullptr, (const int*)nullptr, 0, (const int*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0
    if (variant == 1) {
        SVA_HIP(hipFuncSetAttribute((const void*)sampler_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipLaunchKernelGGL(sampler_kernel, dim3(rows), dim3(1024), (size_t)P * 6 + 16 * 8 + 16 * 4 + 16 * 4, st, logits, V, ldl, P, noise, ldn,
                           seed, frame, 0, 0, it, top_p, tok_out, 1);
    } else if (variant == 2) {
        SVA_CHECK(P == 8192 || P == 1024, "sampler variant 2: V <= 1024 or 4096 < V <= 8192");
        if (P == 8192) hipLaunchKernelGGL((sampler_reg_kernel<1024, 8>), dim3(rows), dim3(1024), 0, st, SVA_ARGS, SVA_NOFUSE);
        else hipLaunchKernelGGL(sampler_small_kernel, dim3(rows), dim3(1024), 0, st, logits, V, ldl, noise, ldn, seed, frame, 0, 0, it, top_p,
                                tok_out, tok_out, 1, (const int*)nullptr, 0, (const int*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0);
    } else if (variant >= 3 && variant <= 5) {
        SVA_CHECK(P == 8192 || P == 1024, "sampler variant 3-5: V <= 1024 or 4096 < V <= 8192");
        if (P == 8192) {
            if (variant == 3) hipLaunchKernelGGL((sampler_bisect_kernel<1024, 8>), dim3(rows), dim3(1024), 0, st, SVA_ARGS, SVA_NOFUSE);
            else hipLaunchKernelGGL((sampler_bisect_kernel<512, 16>), dim3(rows), dim3(512), 0, st, SVA_ARGS, SVA_NOFUSE);
        } else {
            if (variant == 3) hipLaunchKernelGGL((sampler_bisect_kernel<256, 4>), dim3(rows), dim3(256), 0, st, SVA_ARGS, SVA_NOFUSE);
            else if (variant == 4) hipLaunchKernelGGL((sampler_bisect_kernel<128, 8>), dim3(rows), dim3(128), 0, st, SVA_ARGS, SVA_NOFUSE);
            else hipLaunchKernelGGL((sampler_bisect_kernel<64, 16>), dim3(rows), dim3(64), 0, st, SVA_ARGS, SVA_NOFUSE);
        }
    } else if (variant == 6) {
        SVA_CHECK(P == 8192 || P == 1024, "sampler variant 6: V <= 1024 or 4096 < V <= 8192");
        if (P == 8192) hipLaunchKernelGGL((sampler_bisect_kernel<512, 16, 1>), dim3(rows), dim3(512), 0, st, SVA_ARGS, SVA_NOFUSE);
        else hipLaunchKernelGGL((sampler_bisect_kernel<256, 4, 1>), dim3(rows), dim3(256), 0, st, SVA_ARGS, SVA_NOFUSE);
    } else if (variant == 7) {
        SVA_CHECK(P == 8192 || P == 1024, "sampler variant 7: V <= 1024 or 4096 < V <= 8192");
        if (P == 8192) hipLaunchKernelGGL((sampler_bisect_kernel<512, 16, 2>), dim3(rows), dim3(512), 0, st, SVA_ARGS, SVA_NOFUSE);
        else hipLaunchKernelGGL((sampler_bisect_kernel<256, 4, 2>), dim3(rows), dim3(256), 0, st, SVA_ARGS, SVA_NOFUSE);
    } else {
        SVA_CHECK(false, "sampler variant: 1..7");
    }
#undef SVA_ARGS
#undef SVA_NOFUSE
    SVA_HIP(hipGetLastError());
    return 0;
}

}  // namespace sva
