// Fused HiFiGAN ParallelBlock level for the narrow levels (C = 16 / 32 channels): the three ResBlock1 branches
// (kernel sizes 3 / 7 / 11, each  x += c2(silu(c1(silu(x))))  for dilations 1, 3, 5 -- modules/vqgan/modules/firefly.py:149-190,
// 214-215) run as ONE launch; a workgroup owns (time tile, branch, stream) and keeps the running activation and the
// intermediate of the current pair in LDS for all six convs.
//
// State: the only streaming state of the level is the HISTORY OF ITS INPUT x (the 18 (k-1) = 180 rows in front of the new
// rows -- the causal receptive field of the six-conv chain); every intermediate row a tile needs in front of its first output
// row is recomputed from it (trapezoid: each conv produces (k-1) dil fewer leading rows than it consumed).  The reference pads
// EVERY conv's input with zeros on the left of the stream start, so rows in front of the stream start are forced to zero in
// every intermediate (frames_done tells where the stream starts).
//
// Arithmetic: v_mfma_f32_16x16x4_f32, M = 16 time rows, N = 16 output channels, K walks (tap, input channel); the B operand
// (weights, [Cout][k * Cin] as the GEMM path keeps them) stays in registers for a whole conv, the A operand is one ds_read_b128
// per four MFMAs.  SiLU of a conv-1 input is applied on the operand read (k-fold redundant, hidden behind the matrix pipe for
// C = 32); the conv-1 output is stored already activated.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sva_common.h"

namespace sva {
namespace {

typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_v(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// B fragments of one conv: entry (t, cg, nt) of lane (n16, kk) = W[nt * 16 + n16][t * C + cg * 16 + kk * 4 .. + 3]
template <int C, int K>
__device__ __forceinline__ void load_weights(const float* __restrict__ W, int lane, vf4 (&w)[K][C / 16][C / 16]) {
    const int n16 = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int cg = 0; cg < C / 16; ++cg)
#pragma unroll
            for (int nt = 0; nt < C / 16; ++nt)
                w[t][cg][nt] = *reinterpret_cast<const vf4*>(W + (long)(nt * 16 + n16) * (K * C) + t * C + cg * 16 + kk * 4);
}

// one conv of the chain over local output rows [lo, lo + n_out): IN/OUT are LDS row buffers with leading dimension C + 4.
//   FIRST: conv 1 of a pair -- input read through SiLU, output stored as silu(acc) into OUT (= T)
//   else : conv 2 of a pair -- input already activated, OUT (= Y) += acc
// valid0: first local row that lies inside the stream (rows in front of it stay / become zero)
template <int C, int K, bool FIRST, bool ACT_ON_READ>
__device__ __forceinline__ void conv_rows(const float* __restrict__ IN, float* __restrict__ OUT, float* __restrict__ OUT_ACT, const float* __restrict__ W,
                                          const float* __restrict__ bias, int dil, int lo, int n_out, int valid0) {
    constexpr int LD = C + 4, NT = C / 16, CG = C / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kk = lane >> 4;
    vf4 w[K][CG][NT];
    load_weights<C, K>(W, lane, w);
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = bias[nt * 16 + m];
    const int ntile = (n_out + 15) >> 4;
    for (int rt = wave; rt < ntile; rt += 4) {
        const int r0 = lo + rt * 16;
        vf4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = vf4{bv[nt], bv[nt], bv[nt], bv[nt]};
        const float* in_row = IN + (long)(r0 + m - (K - 1) * dil) * LD + kk * 4;
#pragma unroll
        for (int t = 0; t < K; ++t) {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                vf4 a = *reinterpret_cast<const vf4*>(in_row + (long)t * dil * LD + cg * 16);
                if (ACT_ON_READ) { a.x = silu_v(a.x); a.y = silu_v(a.y); a.z = silu_v(a.z); a.w = silu_v(a.w); }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[t][cg][nt].x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[t][cg][nt].y, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[t][cg][nt].z, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[t][cg][nt].w, acc[nt], 0, 0, 0);
                }
            }
        }
        // D fragment: lane holds rows 4 * kk + i (i = 0..3), column m of each 16-wide N tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * kk + i;
            const bool ok = r >= valid0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float* o = OUT + (long)r * LD + nt * 16 + m;
                if (FIRST) *o = ok ? silu_v(acc[nt][i]) : 0.f;
                else if (ok) {
                    const float y = *o + acc[nt][i];
                    *o = y;
                    if (OUT_ACT) OUT_ACT[(long)r * LD + nt * 16 + m] = silu_v(y);      // the next pair's conv 1 reads it activated
                }
            }
        }
    }
}

struct VocLevelArgs {
    const float* X; long x_bstride; int xH;          // level input: xH history rows, then Tl new rows, C floats per row
    int Tl, TR, rows_alloc, three;                   // three: a third LDS buffer holds silu(y) (conv 1 then reads it without the k-fold activation)
    const float* W[3][6]; const float* bias[3][6];   // per branch: c1, c2 of dilation 0, 1, 2
    int dil[3];
    float* Y3[3]; long y_bstride;                    // branch outputs, rows [0, Tl)
    const int* frames_done; int rows_per_frame;      // level rows in front of this call = *frames_done * rows_per_frame
};

template <int C, int K>
__device__ __forceinline__ void run_branch(const VocLevelArgs& a, int br, float* smem) {
    constexpr int LD = C + 4;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, bi = blockIdx.z;
    const int t0 = tile * a.TR, t1 = min(t0 + a.TR, a.Tl);
    const int RF = (K - 1) * 2 * (a.dil[0] + a.dil[1] + a.dil[2]);
    const int n_rows = t1 - t0 + RF;                         // local row r <-> level row t0 - RF + r of this call
    float* Y = smem;
    float* T = smem + (long)a.rows_alloc * LD;
    float* S = a.three ? T + (long)a.rows_alloc * LD : nullptr;
    int fd = *a.frames_done;
    fd = fd > 4 ? 4 : fd;                                    // (only "is the halo inside the stream" matters; no overflow)
    const int valid0 = max(0, -(fd * a.rows_per_frame + t0 - RF));
    const float* xg = a.X + (long)bi * a.x_bstride + (long)(a.xH + t0 - RF) * C;
    for (int i = tid; i < n_rows * (C / 4); i += 256) {
        const int r = i / (C / 4), c4 = i % (C / 4);
        vf4 v = vf4{0.f, 0.f, 0.f, 0.f};
        if (r >= valid0) v = *reinterpret_cast<const vf4*>(xg + (long)r * C + c4 * 4);
        *reinterpret_cast<vf4*>(Y + (long)r * LD + c4 * 4) = v;
        if (S) *reinterpret_cast<vf4*>(S + (long)r * LD + c4 * 4) = vf4{silu_v(v.x), silu_v(v.y), silu_v(v.z), silu_v(v.w)};
    }
    __syncthreads();
    int consumed = 0;
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        const int d = a.dil[j];
        consumed += (K - 1) * d;
        if (S) conv_rows<C, K, true, false>(S, T, nullptr, a.W[br][2 * j], a.bias[br][2 * j], d, consumed, n_rows - consumed, valid0);
        else conv_rows<C, K, true, true>(Y, T, nullptr, a.W[br][2 * j], a.bias[br][2 * j], d, consumed, n_rows - consumed, valid0);
        __syncthreads();
        consumed += (K - 1) * d;
        conv_rows<C, K, false, false>(T, Y, j < 2 ? S : nullptr, a.W[br][2 * j + 1], a.bias[br][2 * j + 1], d, consumed, n_rows - consumed, valid0);
        __syncthreads();
    }
    float* yg = a.Y3[br] + (long)bi * a.y_bstride + (long)t0 * C;
    for (int i = tid; i < (t1 - t0) * (C / 4); i += 256) {
        const int r = i / (C / 4), c4 = i % (C / 4);
        *reinterpret_cast<vf4*>(yg + (long)r * C + c4 * 4) = *reinterpret_cast<const vf4*>(Y + (long)(RF + r) * LD + c4 * 4);
    }
}

template <int C>
__global__ __launch_bounds__(256) void voc_level_kernel(VocLevelArgs a) {
    extern __shared__ __align__(16) float smem[];
    const int br = blockIdx.y;
    if (br == 0) run_branch<C, 3>(a, 0, smem);
    else if (br == 1) run_branch<C, 7>(a, 1, smem);
    else run_branch<C, 11>(a, 2, smem);
}

}  // namespace

bool voc_level_supported(int C) { return C == 16 || C == 32; }

// X: level input (after the transposed conv); res[br][j]: the level's ResBlock convs; y3: the three branch outputs
int launch_voc_level(const float* X, long x_bstride, int xH, int C, int B, int Tl, const float* const W[3][6], const float* const bias[3][6],
                     const int dil[3], float* const y3[3], long y_bstride, const int* frames_done, int rows_per_frame, hipStream_t st) {
    SVA_CHECK(voc_level_supported(C), "voc_level: unsupported channel count");
    const int rf_max = 10 * 2 * (dil[0] + dil[1] + dil[2]);
    SVA_CHECK(xH >= rf_max, "voc_level: the level input keeps too little history");
    SVA_CHECK(rows_per_frame >= rf_max, "voc_level: a frame is shorter than the receptive field");
    // tile rows: as large as LDS allows once the launch fills the chip, smaller (more workgroups) for few streams
    const int three = 1;
    const int tr_max = three ? (C == 16 ? 256 : 128) : (C == 16 ? 512 : 256);
    int TR = 64;
    while (TR < tr_max && (long)((Tl + 2 * TR - 1) / (2 * TR)) * 3 * B >= 256) TR *= 2;
    VocLevelArgs a;
    a.X = X; a.x_bstride = x_bstride; a.xH = xH; a.Tl = Tl; a.TR = TR; a.rows_alloc = TR + rf_max + 16; a.three = three;
    const size_t act_bytes = sizeof(float) * (three ? 3 : 2) * (size_t)a.rows_alloc * (C + 4);
    for (int br = 0; br < 3; ++br) {
        for (int q = 0; q < 6; ++q) { a.W[br][q] = W[br][q]; a.bias[br][q] = bias[br][q]; }
        a.Y3[br] = y3[br];
    }
    for (int j = 0; j < 3; ++j) a.dil[j] = dil[j];
    a.y_bstride = y_bstride; a.frames_done = frames_done; a.rows_per_frame = rows_per_frame;
    const size_t smem = act_bytes;
    SVA_CHECK(smem <= 160 * 1024, "voc_level: tile does not fit LDS");
    const dim3 grid((Tl + TR - 1) / TR, 3, B);
    static DeviceOnce attr_done[2];
    if (C == 16) {
        if (attr_done[0].needed()) { SVA_HIP(hipFuncSetAttribute((const void*)voc_level_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done[0].done(); }
        hipLaunchKernelGGL(voc_level_kernel<16>, grid, dim3(256), smem, st, a);
    } else {
        if (attr_done[1].needed()) { SVA_HIP(hipFuncSetAttribute((const void*)voc_level_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done[1].done(); }
        hipLaunchKernelGGL(voc_level_kernel<32>, grid, dim3(256), smem, st, a);
    }
    SVA_HIP(hipGetLastError());
    return 0;
}

}  // namespace sva
