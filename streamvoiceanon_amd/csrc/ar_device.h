// Device helpers of the persistent AR decode kernel (ar_decode.hip): granule stores, per-wave weight fragments, the GEMV with a
// folded RMSNorm and the sort-free nucleus sampler.
#pragma once
#include "device_util.h"

#include <hip/hip_fp16.h>
#include <type_traits>

namespace sva {
namespace ardev {

typedef unsigned long long u64;
constexpr int D = 768, I = 2304, H = 12, NCB = 8;

__device__ __forceinline__ void store_granule(u64* g, unsigned ep, float v) {
    __hip_atomic_store(g, ((u64)ep << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


__device__ __forceinline__ float h_lo(unsigned u) { return __half2float(__ushort_as_half((unsigned short)(u & 0xffffu))); }
__device__ __forceinline__ float h_hi(unsigned u) { return __half2float(__ushort_as_half((unsigned short)(u >> 16))); }

// one weight row, spread over the 64 lanes of a wave
template <typename WT, int K> struct WFrag;
template <int K> struct WFrag<float, K> {
    float4 v[K / 256];
    __device__ __forceinline__ void load(const void* base, long row, int lane) {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + row * K);
#pragma unroll
        for (int j = 0; j < K / 256; ++j) v[j] = p[lane + 64 * j];
    }
    __device__ __forceinline__ float dot(const float (&x)[K / 64]) const {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < K / 256; ++j) {
            a = fmaf(x[4 * j], v[j].x, a);
            a = fmaf(x[4 * j + 1], v[j].y, a);
            a = fmaf(x[4 * j + 2], v[j].z, a);
            a = fmaf(x[4 * j + 3], v[j].w, a);
        }
        return a;
    }
};
template <int K> struct WFrag<__half, K> {
    static constexpr int NC = K / 512;          // 8-element chunks per lane, then a 4-element tail (K % 512 == 256)
    uint4 v[NC];
    uint2 t;
    __device__ __forceinline__ void load(const void* base, long row, int lane) {
        const __half* r = reinterpret_cast<const __half*>(base) + row * K;
#pragma unroll
        for (int j = 0; j < NC; ++j) v[j] = reinterpret_cast<const uint4*>(r)[lane + 64 * j];
        t = reinterpret_cast<const uint2*>(r + 512 * NC)[lane];
    }
    __device__ __forceinline__ float dot(const float (&x)[K / 64]) const {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            a = fmaf(x[8 * j], h_lo(v[j].x), a);
            a = fmaf(x[8 * j + 1], h_hi(v[j].x), a);
            a = fmaf(x[8 * j + 2], h_lo(v[j].y), a);
            a = fmaf(x[8 * j + 3], h_hi(v[j].y), a);
            a = fmaf(x[8 * j + 4], h_lo(v[j].z), a);
            a = fmaf(x[8 * j + 5], h_hi(v[j].z), a);
            a = fmaf(x[8 * j + 6], h_lo(v[j].w), a);
            a = fmaf(x[8 * j + 7], h_hi(v[j].w), a);
        }
        a = fmaf(x[8 * NC], h_lo(t.x), a);
        a = fmaf(x[8 * NC + 1], h_hi(t.x), a);
        a = fmaf(x[8 * NC + 2], h_lo(t.y), a);
        a = fmaf(x[8 * NC + 3], h_hi(t.y), a);
        return a;
    }
};
static_assert(D % 512 == 256 && I % 512 == 256, "fp16 fragment layout needs K % 512 == 256");

// the lane's elements of an activation / norm-weight row, in the order WFrag<WT, K>::dot consumes them
template <typename WT, int K>
__device__ __forceinline__ void load_x(const float* xl, int lane, float (&x)[K / 64]) {
    if constexpr (std::is_same<WT, float>::value) {
#pragma unroll
        for (int j = 0; j < K / 256; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(xl + 256 * j + 4 * lane);
            x[4 * j] = t.x; x[4 * j + 1] = t.y; x[4 * j + 2] = t.z; x[4 * j + 3] = t.w;
        }
    } else {
        constexpr int NC = K / 512;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const float4 t0 = *reinterpret_cast<const float4*>(xl + 512 * j + 8 * lane);
            const float4 t1 = *reinterpret_cast<const float4*>(xl + 512 * j + 8 * lane + 4);
            x[8 * j] = t0.x; x[8 * j + 1] = t0.y; x[8 * j + 2] = t0.z; x[8 * j + 3] = t0.w;
            x[8 * j + 4] = t1.x; x[8 * j + 5] = t1.y; x[8 * j + 6] = t1.z; x[8 * j + 7] = t1.w;
        }
        const float4 tt = *reinterpret_cast<const float4*>(xl + 512 * NC + 4 * lane);
        x[8 * NC] = tt.x; x[8 * NC + 1] = tt.y; x[8 * NC + 2] = tt.z; x[8 * NC + 3] = tt.w;
    }
}

// out[m][r] = (NORM ? rsqrt(mean(x_m^2) + eps) : 1) * sum_k W_r[k] * x_m[k] * (NORM ? nw[k] : 1); every lane gets every value
// (RMSNorm: modules/dual_ar_stream.py:985-990, folded into the projection like gemv_kernel does)
template <typename WT, int K, int ROWS, int M, bool NORM>
__device__ __forceinline__ void gemv(const WFrag<WT, K> (&w)[ROWS], const float* xl, int ldx, const float* nw, float eps, int lane,
                                     float (&out)[M][ROWS]) {
    float nwv[K / 64];
    if constexpr (NORM) load_x<WT, K>(nw, lane, nwv);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float x[K / 64];
        load_x<WT, K>(xl + m * ldx, lane, x);
        float s = 1.f;
        if constexpr (NORM) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < K / 64; ++i) ss = fmaf(x[i], x[i], ss);
            s = 1.f / sqrtf(wave_sum(ss) / (float)K + eps);
#pragma unroll
            for (int i = 0; i < K / 64; ++i) x[i] *= nwv[i];
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) out[m][r] = wave_sum(w[r].dot(x)) * s;
    }
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }

template <typename KVT> __device__ __forceinline__ float4 ld_kv4(const KVT* p);
template <> __device__ __forceinline__ float4 ld_kv4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld_kv4<__half>(const __half* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(h_lo(u.x), h_hi(u.x), h_lo(u.y), h_hi(u.y));
}
template <typename KVT> __device__ __forceinline__ void st_kv(KVT* p, float v);
template <> __device__ __forceinline__ void st_kv<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_kv<__half>(__half* p, float v) { *p = __float2half(v); }

__device__ __forceinline__ float row16_sum(float v) {        // sum over the 16 lanes of a DPP row, in every lane of the row
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

__device__ __forceinline__ unsigned umax_wave(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Nucleus sampler of logits_to_probs + multinomial_sample_one_no_sync (modules/dual_ar_stream.py:1092-1132) over PER
// logits per thread of NW waves (element of slot r: e0 + r * NW * 64); the sort-free threshold search of
// sampler_bisect_kernel (kernels.hip) with interpolated, key-snapped probes.  NW == 1: no barrier at all.  Returns the
// token in every participating thread.  red: LDS scratch of >= 64 doubles (NW > 1 only).
// ALWAYS inlined: as a called function its logits array lives in scratch memory, and -- what kept ar_batch.hip out of the two-build
// audit in round 4 -- hipcc's -amdgpu-waitcnt-forcezero mode puts an s_waitcnt between the s_getpc_b64 / s_add_u32 @rel32@lo+4 /
// s_addc_u32 @rel32@hi+12 of every call sequence, whose +4 / +12 assume adjacency: the s_swappc lands 4 bytes in front of the callee
// (HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION; /tmp disassembly).  No calls, no call sequences.
template <int NW, int PER>
__device__ __forceinline__ int nucleus_sample(const float (&l)[PER], int V, int e0, const float* noise, unsigned long long seed, int frame, int kind,
                              int noise_elem_off, float inv_temp, float top_p, double* red, long long* dbg = nullptr) {
#define NS_MARK(k) do { if (dbg && threadIdx.x == 0) dbg[k] = wall_clock64(); } while (0)
    NS_MARK(0);
    constexpr int ES = NW * 64;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % NW;
    int slot = 0;
    double* dred = red;                                     // [2][NW]
    unsigned* ured = reinterpret_cast<unsigned*>(red + 2 * NW);       // [2][2][NW]
    float* fred = reinterpret_cast<float*>(ured + 4 * NW);  // [2][NW]
    int* ired = reinterpret_cast<int*>(fred + 2 * NW);      // [NW]
    auto block_sum_d = [&](double x) {
        x = wave_sum_d(x);
        if constexpr (NW == 1) return x;
        if (lane == 0) dred[slot * NW + wave] = x;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += dred[slot * NW + w];
        slot ^= 1;
        return t;
    };
    auto block_sum_f = [&](float x) {
        x = wave_sum(x);
        if constexpr (NW == 1) return x;
        if (lane == 0) fred[slot * NW + wave] = x;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += fred[slot * NW + w];
        slot ^= 1;
        return t;
    };
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < PER; ++r) m = fmaxf(m, l[r]);
    m = wave_max(m);
    float mx = m;
    if constexpr (NW > 1) {
        if (lane == 0) fred[slot * NW + wave] = m;
        __syncthreads();
        mx = fred[slot * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) mx = fmaxf(mx, fred[slot * NW + w]);
        slot ^= 1;
    }
    float p[PER];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        p[r] = (e0 + r * ES) < V ? expf(l[r] - mx) : 0.f;
        s += p[r];
    }
    const float denom = block_sum_f(s);
    unsigned key[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        p[r] = p[r] / denom;
        key[r] = __builtin_bit_cast(unsigned, p[r]);
    }
    NS_MARK(1);
    bool keep[PER];
    double all = 0.0;
#pragma unroll
    for (int r = 0; r < PER; ++r) all += (double)p[r];
    if (!((float)block_sum_d(all) > top_p)) {
#pragma unroll
        for (int r = 0; r < PER; ++r) keep[r] = (e0 + r * ES) < V;
    } else {
        unsigned lo = 0u, hi = __builtin_bit_cast(unsigned, 1.0f / denom) + 2u;
        if (hi > 0x3F800001u) hi = 0x3F800001u;
        double f_lo = 1.0, f_hi = 0.0;
        int it = 0;
        while (hi - lo > 1u) {
            unsigned mid = lo + ((hi - lo) >> 1);
            if ((it % 3) != 2) {       // (placement only: any probe sequence keeps the bracket invariant, so float arithmetic is enough here)
                const float t = (float)(f_lo - (double)top_p) / (float)(f_lo - f_hi);
                const float off = (float)(hi - lo) * (t < 0.f ? 0.f : (t > 1.f ? 1.f : t));
                unsigned m2 = lo + (unsigned)off;
                if (m2 <= lo) m2 = lo + 1u;
                if (m2 >= hi) m2 = hi - 1u;
                mid = m2;
            }
            ++it;
            double t = 0.0;
            unsigned below = 0u, above_inv = 0u;
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
            double fm = t;
            unsigned kl = below, kgi = above_inv;
            if constexpr (NW > 1) {
                if (lane == 0) { dred[slot * NW + wave] = t; ured[(slot * 2) * NW + wave] = below; ured[(slot * 2 + 1) * NW + wave] = above_inv; }
                __syncthreads();
                fm = 0.0; kl = 0u; kgi = 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    fm += dred[slot * NW + w];
                    kl = max(kl, ured[(slot * 2) * NW + w]);
                    kgi = max(kgi, ured[(slot * 2 + 1) * NW + w]);
                }
                slot ^= 1;
            }
            if ((float)fm > top_p) { lo = ~kgi; f_lo = fm; }
            else { hi = kl + 1u; f_hi = fm; }
        }
        NS_MARK(2);
        if (dbg && threadIdx.x == 0) dbg[7] = it;
        const unsigned kb = lo;
        const float pb = __builtin_bit_cast(float, kb);
        double above = 0.0;
        float ties = 0.f;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            above += key[r] > kb ? (double)p[r] : 0.0;
            ties += (key[r] == kb && (e0 + r * ES) < V) ? 1.f : 0.f;
        }
        const double base = block_sum_d(above);
        const int cnt = (int)block_sum_f(ties);
        int nk = 0;
        double run = base;
        for (int j = 0; j < cnt; ++j) {
            run += (double)pb;
            if ((float)run > top_p) break;
            ++nk;
        }
        if (base == 0.0 && nk == 0) nk = 1;
        int id_cut = -1;
        if (nk >= cnt) id_cut = 0x7fffffff;
        else if (nk > 0) {
            int ilo = -1, ihi = V - 1;
            while (ihi - ilo > 1) {
                const int mid = ilo + ((ihi - ilo) >> 1);
                float c = 0.f;
#pragma unroll
                for (int r = 0; r < PER; ++r) c += (key[r] == kb && (e0 + r * ES) <= mid) ? 1.f : 0.f;
                if ((int)block_sum_f(c) >= nk) ihi = mid; else ilo = mid;
            }
            id_cut = ihi;
        }
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int e = e0 + r * ES;
            keep[r] = e < V && (key[r] > kb || (key[r] == kb && e <= id_cut));
        }
    }
    NS_MARK(3);
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
        const int e = e0 + r * ES;
        const float pr = e2[r] / denom2;
        const float q = noise ? noise[e] : exp1_noise_dev(seed, frame, kind, (unsigned)(noise_elem_off + e));
        const float rr = pr / q;
        if (rr > best || (rr == best && e < best_id)) { best = rr; best_id = e; }
    }
    NS_MARK(4);
    {   // wave argmax (ties: the smaller index) with DPP reductions only: the maximum, then the least index among its holders
        const float bm = wave_max(best);
        const unsigned cand = best == bm ? (unsigned)best_id : 0x7fffffffu;
        best_id = (int)~umax_wave(~cand);
        best = bm;
    }
    if constexpr (NW > 1) {
        if (lane == 0) { fred[slot * NW + wave] = best; ired[wave] = best_id; }
        __syncthreads();
        best = fred[slot * NW]; best_id = ired[0];
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (fred[slot * NW + w] > best || (fred[slot * NW + w] == best && ired[w] < best_id)) { best = fred[slot * NW + w]; best_id = ired[w]; }
        __syncthreads();
    }
    NS_MARK(5);
    return best_id;
}

}  // namespace ardev
}  // namespace sva
