// Device-side helpers shared by the kernel translation units (gfx950, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace sva {

// ------------------------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------------------------
// DPP reductions (no LDS round trips as with __shfl_xor = ds_bpermute): xor-1 / xor-2 inside quads, half-row and row
// mirrors give every lane its 16-lane row total; row_bcast15 / row_bcast31 chain the four rows into lane 63, which is
// broadcast through an SGPR.  All lanes return the same value.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);           // row_half_mirror
    v += dpp_mov<0x140>(v);           // row_mirror
    v += dpp_mov<0x142, 0xa>(v);      // row_bcast15 -> rows 1, 3
    v += dpp_mov<0x143, 0xc>(v);      // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    // bound_ctrl supplies 0 to lanes without a source: harmless for a sum, wrong for a max of negatives -> merge explicitly
    const float r15 = dpp_mov<0x142, 0xa>(v);
    v = ((threadIdx.x & 63) >= 16 && (((threadIdx.x & 63) >> 4) & 1)) ? fmaxf(v, r15) : v;
    const float r31 = dpp_mov<0x143, 0xc>(v);
    v = ((threadIdx.x & 63) >= 32) ? fmaxf(v, r31) : v;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// value of lane (l ^ DL): DPP for the distances a row (16 lanes) can serve, ds_bpermute beyond
template <int DL>
__device__ __forceinline__ int lane_xor_i(int v) {
    if constexpr (DL == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
    else if constexpr (DL == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    else if constexpr (DL == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);    // row_ror:8
    else if constexpr (DL == 4) {
        const int up = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, true);                    // row_shl:4  (from lane + 4)
        const int dn = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);                    // row_shr:4  (from lane - 4)
        return (threadIdx.x & 4) ? dn : up;
    } else return __shfl_xor(v, DL, 64);
}
template <int DL>
__device__ __forceinline__ float lane_xor_f(float v) { return __builtin_bit_cast(float, lane_xor_i<DL>(__builtin_bit_cast(int, v))); }

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float exp1_noise_dev(unsigned long long seed, int frame, int kind, unsigned elem) {
    // streamvoiceanon_amd/synth_weights.py: noise_key() + u24_from_key() + exp1_noise()
    unsigned long long z = (seed + 1ull) * 0xD6E8FEB86659FD93ull;
    z ^= ((unsigned long long)frame + 1ull) * 0x9E3779B97F4A7C15ull;
    z ^= ((unsigned long long)kind + 1ull) * 0xC2B2AE3D27D4EB4Full;
    const unsigned long long key = mix64(z);
    const unsigned long long hsh = mix64(key + ((unsigned long long)elem + 1ull) * 0x9E3779B97F4A7C15ull);
    const float k = (float)(unsigned)(hsh >> 40);
    return -logf(fmaxf(k, 0.5f) * (1.0f / 16777216.0f));
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_mov_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_mov_d<0xB1>(v);
    v += dpp_mov_d<0x4E>(v);
    v += dpp_mov_d<0x141>(v);
    v += dpp_mov_d<0x140>(v);
    v += dpp_mov_d<0x142, 0xa>(v);
    v += dpp_mov_d<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

}  // namespace sva
