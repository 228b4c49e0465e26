// Device-side split of fp32 values into the fp16 operand planes of gemm_planes.hip (H3: hi + lo, H1: hi only), shared by the GEMM's own
// epilogue and by the non-GEMM producers that hand their output to a planes GEMM (dwconv7 + LayerNorm, RMSNorm rows).
#pragma once
#include <hip/hip_runtime.h>

namespace sva {

// hi = fp16(x); lo = fp16(x - hi).  The residual is taken from the ROUNDED hi whatever the optimiser makes of the casts (an opaque copy:
// with the product feeding the split contracted, hi and lo came from two different roundings once -- gemm_f16w.hip, split8)
__device__ __forceinline__ void h3_split(float x, unsigned short& hi, unsigned short& lo) {
    _Float16 h = (_Float16)x;
    unsigned short hb = __builtin_bit_cast(unsigned short, h);
    unsigned hw = hb;
    asm("" : "+v"(hw));
    hb = (unsigned short)hw;
    const _Float16 hq = __builtin_bit_cast(_Float16, hb);
    hi = hb;
    lo = __builtin_bit_cast(unsigned short, (_Float16)(x - (float)hq));
}

// Element offset of (row, k) in a plane.  rows == 0: the tensor's own row-major index space (row * ld + k, the caller adds batch offsets);
// rows > 0: K-BLOCKED -- [k / 32][rows][32]: the 32 k of a row that one MFMA step consumes are contiguous (64 bytes), and the 16 rows
// of a 1 KiB operand piece are adjacent, so the LDS-DMA of a piece reads ONE contiguous KiB (full cache lines) instead of sixteen
// 64-byte halves of sixteen lines (profiles/r05_planes_dma_probe_*.txt: the K loop's DMA 74 -> 56 us, 63 -> 39 us)
__device__ __forceinline__ long plane_off_blocked(long row, int k, long rows) { return ((long)(k >> 5) * rows + row) * 32 + (k & 31); }

}  // namespace sva
