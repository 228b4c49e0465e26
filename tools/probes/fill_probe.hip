// L2 -> CU fill rate on gfx950, per path: LDS-DMA (global_load_lds_dwordx4), plain 16-byte loads to VGPRs, or both at once.  Every workgroup (8 waves)
// re-reads its own 64 KiB region (L2-resident: 256 x 64 KiB = 2 MiB per XCD) with DEPTH wave-instructions in flight per wave.
// SHARE = s: groups of s workgroups read the SAME region in lockstep order (the W tile of a GEMM is read by every workgroup of its column);
// ROT: ... but each starts at its own piece (rotated order).
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe tools/probes/fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 1) void fill_kernel(const char* __restrict__ buf, int iters, unsigned* sink, int region, int share, int rot) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = buf + (size_t)(blockIdx.x / share) * region;
    // lane order inside a 1 KiB piece -- rot >= 10 selects: 10 = the planes GEMM's (row = lane & 15, 16-byte chunk = lane >> 4), 11 = rows in lane quads with the
    // chunk XOR-swizzled per row group (conflict-free fragment reads from a row-major image)
    const int fsw[4] = {0, 2, 3, 1};
    const unsigned voff = rot == 10 ? (unsigned)((lane & 15) * 64 + (lane >> 4) * 16) : rot == 11 ? (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ fsw[(lane >> 4) & 3]) * 16)) : lane * 16;
    u32x4 r[DEPTH];
    unsigned acc = 0;
    const int pieces = region / 1024;          // 1 KiB per wave-instruction
    int pc = wave + (rot && rot < 10 ? (int)(blockIdx.x % share) * 8 * rot : 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* p = base + (size_t)(pc % pieces) * 1024;
            pc += 8;
            const unsigned long long pu = (unsigned long long)p;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
            const unsigned long long ps = ((unsigned long long)hi << 32) | lo;
            const bool dma = MODE == 0 || (MODE == 2 && (d & 1) == 0);
            if (dma) {
                const unsigned dst = lds0 + (unsigned)((wave * DEPTH + d) * 1024);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(dst), "s"(ps) : "memory", "m0");
            } else {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[d]) : "v"(voff), "s"(ps) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool dma = MODE == 0 || (MODE == 2 && (d & 1) == 0);
            if (!dma) { asm volatile("" : "+v"(r[d])); acc ^= r[d].x; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The planes GEMM's request pattern without its arithmetic: workgroup = tile (tm, tn) of a 10880 x 1536 x 384 problem on K-blocked planes; per K step
// the 8 waves request 16 A pieces (hi / lo plane: 128 rows x 64 B = 8 KiB runs at k-block stride R x 64 B) and 16 W pieces, wait for them, optionally
// meet at a barrier.  AHEAD steps are requested before the first wait (ring of AHEAD + 1 stages).
template <int AHEAD, bool BARRIER, int READS = 0, int MFMAS = 0>
__global__ __launch_bounds__(512, 2) void gemm_pattern_kernel(const char* __restrict__ A, const char* __restrict__ W, int R, int N, int nk, int tiles_n, int n_tiles, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = lane * 16;
    const size_t a_ps = (size_t)R * nk * 64, w_ps = (size_t)N * nk * 64;
    int issued = 0;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
    f32x4_ acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    u32x4 keep = {0u, 0u, 0u, 0u};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int total = nk * rounds;
        auto issue = [&](int s) {
            const int k = s % nk;
            // this wave's four pieces: A plane 0 / 1 piece `wave`, W plane 0 / 1 piece `wave`
            for (int q = 0; q < 4; ++q) {
                const char* p = (q < 2 ? A + (size_t)(q & 1) * a_ps + ((size_t)k * R + (size_t)tm * 128 + wave * 16) * 64
                                       : W + (size_t)(q & 1) * w_ps + ((size_t)k * N + (size_t)tn * 128 + wave * 16) * 64);
                const unsigned long long pu = (unsigned long long)p;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
                const unsigned long long ps = ((unsigned long long)hi << 32) | lo;
                const unsigned dst = lds0 + (unsigned)(((s % (AHEAD + 1)) * 32 + q * 8 + wave) * 1024);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(dst), "s"(ps) : "memory", "m0");
            }
        };
        for (int s = 0; s < AHEAD && s < total; ++s) issue(s);
        for (int s = 0; s < total; ++s) {
            if (AHEAD <= 1 || s + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (AHEAD == 2 || s + 2 >= total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (BARRIER) __builtin_amdgcn_s_barrier();
            if (s + AHEAD < total) issue(s + AHEAD);
            // READS fragment reads (ds_read_b128, lane-linear) of the landed stage and MFMAS matrix instructions on them, as the GEMM's wave does
            u32x4 f[READS > 0 ? READS : 1];
            const char* st = reinterpret_cast<const char*>(smem) + (s % (AHEAD + 1)) * 32 * 1024 + lane * 16;
#pragma unroll
            for (int r = 0; r < READS; ++r) f[r] = *reinterpret_cast<const u32x4*>(st + ((r * 5 + wave) % 32) * 1024);
#pragma unroll
            for (int m = 0; m < MFMAS; ++m)
                acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_, f[READS > 0 ? m % READS : 0]), __builtin_bit_cast(f16x8_, f[READS > 0 ? (m + 3) % READS : 0]), acc[m % 8], 0, 0, 0);
            if (MFMAS == 0) {
#pragma unroll
                for (int r = 0; r < READS; ++r) keep ^= f[r];
            }
        }
        ++issued;
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += acc[i].x;
    if (t == 123.456f || keep.x == 0x12345678u) sink[0] = keep.x + (unsigned)t;
}
// the same step with LOADER WAVES: waves 8 .. 8 + L - 1 request all 32 pieces of a step, waves 0 .. 7 only read fragments and multiply; one barrier per step
template <int AHEAD, int L, int READS, int MFMAS>
__global__ __launch_bounds__(512 + 64 * L, 1) void gemm_spec_kernel(const char* __restrict__ A, const char* __restrict__ W, int R, int N, int nk, int tiles_n, int n_tiles, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = lane * 16;
    const size_t a_ps = (size_t)R * nk * 64, w_ps = (size_t)N * nk * 64;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
    constexpr int PPL = 32 / L;                    // pieces per loader wave and step
    if (wave >= 8) {
        const int lw = wave - 8;
        for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
            const int total = nk * rounds;
            auto issue = [&](int s) {
                const int k = s % nk;
#pragma unroll
                for (int i = 0; i < PPL; ++i) {
                    const int piece = lw + L * i, q = piece >> 3, rb = piece & 7;
                    const char* p = (q < 2 ? A + (size_t)(q & 1) * a_ps + ((size_t)k * R + (size_t)tm * 128 + rb * 16) * 64
                                           : W + (size_t)(q & 1) * w_ps + ((size_t)k * N + (size_t)tn * 128 + rb * 16) * 64);
                    const unsigned long long pu = (unsigned long long)p;
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
                    const unsigned long long ps = ((unsigned long long)hi << 32) | lo;
                    const unsigned dst = lds0 + (unsigned)(((s % (AHEAD + 1)) * 32 + piece) * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(dst), "s"(ps) : "memory", "m0");
                }
            };
            for (int s = 0; s < AHEAD && s < total; ++s) issue(s);
            for (int s = 0; s < total; ++s) {
                const int later = total - 1 - s < AHEAD - 1 ? total - 1 - s : AHEAD - 1;
                if (later <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPL > 63 ? 63 : 2 * PPL) : "memory");
                __builtin_amdgcn_s_barrier();
                if (s + AHEAD < total) issue(s + AHEAD);
            }
        }
        return;
    }
    f32x4_ acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int total = nk * rounds;
        for (int s = 0; s < total; ++s) {
            __builtin_amdgcn_s_barrier();
            u32x4 f[READS];
            const char* st = reinterpret_cast<const char*>(smem) + (s % (AHEAD + 1)) * 32 * 1024 + lane * 16;
#pragma unroll
            for (int r = 0; r < READS; ++r) f[r] = *reinterpret_cast<const u32x4*>(st + ((r * 5 + wave) % 32) * 1024);
#pragma unroll
            for (int m = 0; m < MFMAS; ++m)
                acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_, f[m % READS]), __builtin_bit_cast(f16x8_, f[(m + 3) % READS]), acc[m % 8], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += acc[i].x;
    if (t == 123.456f) sink[0] = (unsigned)t;
}
template <int AHEAD, int L, int READS, int MFMAS>
int run_spec(const char* A, const char* W, int cus, unsigned* sink) {
    const int R = 10880, N = 1536, nk = 12, tiles_n = N / 128, n_tiles = (R / 128) * tiles_n, rounds = 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t smem = (size_t)(AHEAD + 1) * 32 * 1024;
    CK(hipFuncSetAttribute((const void*)gemm_spec_kernel<AHEAD, L, READS, MFMAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((gemm_spec_kernel<AHEAD, L, READS, MFMAS>), dim3(cus), dim3(512 + 64 * L), smem, 0, A, W, R, N, nk, tiles_n, n_tiles, rounds, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)n_tiles * nk * rounds * 32 * 1024;
        if (rep) printf("GEMM pattern, %d loader waves + 8 consumers, ahead %d reads %2d mfmas %2d: %7.1f GB/s per CU, %.3f us per step and workgroup (%.3f ms)\n", L, AHEAD, READS, MFMAS,
                        bytes / ms * 1e-6 / cus, ms * 1e3 / ((double)n_tiles * nk * rounds / cus), ms);
    }
    return 0;
}

template <int AHEAD, bool BARRIER, int READS = 0, int MFMAS = 0>
int run_gemm(const char* A, const char* W, int cus, int wg_per_cu, unsigned* sink) {
    const int R = 10880, N = 1536, nk = 12, tiles_n = N / 128, n_tiles = (R / 128) * tiles_n, rounds = 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t smem = (size_t)(AHEAD + 1) * 32 * 1024;
    CK(hipFuncSetAttribute((const void*)gemm_pattern_kernel<AHEAD, BARRIER, READS, MFMAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((gemm_pattern_kernel<AHEAD, BARRIER, READS, MFMAS>), dim3(cus * wg_per_cu), dim3(512), smem, 0, A, W, R, N, nk, tiles_n, n_tiles, rounds, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)n_tiles * nk * rounds * 32 * 1024;
        if (rep) printf("GEMM pattern  ahead %d barrier %d reads %2d mfmas %2d wg/cu %d: %7.1f GB/s per CU, %6.2f TB/s chip, %.3f us per step and workgroup (%.3f ms)\n", AHEAD, (int)BARRIER, READS, MFMAS, wg_per_cu,
                        bytes / ms * 1e-6 / cus, bytes / ms * 1e-9, ms * 1e3 / ((double)n_tiles * nk * rounds / (cus * wg_per_cu)), ms);
    }
    return 0;
}

template <int MODE, int DEPTH>
int run(const char* buf, unsigned* sink, int cus, const char* name, int share = 1, int rot = 0) {
    const int region = 64 * 1024, iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(cus), dim3(512), 8 * DEPTH * 1024, 0, buf, iters, sink, region, share, rot);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)cus * iters * DEPTH * 8 * 1024;
        if (rep) printf("%-28s share %3d rot %d depth %2d: %7.1f GB/s per CU, %6.2f TB/s chip (%.3f ms)\n", name, share, rot, DEPTH, bytes / ms * 1e-6 / cus, bytes / ms * 1e-9, ms);
    }
    return 0;
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    char* buf; unsigned* sink;
    CK(hipMalloc((void**)&buf, (size_t)cus * 64 * 1024));
    CK(hipMemset(buf, 1, (size_t)cus * 64 * 1024));
    CK(hipMalloc((void**)&sink, 64));
    {
        char *A, *W;
        CK(hipMalloc((void**)&A, (size_t)2 * 10880 * 12 * 64 + 65536));
        CK(hipMalloc((void**)&W, (size_t)2 * 1536 * 12 * 64 + 65536));
        CK(hipMemset(A, 1, (size_t)2 * 10880 * 12 * 64)); CK(hipMemset(W, 1, (size_t)2 * 1536 * 12 * 64));
        unsigned* sk; CK(hipMalloc((void**)&sk, 64));
        if (run_gemm<1, true>(A, W, cus, 2, sk)) return 1;
        if (run_gemm<1, true, 12, 0>(A, W, cus, 2, sk)) return 1;
        if (run_gemm<1, true, 12, 24>(A, W, cus, 2, sk)) return 1;
        if (run_gemm<1, true, 0, 24>(A, W, cus, 2, sk)) return 1;
        if (run_gemm<3, true>(A, W, cus, 1, sk)) return 1;
        if (run_gemm<3, true, 12, 0>(A, W, cus, 1, sk)) return 1;
        if (run_gemm<3, true, 12, 24>(A, W, cus, 1, sk)) return 1;
        if (run_gemm<3, true, 0, 24>(A, W, cus, 1, sk)) return 1;
        if (run_gemm<3, true, 6, 24>(A, W, cus, 1, sk)) return 1;
        if (run_spec<3, 1, 12, 24>(A, W, cus, sk)) return 1;
        if (run_spec<3, 2, 12, 24>(A, W, cus, sk)) return 1;
        if (run_spec<3, 4, 12, 24>(A, W, cus, sk)) return 1;
        if (run_spec<3, 8, 12, 24>(A, W, cus, sk)) return 1;
        if (run_spec<2, 4, 12, 24>(A, W, cus, sk)) return 1;
    }
    if (run<0, 4>(buf, sink, cus, "LDS-DMA")) return 1;
    if (run<0, 8>(buf, sink, cus, "LDS-DMA")) return 1;
    if (run<0, 12>(buf, sink, cus, "LDS-DMA")) return 1;
    if (run<1, 4>(buf, sink, cus, "loads to VGPRs")) return 1;
    if (run<1, 8>(buf, sink, cus, "loads to VGPRs")) return 1;
    if (run<1, 12>(buf, sink, cus, "loads to VGPRs")) return 1;
    for (int ord : {0, 10, 11}) {
        if (run<0, 4>(buf, sink, cus, "LDS-DMA lane order", 1, ord)) return 1;
        if (run<0, 12>(buf, sink, cus, "LDS-DMA lane order", 1, ord)) return 1;
        if (run<1, 4>(buf, sink, cus, "VGPR loads lane order", 1, ord)) return 1;
    }
    for (int share : {64}) {
        if (run<0, 4>(buf, sink, cus, "LDS-DMA", share, 0)) return 1;
        if (run<0, 4>(buf, sink, cus, "LDS-DMA", share, 1)) return 1;
        if (run<0, 4>(buf, sink, cus, "LDS-DMA", share, 3)) return 1;
        if (run<0, 12>(buf, sink, cus, "LDS-DMA", share, 0)) return 1;
        if (run<0, 12>(buf, sink, cus, "LDS-DMA", share, 3)) return 1;
        if (run<1, 4>(buf, sink, cus, "loads to VGPRs", share, 0)) return 1;
    }
    if (run<2, 4>(buf, sink, cus, "half DMA, half VGPR loads")) return 1;
    if (run<2, 8>(buf, sink, cus, "half DMA, half VGPR loads")) return 1;
    if (run<2, 12>(buf, sink, cus, "half DMA, half VGPR loads")) return 1;
    return 0;
}
