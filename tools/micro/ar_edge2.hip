// Edge probe 2 for the persistent AR decode kernels: what does an all-to-all hand-off cost when the workgroup ALSO streams weights?
// G workgroups run P dependent phases.  In every phase a workgroup gathers the N-value vector of the previous phase (8-byte
// {tag, value} granules, sc1 stores / loads -- cdna_hip_programming.md Guideline 16 form R2), its compute waves "use" weights that
// were requested RING phases earlier (a register ring of RING sets of NL float4 per lane), publish their slice of the next vector and
// re-request the set they just consumed.  Variants:
//   comm = 0: all four waves poll (256 threads, the round-2 kernel);  comm = 1 / 2: one / two extra waves do nothing but poll
//   place = 0: weights requested right after the publish (just before the next gather starts);
//   place = 1: requested right after the barrier (before the phase's arithmetic);  place = 2: no weights at all
//   NL: float4 loads per lane per phase (9 = 9 KiB per wave = the wqkv rows of a wave at 192 workgroups)
// Prints microseconds per phase.
//   hipcc --offload-arch=gfx950 -O3 -o ar_edge2 ar_edge2.hip && ./ar_edge2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ void store_granule(u64* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// nthr threads (tid0 = index within the polling group) gather n granules into lds
template <int MAXPER>
__device__ __forceinline__ void gather(const u64* g, int n, unsigned epoch, float* lds, int* fail, int tid0, int nthr) {
    u64 pending = 0;
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) if (tid0 + k * nthr < n) pending |= 1ull << k;
    for (int spins = 0; pending; ++spins) {
        u64 x[MAXPER];
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if (pending >> k & 1) x[k] = __hip_atomic_load(g + tid0 + k * nthr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if ((pending >> k & 1) && (unsigned)(x[k] >> 32) == epoch) { lds[tid0 + k * nthr] = __uint_as_float((unsigned)x[k]); pending &= ~(1ull << k); }
        if (spins > 2000000) { *fail = 1; break; }
    }
}

// 16-byte polls: one buffer_load_dwordx4 sc1 covers two granules (each still validated by its own tag)
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MAXPER, int DEPTH>     // MAXPER pairs per thread; DEPTH = poll rounds kept in flight
__device__ __forceinline__ void gather16(const u64* g, int n, unsigned epoch, float* lds, int* fail, int tid0, int nthr) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(g), 0, ((n + 1) & ~1) * 8, 0x00020000);
    unsigned pending = 0;
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) if (2 * (tid0 + k * nthr) < n) pending |= 1u << k;
    v4i x[DEPTH][MAXPER];
    auto issue = [&](int d) {
        asm volatile("" ::: "memory");      // (the buffer-load builtin is a plain read to the optimiser: keep it inside the poll loop)
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if (pending >> k & 1) x[d][k] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid0 + k * nthr) * 16, 0, 16));
    };
    auto check = [&](int d) {
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if ((pending >> k & 1) && (unsigned)x[d][k].y == epoch && ((unsigned)x[d][k].w == epoch || 2 * (tid0 + k * nthr) + 1 >= n)) {
                const int i = 2 * (tid0 + k * nthr);
                lds[i] = __int_as_float(x[d][k].x);
                if (i + 1 < n) lds[i + 1] = __int_as_float(x[d][k].z);
                pending &= ~(1u << k);
            }
    };
    if constexpr (DEPTH == 1) {
        for (int spins = 0; pending; ++spins) {
            issue(0); check(0);
            if (spins > 2000000) { *fail = 1; break; }
        }
    } else {
        issue(0);
        for (int spins = 0; pending; ++spins) {
            issue(1); check(0);
            if (!pending) break;
            issue(0); check(1);
            if (spins > 2000000) { *fail = 1; break; }
        }
    }
}

template <int NL, int COMM, int PLACE>
__global__ __launch_bounds__(256 + 64 * (COMM >= 10 ? 0 : COMM), 2) void phases(u64* bufA, u64* bufB, int n_small, int n_big, int P, unsigned epoch0, float* out, int* fail,
                                                              const float4* wts, long wts_elems) {
    extern __shared__ float lds[];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_comm = COMM > 0 && COMM < 10 && wave >= 4;
    constexpr int RING = 4;
    float4 w[RING][NL > 0 ? NL : 1];
    long wpos = ((long)(wg * 4 + (wave & 3)) * 64 + lane) % wts_elems;
    const long wstep = (long)G * 4 * 64;
    auto issue = [&](int slot) {
        if constexpr (NL > 0 && PLACE != 2) {
#pragma unroll
            for (int j = 0; j < NL; ++j) { w[slot][j] = wts[wpos]; wpos += wstep; if (wpos >= wts_elems) wpos -= wts_elems; }
            asm volatile("" ::: "memory");
        }
    };
    float acc = 0.f;
    if (!is_comm) {
#pragma unroll
        for (int s = 0; s < RING; ++s) issue(s);
    }
#pragma unroll 1
    for (int p4 = 0; p4 < P; p4 += RING) {
#pragma unroll
        for (int ps = 0; ps < RING; ++ps) {
            const int p = p4 + ps;
            const int n_in = (p & 1) ? n_big : n_small, n_out = (p & 1) ? n_small : n_big;
            u64* in = (p & 1) ? bufB : bufA;
            u64* outb = (p & 1) ? bufA : bufB;
            const unsigned ep = epoch0 + p;
            if (p > 0) {
                if (COMM == 0) gather<9>(in, n_in, ep, lds, fail, tid, 256);
                else if (COMM == 10) gather16<5, 1>(in, n_in, ep, lds, fail, tid, 256);
                else if (COMM == 11) gather16<5, 2>(in, n_in, ep, lds, fail, tid, 256);
                else if (is_comm) gather<(COMM == 1 ? 36 : 18)>(in, n_in, ep, lds, fail, tid - 256, 64 * COMM);
            } else if (tid < 256) { for (int i = tid; i < n_in; i += 256) lds[i] = 1.0f; }
            __syncthreads();
            if (!is_comm) {
                if (PLACE == 1) { /* re-request the set consumed in the PREVIOUS phase now */ }
                float s = 0.f;
                for (int i = lane; i < n_in; i += 64) s += lds[i];
                if constexpr (NL > 0 && PLACE != 2) {
#pragma unroll
                    for (int j = 0; j < NL; ++j) s += w[ps][j].x * 1e-9f + w[ps][j].y * 1e-9f + w[ps][j].z * 1e-9f + w[ps][j].w * 1e-9f;
                }
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                acc += s;
                if (PLACE == 1) issue(ps);
                const int per = (n_out + G - 1) / G;
                const int lo = wg * per, hi = min(n_out, lo + per);
                if (lo + tid < hi) store_granule(outb + lo + tid, ep + 1, s * 1e-6f + (float)(lo + tid));
                if (PLACE == 0) issue(ps);
            }
            if (COMM == 0 || COMM >= 10) __syncthreads();
        }
    }
    if (tid == 0) out[wg] = acc;
}

template <int NL, int COMM, int PLACE>
int run(const char* label, int G, hipStream_t st, u64* bufA, u64* bufB, float* out, int* fail, const float4* wts, long wts_elems, unsigned& epoch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int P = 2000;
    float best = 1e9f; int f = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((phases<NL, COMM, PLACE>), dim3(G), dim3(256 + 64 * (COMM >= 10 ? 0 : COMM)), 40 * 1024, st, bufA, bufB, 768, 2304, P, epoch, out, fail, wts, wts_elems);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        epoch += P + 8;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    }
    printf("%-44s G=%3d NL=%2d (%4.1f KiB/wave/phase): %.2f us per phase, fail=%d\n", label, G, NL, NL * 1.0, best * 1e3 / P, f);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    u64 *bufA, *bufB; float* out; int* fail; float4* wts;
    const long wts_elems = (long)512 * 1024 * 1024 / 16;       // 512 MiB of "weights": beyond the 256 MiB MALL
    CK(hipMalloc(&bufA, 8 * 4096)); CK(hipMalloc(&bufB, 8 * 4096)); CK(hipMalloc(&out, 4 * 1024)); CK(hipMalloc(&fail, 4));
    CK(hipMalloc(&wts, wts_elems * 16));
    CK(hipMemset(bufA, 0, 8 * 4096)); CK(hipMemset(bufB, 0, 8 * 4096)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(wts, 0, wts_elems * 16));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned epoch = 1;
    const long small = (long)32 * 1024 * 1024 / 16;            // 32 MiB: stays in the MALL (the fast AR's weights)
    for (int G : {96, 192}) {
        run<0, 0, 2>("8-B polls, no weights", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<0, 10, 2>("16-B polls, no weights", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<0, 11, 2>("16-B polls x2 in flight, no weights", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<9, 0, 0>("8-B polls, weights after publish", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<9, 10, 0>("16-B polls, weights after publish", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<9, 11, 0>("16-B polls x2, weights after publish", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<9, 10, 1>("16-B polls, weights after barrier", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<9, 10, 0>("16-B polls, after publish, MALL-resident", G, st, bufA, bufB, out, fail, wts, small, epoch);
        run<12, 10, 0>("16-B polls, weights after publish", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
        run<15, 10, 0>("16-B polls, weights after publish", G, st, bufA, bufB, out, fail, wts, wts_elems, epoch);
    }
    return 0;
}
