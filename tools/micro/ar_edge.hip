// Edge probe for the persistent AR decode kernel: G workgroups x 256 threads run P dependent phases; in every phase each
// workgroup gathers the N-value vector the previous phase published (8-byte {tag, value} granules, sc1 stores / sc1 loads,
// no flags, no fences -- cdna_hip_programming.md Guideline 16 form R2), does a token amount of arithmetic on it and
// publishes its slice of the next vector.  Prints microseconds per phase: the price of one all-to-all dependency edge,
// which is what bounds a batch-1 decode chain of ~200 tiny GEMVs.
//   hipcc --offload-arch=gfx950 -O3 -o ar_edge ar_edge.hip && ./ar_edge
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ void store_granule(u64* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all 256 threads: gather n granules into lds[0..n) (each thread owns granules tid, tid+256, ...; re-polls until its tags match)
template <int MAXPER>
__device__ __forceinline__ bool gather(const u64* g, int n, unsigned epoch, float* lds, int* fail) {
    const int tid = threadIdx.x;
    unsigned pending = 0;
    for (int k = 0; k < MAXPER; ++k) if (tid + k * 256 < n) pending |= 1u << k;
    for (int spins = 0; pending; ++spins) {
        u64 x[MAXPER];
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if (pending & (1u << k)) x[k] = __hip_atomic_load(g + tid + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if ((pending & (1u << k)) && (unsigned)(x[k] >> 32) == epoch) { lds[tid + k * 256] = __uint_as_float((unsigned)x[k]); pending &= ~(1u << k); }
        if (spins > 2000000) { *fail = 1; break; }
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(256, 1) void phases(u64* bufA, u64* bufB, int n_small, int n_big, int P, unsigned epoch0, float* out, int* fail) {
    extern __shared__ float lds[];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    // phase p reads vector of size n(p) from buf[p&1], writes n(p+1) values to buf[(p+1)&1]; sizes alternate small, small, big, small
    // (x -> qkv(big) -> x -> g(big) -> x ...: here: pattern of a transformer layer: 768 -> 2304 -> 768 -> 2304 -> 768)
    float acc = 0.f;
    for (int p = 0; p < P; ++p) {
        const int n_in = (p & 1) ? n_big : n_small, n_out = (p & 1) ? n_small : n_big;
        u64* in = (p & 1) ? bufB : bufA;
        u64* outb = (p & 1) ? bufA : bufB;
        const unsigned ep = epoch0 + p;
        if (p > 0) gather<9>(in, n_in, ep, lds, fail);
        else { for (int i = tid; i < n_in; i += 256) lds[i] = 1.0f; __syncthreads(); }
        // token compute: each thread sums a few values (stands for the dot products whose weights sit in registers)
        float s = 0.f;
        for (int i = tid & 63; i < n_in; i += 64) s += lds[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        acc += s;
        // publish this workgroup's slice of the output vector
        const int per = (n_out + G - 1) / G;
        const int lo = wg * per, hi = min(n_out, lo + per);
        if (lo + tid < hi) store_granule(outb + lo + tid, ep + 1, s * 1e-6f + (float)(lo + tid));
        __syncthreads();
    }
    if (tid == 0) out[wg] = acc;
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    u64 *bufA, *bufB; float* out; int* fail;
    CK(hipMalloc(&bufA, 8 * 4096)); CK(hipMalloc(&bufB, 8 * 4096)); CK(hipMalloc(&out, 4 * 1024)); CK(hipMalloc(&fail, 4));
    CK(hipMemset(bufA, 0, 8 * 4096)); CK(hipMemset(bufB, 0, 8 * 4096)); CK(hipMemset(fail, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int P = 2000;
    unsigned epoch = 1;
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t st;
        if (masked) {
            uint32_t mask[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0};      // CUs 0..95
            CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        } else CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const int grids[] = {32, 48, 64, 96, 128, 192, 256};
        for (int G : grids) {
            if (masked && G > 96) continue;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(phases, dim3(G), dim3(256), 100 * 1024, st, bufA, bufB, 768, 2304, P, epoch, out, fail);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                epoch += P + 8;
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                if (rep) printf("%s G=%3d: %.2f us per phase (768 <-> 2304 granules), fail=%d\n", masked ? "cu-mask 0..95" : "whole chip  ", G, ms * 1e3 / P, f);
            }
        }
    }
    return 0;
}
