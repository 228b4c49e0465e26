// v_pk_fma_f32 with op_sel:[0,1,0] (low result lane <- HIGH register of the src1 pair) on gfx950, under a loaded launch.
// Reproduces, outside the engine, what the paired decode attention kernel met in round 6: 768 workgroups each stream their own
// key rows (distinct 512 KiB regions, so the loads are real HBM / L2 traffic) and form, from the SAME loaded registers, two 16-term dot products per
// key -- once as a packed chain of v_pk_fma_f32, once as plain v_fmac_f32 -- and count the keys whose two results differ in any bit, per 16-lane
// quarter of the wave.
//   FORM 0: odd terms read the key through op_sel:[0,1,0]             (what the compiler emits for fmaf pairs over a loaded float4)
//   FORM 1: odd terms read a copied pair {k, k} with op_sel_hi:[1,0,1] (no low-lane read of a high register)
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_probe tools/probes/pk_opsel_probe.hip && ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void fill_kernel(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = ((float)(unsigned)(z >> 40) * (1.f / 16777216.f) - 0.5f) * 4.f;
    }
}

template <int FORM>
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ keys, const float* __restrict__ q, int L, long region, unsigned* bad, float* sink) {
    const int tid = threadIdx.x, part4 = tid & 3, quarter = (tid & 63) >> 4;
    float q0[16], q1[16];
    const float* qr = q + (size_t)blockIdx.x * 128 + part4 * 16;
#pragma unroll
    for (int d = 0; d < 16; ++d) { q0[d] = qr[d]; q1[d] = qr[64 + d]; }
    const float* kc = keys + (size_t)blockIdx.x * region;
    float keep = 0.f;
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int j = j0 + (tid >> 2);
        if (j >= L) continue;
        const float* kr = kc + (size_t)j * 64 + part4 * 16;
        float k[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) k[d] = kr[d];
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 16; d += 2) {
            const f2 kk = {k[d], k[d + 1]};
            const f2 qa = {q1[d], q0[d]}, qb = {q1[d + 1], q0[d + 1]};
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(qa), "v"(kk));
            if constexpr (FORM == 0) {
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(qb), "v"(kk));
            } else {
                f2 kb = {kk.y, kk.y};
                asm("" : "+v"(kb));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(qb), "v"(kb));
            }
        }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            asm("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(q0[d]), "v"(k[d]));
            asm("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(q1[d]), "v"(k[d]));
        }
        if (__float_as_uint(acc.x) != __float_as_uint(a1)) atomicAdd(bad + quarter, 1u);            // low result lane
        if (__float_as_uint(acc.y) != __float_as_uint(a0)) atomicAdd(bad + 4 + quarter, 1u);        // high result lane
        keep += acc.x + acc.y;
    }
    if (keep == 12345.678f) sink[0] = keep;
}

// FORM 2 / 3: the same two chains inside the paired attention kernel's score loop as it was when it failed (scores of two query rows per key, quad exchange through
// ds_bpermute, the score stored to LDS by one lane of the quad under a narrowed EXEC, keys [0, L0) for row 0 and [0, L0 + 1) for row 1); the reference scores
// go through the same exchange from plain v_fmac_f32 chains; compared after the barrier.  FORM 2: odd terms through op_sel:[0,1,0]; FORM 3: through a copied pair.
template <int FORM>
__global__ __launch_bounds__(256) void probe_loop_kernel(const float* __restrict__ keys, const float* __restrict__ q, int L0, long region, unsigned* bad) {
    extern __shared__ float lds[];
    float* sc0 = lds; float* sc1 = sc0 + 2048; float* rf0 = sc1 + 2048; float* rf1 = rf0 + 2048;
    const int tid = threadIdx.x, part4 = tid & 3, L1 = L0 + 1;
    float q0[16], q1[16];
    const float* qr = q + (size_t)blockIdx.x * 128 + part4 * 16;
#pragma unroll
    for (int d = 0; d < 16; ++d) { q0[d] = qr[d]; q1[d] = qr[64 + d]; }
    const float* kc = keys + (size_t)blockIdx.x * region;
    for (int j0 = 0; j0 < L1; j0 += 64) {
        const int j = j0 + (tid >> 2);
        float a0 = 0.f, a1 = 0.f, r0 = 0.f, r1 = 0.f;
        if (j < L1) {
            const float* kr = kc + (size_t)j * 64 + part4 * 16;
            f2 acc = {0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 16; d += 2) {
                const f2 kk = {kr[d], kr[d + 1]};
                const f2 qa = {q1[d], q0[d]}, qb = {q1[d + 1], q0[d + 1]};
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(qa), "v"(kk));
                if constexpr (FORM == 2) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(qb), "v"(kk));
                } else {
                    f2 kb = {kk.y, kk.y};
                    asm("" : "+v"(kb));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(qb), "v"(kb));
                }
                asm("v_fmac_f32 %0, %1, %2" : "+v"(r0) : "v"(q0[d]), "v"(kk.x));
                asm("v_fmac_f32 %0, %1, %2" : "+v"(r1) : "v"(q1[d]), "v"(kk.x));
                asm("v_fmac_f32 %0, %1, %2" : "+v"(r0) : "v"(q0[d + 1]), "v"(kk.y));
                asm("v_fmac_f32 %0, %1, %2" : "+v"(r1) : "v"(q1[d + 1]), "v"(kk.y));
            }
            a1 = acc.x; a0 = acc.y;
        }
        a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
        a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64);
        r0 += __shfl_xor(r0, 1, 64); r1 += __shfl_xor(r1, 1, 64);
        r0 += __shfl_xor(r0, 2, 64); r1 += __shfl_xor(r1, 2, 64);
        if (j < L0) { if (part4 == 0) { sc0[j] = a0 * 0.125f; rf0[j] = r0 * 0.125f; } }
        if (j < L1) { if (part4 == 0) { sc1[j] = a1 * 0.125f; rf1[j] = r1 * 0.125f; } }
    }
    __syncthreads();
    for (int j = tid; j < L1; j += 256) {
        const int quarter = ((j & 63) >> 2) >> 2;          // the 16-lane quarter of the wave whose lanes computed key j (4 lanes per key)
        if (j < L0 && __float_as_uint(sc0[j]) != __float_as_uint(rf0[j])) atomicAdd(bad + 4 + quarter, 1u);
        if (__float_as_uint(sc1[j]) != __float_as_uint(rf1[j])) atomicAdd(bad + quarter, 1u);
    }
}

int main() {
    const int blocks = 768, L = 1024;
    const long region = 2048L * 64;                       // floats per workgroup: 512 KiB
    float *keys, *q, *sink;
    unsigned* bad;
    CK(hipMalloc(&keys, sizeof(float) * (size_t)blocks * region));
    CK(hipMalloc(&q, sizeof(float) * (size_t)blocks * 128));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&bad, 32));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, keys, (size_t)blocks * region);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, q, (size_t)blocks * 128);
    CK(hipDeviceSynchronize());
    for (int form = 0; form < 2; ++form) {
        unsigned tot[8] = {0};
        const int reps = 50;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(bad, 0, 32));
            if (form == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, keys, q, L, region, bad, sink);
            else hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, keys, q, L, region, bad, sink);
            CK(hipDeviceSynchronize());
            unsigned h[8];
            CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; ++i) tot[i] += h[i];
        }
        const double checks = (double)reps * blocks * L * 4;
        printf("form %d (%s): %d launches x %d workgroups x %d keys; mismatching (key, lane) results by 16-lane quarter of the wave --\n", form,
               form == 0 ? "odd terms through op_sel:[0,1,0]" : "odd terms through a copied pair", reps, blocks, L);
        printf("   low result lane : %u %u %u %u   high result lane: %u %u %u %u   (of %.3g checks each)\n", tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], tot[6],
               tot[7], checks);
    }
    for (int form = 2; form < 4; ++form) {
        unsigned tot[8] = {0};
        const int reps = 200;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(bad, 0, 32));
            const int L0 = 255 + 7 * (r % 97);
            if (form == 2) hipLaunchKernelGGL(probe_loop_kernel<2>, dim3(blocks), dim3(256), 4 * 2048 * sizeof(float), 0, keys, q, L0, region, bad);
            else hipLaunchKernelGGL(probe_loop_kernel<3>, dim3(blocks), dim3(256), 4 * 2048 * sizeof(float), 0, keys, q, L0, region, bad);
            CK(hipDeviceSynchronize());
            unsigned h[8];
            CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; ++i) tot[i] += h[i];
        }
        printf("form %d (the paired attention score loop, %s): %d launches x %d workgroups, 255 .. 927 keys; keys whose score differs from the v_fmac chain, by the 16-lane quarter that computed them --\n",
               form, form == 2 ? "odd terms through op_sel:[0,1,0]" : "odd terms through a copied pair", reps, blocks);
        printf("   row 1 (low result lane): %u %u %u %u   row 0 (high result lane): %u %u %u %u\n", tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], tot[6], tot[7]);
    }
    return 0;
}
