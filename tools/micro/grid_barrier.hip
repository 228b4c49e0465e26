// Grid-barrier probe: cost of a device-wide barrier inside one persistent kernel (atomic counter + agent-scope fences)
// as a function of the number of resident workgroups, with a data hand-off check across workgroups (and XCDs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// every round: each workgroup publishes a vector, barrier, reads the neighbour's vector (written by another CU/XCD)
__global__ void persist(unsigned* counter, float* buf, int rounds, int vec, int* errors) {
    const unsigned nb = gridDim.x;
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < vec; i += blockDim.x) buf[(size_t)blockIdx.x * vec + i] = (float)(r * 131 + blockIdx.x + i);
        grid_barrier(counter, (unsigned)(2 * r + 1) * nb);
        const unsigned src = (blockIdx.x + nb / 2 + 1) % nb;
        for (int i = threadIdx.x; i < vec; i += blockDim.x)
            if (buf[(size_t)src * vec + i] != (float)(r * 131 + src + i)) ++bad;
        grid_barrier(counter, (unsigned)(2 * r + 2) * nb);
    }
    if (bad) atomicAdd(errors, bad);
}

int main() {
    unsigned* d_cnt; float* d_buf; int* d_err;
    CK(hipMalloc(&d_cnt, 4)); CK(hipMalloc(&d_buf, 4 * 2048 * 1024)); CK(hipMalloc(&d_err, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 2000;
    const int grids[] = {32, 64, 128, 256, 512, 1024};
    const int threads[] = {64, 256, 1024};
    for (int t : threads)
        for (int gsz : grids) {
            if (gsz * t > 256 * 2048) continue;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(d_cnt, 0, 4, st)); CK(hipMemsetAsync(d_err, 0, 4, st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(persist, dim3(gsz), dim3(t), 0, st, d_cnt, d_buf, rounds, 768, d_err);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int err; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
                if (rep) printf("grid %4d x %4d threads: %.2f us per barrier (incl. 768-float publish/read), hand-off errors %d\n", gsz, t,
                                ms * 1e3 / (2 * rounds), err);
            }
        }
    return 0;
}
