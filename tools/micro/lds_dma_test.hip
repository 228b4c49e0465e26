#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* dst) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave: one 1 KiB DMA: lane L fetches the 16-byte chunk (63 - L) of its wave's source block
    const float* g = src + wave * 256 + (63 - lane) * 4;
    unsigned lds_dst = (unsigned)(size_t)(smem + wave * 256);
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) dst[i] = smem[i];
}
int main() {
    float h[1024], o[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
    float *d, *e; hipMalloc(&d, 4096); hipMalloc(&e, 4096); hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, e);
    hipMemcpy(o, e, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int L = 0; L < 64; ++L) for (int j = 0; j < 4; ++j) {
        float want = w * 256 + (63 - L) * 4 + j;   // LDS position L holds chunk 63-L
        if (o[w * 256 + L * 4 + j] != want) ++bad;
    }
    printf("bad=%d  o[0..7]= %g %g %g %g %g %g %g %g\n", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    return 0;
}
