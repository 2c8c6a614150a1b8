// which SIMD does wave w of a 768-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768) void k(unsigned *out)
{
    extern __shared__ unsigned char sm[];
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = v;
}
int main()
{
    unsigned *d, h[48];
    (void)hipMalloc(&d, sizeof h);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int threads : { 640, 768 }) {
        hipLaunchKernelGGL(k, dim3(4), dim3(threads), 150 * 1024, 0, d);
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b) { printf("%d threads, block %d (cu %2u se %u): simd of wave 0..%d:", threads, b, (h[b * 12] >> 8) & 15, (h[b * 12] >> 13) & 7, threads / 64 - 1); for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 12 + w] >> 4) & 3); printf("\n"); }
    }
    return 0;
}
