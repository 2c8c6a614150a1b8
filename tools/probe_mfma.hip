// raw v_mfma_f32_32x32x2_f32 rate: W waves per SIMD, A independent accumulators each, operands in registers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float acc_t __attribute__((ext_vector_type(16)));
template <int A>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *clk)
{
    acc_t acc[A];
    for (int a = 0; a < A; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < A; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < A; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main()
{
    float *out; unsigned long long *clk; hipMalloc(&out, 4 * 256 * 256 * 8); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 5}) for (int A : {1, 4}) {
        int blocks = 256 * wps, iters = 20000 / A;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (A == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * (double)iters * A * 4096.0;
        printf("waves/SIMD %d, %d accumulators: %.2f ms, %.1f TFLOP/s, clock %.0f MHz\n", wps, A, ms, flop / ms / 1e9, h[0] / (h[1] * 0.01));
    }
    return 0;
}
