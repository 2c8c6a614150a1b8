// which ingredient of the K9 stage loop costs MFMA rate?  V0: registers only; V1: + LDS operand reads;
// V2: + one __syncthreads per 16 MFMAs; V3: + ds_write_b128 x4 per stage; V4: + 4 global dwordx4 loads per stage
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float acc_t __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void k(float *out, const float *W, int ldw, int stages, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(16))) float sA[2][32][64], sB[2][32][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave >> 1, wj = wave & 1;
    for (int i = tid; i < 2 * 32 * 64; i += 256) { (&sA[0][0][0])[i] = i * 1e-4f; (&sB[0][0][0])[i] = 1.f - i * 1e-4f; }
    __syncthreads();
    acc_t acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    v4 ra[2], rb[2];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = tid * 1e-3f, y = 1.f - tid * 1e-3f;
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        if (V >= 4) {
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                int v = tid + l * 256, kr = v / 16, cv = (v % 16) * 4;
                ra[l] = *reinterpret_cast<const v4 *>(W + (size_t)((s & 15) * 32 + kr) * ldw + blockIdx.x % 40 * 64 + cv);
                rb[l] = *reinterpret_cast<const v4 *>(W + (size_t)((s & 15) * 32 + kr) * ldw + (blockIdx.x % 40 + 3) * 64 + cv);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float a = x, b = y;
            if (V >= 1) { int krow = ks * 2 + (lane >> 5); a = sA[buf][krow][wi * 32 + (lane & 31)]; b = sB[buf][krow][wj * 32 + (lane & 31)]; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (V >= 3) {
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                int v = tid + l * 256, kr = v / 16, cv = (v % 16) * 4;
                v4 wa = V >= 4 ? ra[l] : v4{ x, y, x, y }, wb = V >= 4 ? rb[l] : v4{ y, x, y, x };
                *reinterpret_cast<v4 *>(&sA[buf ^ 1][kr][cv]) = wa;
                *reinterpret_cast<v4 *>(&sB[buf ^ 1][kr][cv]) = wb;
            }
        }
        if (V >= 2) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sacc = 0; for (int e = 0; e < 16; ++e) sacc += acc[e];
    out[blockIdx.x * 256 + tid] = sacc;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main()
{
    float *out, *W; unsigned long long *clk; hipMalloc(&out, 4 * 256 * 256 * 8); hipMalloc(&clk, 16);
    const int ldw = 3136; hipMalloc(&W, sizeof(float) * 512 * ldw); hipMemset(W, 0, sizeof(float) * 512 * ldw);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int stages = 400;
    for (int wps : {1, 3, 5}) for (int V = 0; V <= 4; ++V) {
        int blocks = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            switch (V) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, W, ldw, stages, clk); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, W, ldw, stages, clk); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, W, ldw, stages, clk); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, W, ldw, stages, clk); break;
            default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, W, ldw, stages, clk); break;
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * (double)stages * 16 * 4096.0;
        printf("WG/CU %d  V%d: %.3f ms  %.1f TFLOP/s  clock %.0f MHz\n", wps, V, ms, flop / ms / 1e9, h[0] / (h[1] * 0.01));
    }
    return 0;
}
