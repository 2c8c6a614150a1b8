// issue-to-issue cost of the matrix instructions the matchers use (one wave per SIMD, 4 independent accumulators, operands in registers):
// v_mfma_f32_16x16x32_bf16, v_mfma_f32_32x32x16_bf16, v_mfma_i32_16x16x64_i8, v_mfma_i32_32x32x32_i8.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma3 tools/probe_mfma3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int NA = 4>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *clk)
{
    v4i xa = { (int)threadIdx.x, 1, 2, 3 }, xb = { 5, (int)threadIdx.x, 7, 8 };
    v4f a4[NA] = {}; v16f a16[4] = {}; v4i i4[NA] = {}; v16i i16[4] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < (KIND == 0 || KIND == 2 ? NA : 4); ++a) {
            if (KIND == 0) a4[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, xb), a4[a], 0, 0, 0);
            if (KIND == 1) a16[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, xb), a16[a], 0, 0, 0);
            if (KIND == 2) i4[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa, xb, i4[a], 0, 0, 0);
            if (KIND == 3) i16[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, xb, i16[a], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < 4; ++a) { s += a16[a][0] + (float)i16[a][0]; }
    for (int a = 0; a < NA; ++a) { s += a4[a][0] + (float)i4[a][0]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *clk; hipMalloc(&out, 4 * 256 * 256); hipMalloc(&clk, 16);
    const char *names[4] = { "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_i32_16x16x64_i8", "v_mfma_i32_32x32x32_i8" };
    const int iters = 5000;
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, clk);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, clk);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, clk);
            if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters, clk);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.1f shader cycles per instruction (one wave per SIMD, 4 independent accumulators)\n", names[kind], (double)h / (iters * 4.0));
    }
    // the 16x16 forms with more independent accumulators: is the 4-accumulator figure the instruction's own issue interval or its latency / 4?
    for (int na : { 8, 12 }) {
        for (int rep = 0; rep < 2; ++rep) {
            if (na == 8) hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(256), 0, 0, out, iters, clk);
            else hipLaunchKernelGGL((k<0, 12>), dim3(256), dim3(256), 0, 0, out, iters, clk);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f32_16x16x32_bf16     %.1f shader cycles per instruction with %d independent accumulators\n", (double)h / (iters * (double)na), na);
    }
    return 0;
}
