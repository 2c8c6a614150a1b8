// A neighbour for tools/overlap_probe.py: an HBM-streaming kernel (copy of a 36 MB buffer) whose workgroups carry `lds_kb` of LDS, so that
// they can (small) or cannot (> 85 KB) share a CU with a 75 KB panel workgroup of k_chol_step.  Build: hipcc --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>
extern "C" {
__global__ __launch_bounds__(256) void nb_stream(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4, int mfma_iters)
{
    extern __shared__ float pad[];
    if (threadIdx.x == 0) pad[0] = 0.f;
    typedef float v16f __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    v16f acc = {};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = src[i];
        if (mfma_iters) {
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)v.x; b[j] = (__bf16)v.y; }
            for (int t = 0; t < mfma_iters; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            v.z += acc[0];
        }
        dst[i] = v;
    }
}
static float4 *g_a = nullptr, *g_b = nullptr;
static hipStream_t g_s = nullptr;
static const size_t N4 = (size_t)36 * 1024 * 1024 / 16;
__attribute__((visibility("default"))) int nb_run(int reps, int lds_kb, int wgs, int mfma_iters)
{
    if (!g_a) {
        if (hipMalloc(&g_a, N4 * 16) != hipSuccess || hipMalloc(&g_b, N4 * 16) != hipSuccess || hipStreamCreate(&g_s) != hipSuccess) return -1;
        hipMemset(g_a, 0, N4 * 16);
        hipFuncSetAttribute((const void *)nb_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(nb_stream, dim3(wgs), dim3(256), (size_t)lds_kb * 1024, g_s, g_a, g_b, N4, mfma_iters);
    return hipStreamSynchronize(g_s) == hipSuccess ? 0 : -2;
}
}
