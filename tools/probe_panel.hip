// micro-probe: time k_chol_panel variants in isolation and read the shader clock (s_memtime vs s_memrealtime)
#define PRE3_PROBE 1
#include "../3pre_amd/csrc/pre3_update.hip"
#include <vector>
#include <cstdlib>
namespace pre3 { void set_error(const char*, ...) {} int launch_update_x(pre3_ctx*, int, int) { return 0; } int launch_jnorm(pre3_ctx*, int) { return 0; } }
using namespace pre3;

__global__ void k_clock(unsigned long long *out, int spin)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_loop_probe(float *out, int steps)
{
    __shared__ float buf[2][64];
    float v[4][4];
    int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    for (int p = 0; p < 4; ++p) for (int q = 0; q < 4; ++q) v[p][q] = 1.0f + 0.001f * (tid + p + q);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < steps; ++c) {
        int b = c & 1;
        if (tc == (c & 15)) for (int p = 0; p < 4; ++p) buf[b][tr + 16 * p] = v[p][(c >> 4) & 3];
        __syncthreads();
        float piv = buf[b][c & 63];
        float inv = MODE == 0 ? 1.0f / piv : __builtin_amdgcn_rcpf(piv);
        float li[4], lj[4];
        for (int p = 0; p < 4; ++p) { li[p] = buf[b][tr + 16 * p] * inv; lj[p] = buf[b][tc + 16 * p]; }
        for (int p = 0; p < 4; ++p) for (int q = 0; q < 4; ++q) v[p][q] -= 1e-6f * li[p] * lj[q];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int p = 0; p < 4; ++p) for (int q = 0; q < 4; ++q) s += v[p][q];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

int main()
{
    const int r_pad = 640, ldw = 3072 + 64, nrb = r_pad / 64, nW = ldw / 64;
    float *S, *W; int *status; unsigned long long *clk; float *po;
    hipMalloc(&S, sizeof(float) * r_pad * r_pad); hipMalloc(&W, sizeof(float) * r_pad * ldw); hipMalloc(&status, 4); hipMalloc(&clk, 64); hipMalloc(&po, 4 * 256 * 64);
    std::vector<float> hS((size_t)r_pad * r_pad), hW((size_t)r_pad * ldw);
    for (int i = 0; i < r_pad; ++i) for (int j = 0; j < r_pad; ++j) hS[(size_t)i * r_pad + j] = (i == j ? 50.f : 0.f) + 0.01f * ((i * 7 + j * 13) % 17 + (j * 7 + i * 13) % 17);
    for (auto &w : hW) w = (rand() % 1000) * 1e-3f;
    hipMemcpy(S, hS.data(), hS.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chol_panel<float>, dim3(1 + (nrb - 1) + nW), dim3(320), 0, 0, S, r_pad, W, ldw, 0, nrb, status);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("k_chol_panel J=0 (59 WGs): %.1f us\n", ms * 1e3);
        hipMemcpy(S, hS.data(), hS.size() * 4, hipMemcpyHostToDevice);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chol_panel<float>, dim3(1), dim3(320), 0, 0, S, r_pad, W, ldw, 0, nrb, status);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("k_chol_panel diag only (1 WG): %.1f us\n", ms * 1e3);
        { unsigned long long g[16]; hipMemcpyFromSymbol(g, HIP_SYMBOL(pre3::g_probe), sizeof g);
          printf("   cycles: load %llu, loop %llu\n", g[1]-g[0], g[2]-g[1]); }
        hipMemcpy(S, hS.data(), hS.size() * 4, hipMemcpyHostToDevice);
    }
    for (int spin : {1000, 100000}) {
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, clk, spin);
        unsigned long long h[3]; hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost);
        printf("clock probe spin=%d: %llu shader cycles in %llu x10ns -> %.0f MHz\n", spin, h[0], h[1], h[0] / (h[1] * 0.01));
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int nb : {1, 59}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k_loop_probe<0>, dim3(nb), dim3(256), 0, 0, po, 64);
                else hipLaunchKernelGGL(k_loop_probe<1>, dim3(nb), dim3(256), 0, 0, po, 64);
                hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            float cyc; hipMemcpy(&cyc, po, 4, hipMemcpyDeviceToHost);
            printf("loop probe mode %d, %d WGs: %.1f us wall, %.0f cycles for 64 steps (%.0f / step)\n", mode, nb, ms * 1e3, cyc, cyc / 64);
        }
    return 0;
}
