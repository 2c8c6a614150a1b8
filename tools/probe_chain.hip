// micro-probe of the factor wave's dependent chain (one wave, no LDS): what do the chain's instructions cost back to back?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_chain.hip -o tools/probe_chain && tools/probe_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline float rdl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// MODE 0: dependent v_fma chain; 1: dependent (v_readlane -> v_fma); 2: dependent v_rcp; 3: dependent v_rsq; 4: rdlane->rcp->mul->fma (one column of the chain)
template <int MODE>
__global__ void k_lat(float *out, unsigned long long *clk, int n)
{
    float x = 1.0f + 0.001f * threadIdx.x, y = 0.999f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) x = __builtin_fmaf(x, y, 0.001f);
            if (MODE == 1) x = __builtin_fmaf(x, rdl(x, (u * 5) & 63), 0.5f);
            if (MODE == 2) x = __builtin_amdgcn_rcpf(x) + 0.5f;
            if (MODE == 3) x = __builtin_amdgcn_rsqf(x) + 0.5f;
            if (MODE == 4) { float p = rdl(x, u & 63); float r = __builtin_amdgcn_rcpf(p); float w = r * 0.5f; x = __builtin_fmaf(-x, w, x + 1.0f); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}

// the 8-column sub-panel factor step as in chol_panel_body (rcp form), REP times on register data
template <int FORM>
__global__ void k_sub(float *out, unsigned long long *clk, int n)
{
    const int i = threadIdx.x & 63;
    float a[8], y[8];
    for (int t = 0; t < 8; ++t) a[t] = (i == t ? 50.f : 0.f) + 0.01f * ((i * 7 + t * 13) % 17);
    float accum = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < n; ++it) {
        float b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = a[t] + 1e-3f * accum;
        if (FORM == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float piv = rdl(b[c], c);
                float st[8];
#pragma unroll
                for (int t = c + 1; t < 8; ++t) st[t] = rdl(b[c], t);
                const float rinv = __builtin_amdgcn_rcpf(piv), rs = __builtin_amdgcn_rsqf(piv);
#pragma unroll
                for (int t = c + 1; t < 8; ++t) b[t] -= b[c] * (st[t] * rinv);
                y[c] = b[c] * rs;
            }
        } else if (FORM == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float piv = rdl(b[c], c);
                if (!(piv > 0.f)) piv = 1.f;
                const float rs = __builtin_amdgcn_rsqf(piv);
                y[c] = b[c] * rs;
#pragma unroll
                for (int t = c + 1; t < 8; ++t) b[t] -= y[c] * rdl(y[c], t);
            }
        } else {
            // FORM 2: the 8x8 diagonal block factored on uniform values first (36 readlanes up front, then a uniform chain), rows solved afterwards
            float d[8][8], rs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int w = 0; w <= u; ++w) d[u][w] = rdl(b[w], u);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                rs[c] = __builtin_amdgcn_rsqf(d[c][c]);
#pragma unroll
                for (int u = c + 1; u < 8; ++u) d[u][c] *= rs[c];
#pragma unroll
                for (int u = c + 1; u < 8; ++u)
#pragma unroll
                    for (int w = c + 1; w <= u; ++w) d[u][w] -= d[u][c] * d[w][c];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float acc = b[t];
#pragma unroll
                for (int u = 0; u < t; ++u) acc -= y[u] * d[t][u];
                y[t] = acc * rs[t];
            }
        }
        accum = y[7] + y[3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = accum;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}

int main()
{
    float *o; unsigned long long *clk, h;
    hipMalloc(&o, 4096); hipMalloc(&clk, 64);
    const int n = 64;
#define RUN(K, label, per) do { for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, o, clk, n); hipDeviceSynchronize(); } \
        hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); printf("%-60s %8.1f cycles per %s\n", label, (double)h / (n * (per == 0 ? 16 : 1)), per == 0 ? "op" : "sub-panel"); } while (0)
    RUN(k_lat<0>, "dependent v_fma_f32", 0);
    RUN(k_lat<1>, "dependent v_readlane -> v_fma", 0);
    RUN(k_lat<2>, "dependent v_rcp_f32 + v_add", 0);
    RUN(k_lat<3>, "dependent v_rsq_f32 + v_add", 0);
    RUN(k_lat<4>, "readlane -> rcp -> mul -> fma(+add)", 0);
    RUN(k_sub<0>, "8-col sub-panel, rcp form", 1);
    RUN(k_sub<1>, "8-col sub-panel, rsq + readlane form", 1);
    RUN(k_sub<2>, "8-col sub-panel, uniform 8x8 block first", 1);
    return 0;
}
