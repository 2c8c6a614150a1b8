// micro-probe 2: the factor wave's whole step (LDS reads, lookahead, 8 columns, LDS writes, barrier) alone and beside worker-like waves
//   hipcc --offload-arch=gfx950 -O3 tools/probe_chain2.hip -o tools/probe_chain2 && tools/probe_chain2
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline float rdl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
typedef float v4_t __attribute__((ext_vector_type(4)));

// WORK: what the other waves do per step: 0 idle (barrier only), 1 LDS reads + FMAs like a worker wave, 2 FMAs only, 3 LDS reads only
// PARTS bitmask for wave 0: 1 Pn read, 2 lookahead (16 b128 + 64 fma), 4 columns, 8 writes
template <int WORK, int PARTS, int BARRIER, int CW = 0, int UNROLL = 0>
__global__ __launch_bounds__(384) void k_step(float *out, unsigned long long *clk, int n)
{
    __shared__ __attribute__((aligned(16))) float Ls[64][68];
    __shared__ __attribute__((aligned(16))) float Pn[2][64][8];
    const int tid = threadIdx.x, role = (tid >> 6) == CW ? 0 : 1, i = tid & 63;
    unsigned long long own = 0;
    for (int e = tid; e < 64 * 68; e += blockDim.x) Ls[e / 68][e % 68] = 0.01f * ((e * 7) % 13);
    for (int e = tid; e < 2 * 64 * 8; e += blockDim.x) (&Pn[0][0][0])[e] = ((e >> 3) & 63) == (e & 7) ? 50.f : 0.02f * (e % 11);
    __syncthreads();
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    float yprev[8] = { 0.1f, 0.2f, 0.1f, 0.3f, 0.2f, 0.1f, 0.2f, 0.1f };
    float wv[16];
    for (int e = 0; e < 16; ++e) wv[e] = 1.f + e;
    float accum = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll UNROLL + 1
    for (int it = 0; it < (UNROLL ? UNROLL + 1 : n); ++it) {
        const int C = 8 * (1 + (it & 3)), par = it & 1;
        if (role == 0) {
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
            float a[8], y[8];
            v4_t v0 = { 50.f, 0.1f, 0.2f, 0.3f }, v1 = { 0.1f, 0.2f, 0.3f, 0.1f };
            if (PARTS & 1) { v0 = *reinterpret_cast<const v4_t *>(&Pn[par][i][0]); v1 = *reinterpret_cast<const v4_t *>(&Pn[par][i][4]); }
            v4_t Lh[8][2];
            if (PARTS & 2) {
#pragma unroll
                for (int t = 0; t < 8; ++t) { Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - 8]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - 4]); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t] = v0[t] + 1e-6f * accum; a[4 + t] = v1[t]; }
            if (PARTS & 2) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    float s0 = yprev[0] * Lh[t][0][0], s1 = yprev[4] * Lh[t][1][0];
#pragma unroll
                    for (int u = 1; u < 4; ++u) { s0 += yprev[u] * Lh[t][0][u]; s1 += yprev[4 + u] * Lh[t][1][u]; }
                    a[t] -= 1e-3f * (s0 + s1);
                }
            }
            if (PARTS & 4) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float piv = rdl(a[c], c);
                    float st[8];
#pragma unroll
                    for (int t = c + 1; t < 8; ++t) st[t] = rdl(a[c], t);
                    const float rinv = __builtin_amdgcn_rcpf(piv), rs = __builtin_amdgcn_rsqf(piv);
#pragma unroll
                    for (int t = c + 1; t < 8; ++t) a[t] -= a[c] * (st[t] * rinv);
                    y[c] = a[c] * rs;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) y[c] = a[c];
            }
            if (PARTS & 8) {
                *reinterpret_cast<v4_t *>(&Ls[i][C]) = v4_t{ y[0], y[1], y[2], y[3] };
                *reinterpret_cast<v4_t *>(&Ls[i][C + 4]) = v4_t{ y[4], y[5], y[6], y[7] };
            }
            accum = y[7] + y[2];
            own += __builtin_amdgcn_s_memtime() - w0;
        } else if (WORK != 0) {
            const int tr = (tid >> 4) & 15, tc = tid & 15;
            v4_t Yi[3][2], Yj[3][2];
            if (WORK == 1 || WORK == 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    Yi[p][0] = *reinterpret_cast<const v4_t *>(&Ls[tr + 16 * p][C - 8]); Yi[p][1] = *reinterpret_cast<const v4_t *>(&Ls[tr + 16 * p][C - 4]);
                    Yj[p][0] = *reinterpret_cast<const v4_t *>(&Ls[tc + 16 * p][C - 8]); Yj[p][1] = *reinterpret_cast<const v4_t *>(&Ls[tc + 16 * p][C - 4]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) { Yi[p][0] = v4_t{ wv[0], wv[1], wv[2], wv[3] }; Yi[p][1] = Yi[p][0]; Yj[p][0] = Yi[p][0]; Yj[p][1] = Yi[p][0]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WORK == 1 || WORK == 2) {
#pragma unroll
                for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            float s0 = Yi[p][0][0] * Yj[q][0][0], s1 = Yi[p][1][0] * Yj[q][1][0];
#pragma unroll
                            for (int t = 1; t < 4; ++t) { s0 += Yi[p][0][t] * Yj[q][0][t]; s1 += Yi[p][1][t] * Yj[q][1][t]; }
                            wv[(p * 3 + q + rep * 7) & 15] -= 1e-6f * (s0 + s1);
                        }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) wv[p] += Yi[p][0][0] + Yi[p][1][1] + Yj[p][0][2] + Yj[p][1][3];
            }
            if ((tc >> 3) == (it & 1)) {
#pragma unroll
                for (int p = 0; p < 4; ++p) Pn[par ^ 1][tr + 16 * p][tc & 7] = ((tr + 16 * p) == (tc & 7) ? 50.f : 0.f) + 1e-9f * wv[p];
            }
        }
        if (BARRIER) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = accum;
    for (int e = 0; e < 16; ++e) s += wv[e];
    out[tid] = s;
    if (tid == CW * 64) { clk[0] = t1 - t0; clk[1] = own; }
}

int main()
{
    float *o; unsigned long long *clk, h[2];
    (void)hipMalloc(&o, 4096); (void)hipMalloc(&clk, 64);
    const int n = 64;
#define RUN(K, threads, label) do { for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(1), dim3(threads), 0, 0, o, clk, n); (void)hipDeviceSynchronize(); } \
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); printf("%-86s %8.1f cycles per step, factor wave busy %8.1f\n", label, (double)h[0] / n, (double)h[1] / n); } while (0)
    RUN((k_step<0, 4, 0>), 64, "1 wave: columns only, no barrier");
    RUN((k_step<0, 5, 0>), 64, "1 wave: Pn read + columns");
    RUN((k_step<0, 13, 0>), 64, "1 wave: Pn read + columns + writes");
    RUN((k_step<0, 15, 0>), 64, "1 wave: whole step (Pn, lookahead, columns, writes), no barrier");
    RUN((k_step<0, 15, 1>), 64, "1 wave: whole step + barrier");
    RUN((k_step<0, 15, 1>), 384, "6 waves, others idle at the barrier");
    RUN((k_step<2, 15, 1>), 384, "6 waves, others FMA only");
    RUN((k_step<3, 15, 1>), 384, "6 waves, others LDS reads only");
    RUN((k_step<1, 15, 1>), 384, "6 waves, others LDS reads + FMAs (worker-like)");
    RUN((k_step<1, 15, 1>), 128, "2 waves (other SIMD), other worker-like");
    RUN((k_step<1, 4, 1>), 384, "6 waves worker-like, wave 0 columns only");
    RUN((k_step<1, 6, 1>), 384, "6 waves worker-like, wave 0 lookahead + columns");
    RUN((k_step<1, 15, 1, 2>), 384, "6 waves worker-like, factor wave = wave 2 (alone on its SIMD)");
    RUN((k_step<1, 15, 1, 0>), 320, "5 waves worker-like, factor wave = wave 0");
    RUN((k_step<3, 15, 1, 2>), 384, "6 waves LDS reads only, factor wave = wave 2");
    RUN((k_step<2, 15, 1, 2>), 384, "6 waves FMA only, factor wave = wave 2");
    RUN((k_step<1, 15, 1, 2, 63>), 384, "6 waves worker-like, factor = wave 2, 64 steps fully UNROLLED (cold code)");
    RUN((k_step<0, 15, 1, 2, 63>), 384, "6 waves others idle, factor = wave 2, 64 steps fully UNROLLED (cold code)");
    return 0;
}
