// micro-probe 3 (round 6): the whole chain of one 64-column panel (chol_chain, pre3_chain.h) against the flag-driven form
// (chol_chain_async, pre3_chain_async.h) on ONE workgroup: shader clocks per chain, and both results against a double-precision factorisation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 tools/probe_chain3.hip -o tools/probe_chain3 && tools/probe_chain3
#ifdef STAMPS
#define PRE3_PROBE_CHA 1
#endif
#include "../3pre_amd/csrc/pre3_chain_async.h"
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
using namespace pre3;

template <typename T, int FORM, bool XTRI = false>
__global__ __launch_bounds__(768) void k_chain(const T *A, T *Lout, T *Mout, unsigned long long *clk, int nrep, int *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ChSmem<T> &sm = *reinterpret_cast<ChSmem<T> *>(smem);
    const int tid = threadIdx.x;
    unsigned long long total = 0;
    bool bad = false;
    for (int rep = 0; rep < nrep; ++rep) {
        for (int idx = tid; idx < NB * NB; idx += blockDim.x) { const int i = idx >> 6, c = idx & 63; sm.Ls[i][c] = A[idx]; sm.Xs[i][c] = i == c ? (T)1 : (T)0; }
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        typename ChW<T>::acc_t acc[ChW<T>::NBLK][ChW<T>::NBLK];
        if (tid < 640 || FORM == 1) {
            if constexpr (FORM == 0) chol_chain<T, false, false>(sm, acc, false, true, bad, [](int) {});
            else chol_chain_async<T, false, XTRI>(sm, acc, false, true, bad);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        total += t1 - t0;
        if (FORM == 0) __syncthreads();
    }
    for (int idx = tid; idx < NB * NB; idx += blockDim.x) { const int i = idx >> 6, c = idx & 63; Lout[idx] = c <= i ? sm.Ls[i][c] : (T)0; Mout[idx] = sm.Xs[i][c]; }
    if (tid == 0) clk[0] = total;
    if (bad) atomicExch(status, 1);
}

template <typename T, int FORM, bool XTRI = false>
static void run(const char *label, int threads)
{
    std::vector<double> A(NB * NB), L(NB * NB, 0.0);
    srand(7);
    std::vector<double> G(NB * NB);
    for (auto &g : G) g = (rand() % 2001 - 1000) * 1e-3;
    for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) { double s = i == j ? 8.0 : 0.0; for (int k = 0; k < NB; ++k) s += G[i * NB + k] * G[j * NB + k] * 0.25; A[i * NB + j] = s; }
    for (int j = 0; j < NB; ++j) {
        double d = A[j * NB + j];
        for (int k = 0; k < j; ++k) d -= L[j * NB + k] * L[j * NB + k];
        L[j * NB + j] = std::sqrt(d);
        for (int i = j + 1; i < NB; ++i) { double s = A[i * NB + j]; for (int k = 0; k < j; ++k) s -= L[i * NB + k] * L[j * NB + k]; L[i * NB + j] = s / L[j * NB + j]; }
    }
    std::vector<T> hA(NB * NB), hL(NB * NB), hM(NB * NB);
    for (int i = 0; i < NB * NB; ++i) hA[i] = (T)A[i];
    T *dA, *dL, *dM; unsigned long long *clk; int *st;
    (void)hipMalloc(&dA, sizeof(T) * NB * NB); (void)hipMalloc(&dL, sizeof(T) * NB * NB); (void)hipMalloc(&dM, sizeof(T) * NB * NB); (void)hipMalloc(&clk, 64); (void)hipMalloc(&st, 4);
    (void)hipMemset(st, 0, 4);
    (void)hipMemcpy(dA, hA.data(), sizeof(T) * NB * NB, hipMemcpyHostToDevice);
    const int nrep = 50;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain<T, FORM, XTRI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChSmem<T>));
    unsigned long long h = 0;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k_chain<T, FORM, XTRI>), dim3(1), dim3(threads), sizeof(ChSmem<T>), 0, dA, dL, dM, clk, nrep, st); (void)hipDeviceSynchronize(); }
    int hs = 0;
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hs, st, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hL.data(), dL, sizeof(T) * NB * NB, hipMemcpyDeviceToHost); (void)hipMemcpy(hM.data(), dM, sizeof(T) * NB * NB, hipMemcpyDeviceToHost);
    double eL = 0, eM = 0;
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) eL = std::max(eL, std::fabs((double)hL[i * NB + j] - L[i * NB + j]));
    // M L = I ?
    for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) { double s = 0; for (int k = 0; k < NB; ++k) s += (double)hM[i * NB + k] * L[k * NB + j]; eM = std::max(eM, std::fabs(s - (i == j ? 1.0 : 0.0))); }
#ifdef STAMPS
    if (FORM == 1 && sizeof(T) == 4 && threads == 768) {
        unsigned long long g[8 * 12 * 4];
        (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(pre3::g_cha), sizeof g);
        const unsigned long long t0 = g[(0 * 12 + 1) * 4 + 0];
        const char *names[8] = { "F", "z", "D0", "D2", "D3", "X0", "X1", "X2" };
        for (int r = 0; r < 8; ++r) {
            printf("  %-3s", names[r]);
            for (int k = -1; k <= 8; ++k) { printf(" |k=%d", k); for (int sl = 0; sl < 4; ++sl) { const unsigned long long v = g[(r * 12 + k + 1) * 4 + sl]; if (v > t0 && v - t0 < 100000) printf(" %5llu", v - t0); else printf("     -"); } }
            printf("\n");
        }
    }
#endif
    printf("%-44s %9.1f cycles per chain   max|L - L64| %.3e   max|M L64 - I| %.3e   status %d   hip: %s\n", label, (double)h / nrep, eL, eM, hs, hipGetErrorString(hipGetLastError()));
    (void)hipFree(dA); (void)hipFree(dL); (void)hipFree(dM); (void)hipFree(clk); (void)hipFree(st);
}

int main()
{
    run<float, 0>("f32 lock-step chain (10 barriers), 640 threads", 640);
    run<float, 1>("f32 flag-driven chain, 640 threads", 640);
    run<float, 1>("f32 flag-driven chain, 768 threads", 768);
    run<float, 1, true>("f32 flag-driven chain, X = I (XTRI), 768 thr", 768);
    run<double, 0>("f64 lock-step chain, 640 threads", 640);
    run<double, 1>("f64 flag-driven chain, 640 threads", 640);
    return 0;
}
