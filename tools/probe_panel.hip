// micro-probe: k_chol_step (panel J = 0) in isolation with 1, 2, 8 and 59 workgroups: per-step stamps of the chain waves
//   hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=200000 tools/probe_panel.hip -o tools/probe_panel && tools/probe_panel
#define PRE3_PROBE 1
// -DPRE3_PROBE_STEPS adds per-step stamps (each costs 150-400 cycles: use the phase totals of the plain build for absolute numbers)
#include "../3pre_amd/csrc/pre3_update.hip"
#include <vector>
#include <cstdlib>
namespace pre3 { void set_error(const char*, ...) {} int launch_update_x(pre3_ctx*, int, int) { return 0; } int launch_jnorm(pre3_ctx*, int) { return 0; }
                 ProjRide make_proj_ride(pre3_ctx *, int, int, int, int) { return ProjRide{}; } }
using namespace pre3;

int main()
{
    const int r_pad = 640, ldw = 3072 + 64, nrb = r_pad / 64, nW = ldw / 64;
    float *S, *W; int *status; unsigned int *arrive;
    (void)hipMalloc(&S, sizeof(float) * r_pad * r_pad); (void)hipMalloc(&W, sizeof(float) * r_pad * ldw); (void)hipMalloc(&status, 4); (void)hipMalloc(&arrive, 64);
    (void)hipMemset(arrive, 0, 64);
    std::vector<float> hS((size_t)r_pad * r_pad), hW((size_t)r_pad * ldw);
    for (int i = 0; i < r_pad; ++i) for (int j = 0; j < r_pad; ++j) hS[(size_t)i * r_pad + j] = (i == j ? 50.f : 0.f) + 0.01f * ((i * 7 + j * 13) % 17 + (j * 7 + i * 13) % 17);
    for (auto &w : hW) w = (rand() % 1000) * 1e-3f;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    unsigned int target = 0;
    for (int pb : { 5, 0, 30 })
    for (int nwg : { 59, 59 }) {
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pre3::g_probe_block), &pb, sizeof pb);
        (void)hipMemcpy(S, hS.data(), hS.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
        target += (unsigned)(nwg - 1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chol_step<float>, dim3(nwg), dim3(CH_NTH), 0, 0, S, r_pad, W, ldw, 0, nrb, nW, nwg, status, arrive, target, nwg, nullptr, 0, 0, 0, nullptr, 0);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long g[64 * 8 * 4], pr[16];
        (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(pre3::g_k9), sizeof g); (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(pre3::g_probe), sizeof pr);
        printf("k_chol_step J=0, %2d workgroups: %.1f us", nwg, ms * 1e3);
        if (nwg > 5) {
            printf("  [wg %d] load %llu pro %llu chain %llu store %llu;  steps (factor|z|worker0 busy):", pb, pr[4] - pr[0], pr[1] - pr[4], pr[2] - pr[1], pr[3] - pr[2]);
            for (int k = 0; k < 10; ++k) printf(" %llu|%llu|%llu", g[k * 8 + 1] - g[k * 8], g[k * 8 + 3] - g[k * 8 + 2], g[k * 8 + 5] - g[k * 8 + 4]);
        }
        printf("\n");
    }
    return 0;
}
