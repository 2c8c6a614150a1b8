// probe: what does project_core (projection + Jacobian of one landmark, one lane, fp64) cost, and which of its library calls?  cycles of one lane alone on a CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../3pre_amd/csrc/pre3_geomdev.h"
using namespace pre3;
__global__ void k(const double *x, const double *y, CamD cam, double *out, unsigned long long *cyc)
{
    double xp[7], yl[6];
    for (int i = 0; i < 7; ++i) xp[i] = x[i];
    for (int i = 0; i < 6; ++i) yl[i] = y[i];
    double Hc[14], Hl[12], zi[2], h_old[2] = { 0, 0 };
    bool fresh;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    project_core(PRE3_INVDEPTH, xp, yl, cam, 0, h_old, zi, fresh, Hc, Hl);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s1, c1, s2, c2;
    sincos(yl[3] + zi[0] * 1e-9, &s1, &c1); sincos(yl[4] + zi[1] * 1e-9, &s2, &c2);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    double a1 = atan2(s1 + 0.3, c1 + 2.0), a2 = atan2(s2 - 0.2, c2 + 2.0);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    double d = (a1 + 1.0) / (a2 + 3.0), e = sqrt(d + 2.0);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    out[0] = zi[0] + Hc[3] + Hl[5] + a1 + a2 + e;
    cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3;
}
int main()
{
    double hx[7] = { 0.1, -0.05, 0.02, 0.999, 0.01, -0.02, 0.015 }, hy[6] = { 0.3, 0.1, -0.1, 0.2, -0.1, 0.4 }, *dx, *dy, *dout;
    unsigned long long *dc, hc[4];
    hipMalloc(&dx, sizeof hx); hipMalloc(&dy, sizeof hy); hipMalloc(&dout, 8); hipMalloc(&dc, 32);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
    CamD cam{ 250.57731, 90, 70, -0.84656, 0.53701, 144, 176 };
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, dx, dy, cam, dout, dc);
        hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost);
        printf("project_core %llu cycles (100 MHz ticks x? : s_memtime counts shader clocks) | 2 sincos %llu | 2 atan2 %llu | div + sqrt %llu\n", hc[0], hc[1], hc[2], hc[3]);
    }
    return 0;
}
