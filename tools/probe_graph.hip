// probe: host cost and GPU span of 11 small kernels + one 16 KB H2D copy, launched directly vs as a captured hipGraph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_small(float *p, int n, int spin) { int i = blockIdx.x * blockDim.x + threadIdx.x; float v = p[i % n]; for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f; p[i % n] = v; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    float *d; CK(hipMalloc(&d, 1 << 20)); void *h; CK(hipHostMalloc(&h, 16384, 0)); void *dd; CK(hipMalloc(&dd, 16384));
    auto enqueue = [&] {
        (void)hipMemcpyAsync(dd, h, 16384, hipMemcpyHostToDevice, st);
        for (int k = 0; k < 11; ++k) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, 1 << 18, 300);
    };
    for (int w = 0; w < 20; ++w) enqueue();
    CK(hipStreamSynchronize(st));
    const int R = 200;
    double th = 0, tt = 0;
    for (int r = 0; r < R; ++r) { double a = now(); enqueue(); double b = now(); (void)hipStreamSynchronize(st); double c = now(); th += b - a; tt += c - a; }
    printf("direct : host enqueue %.1f us, enqueue->sync %.1f us\n", th / R, tt / R);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); enqueue(); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    th = tt = 0;
    for (int r = 0; r < R; ++r) { double a = now(); (void)hipGraphLaunch(ge, st); double b = now(); (void)hipStreamSynchronize(st); double c = now(); th += b - a; tt += c - a; }
    printf("graph  : host launch %.1f us, launch->sync %.1f us\n", th / R, tt / R);
    // single kernel span for reference
    th = tt = 0;
    for (int r = 0; r < R; ++r) { double a = now(); hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, 1 << 18, 300); double b = now(); (void)hipStreamSynchronize(st); double c = now(); th += b - a; tt += c - a; }
    printf("1 kern : host %.1f us, launch->sync %.1f us\n", th / R, tt / R);
    return 0;
}
