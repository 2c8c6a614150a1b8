// probe: does a 16-byte raw buffer load at a dword-aligned (not 16-byte-aligned) offset return the bytes at that offset?  (and global_load_dwordx4)
// (found on the way: __builtin_bit_cast(float, v[j]) / (…, v.x) on ONE element of an ext-vector reads element 0 for every j with this compiler --
//  the whole vector has to be re-typed first; the first version of this probe printed "16 16 16 16")
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
__global__ void k(const float *src, float *out)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 0x7fffffff, 0x00020000);
    for (int sh = 0; sh < 4; ++sh) {
        f4_t v = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(sh * 4 + threadIdx.x * 64), 0, 0));
        f4_t w = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(sh * 4 + threadIdx.x * 64), 0, 16));
        f4u_t g = *reinterpret_cast<const f4u_t *>(src + sh + threadIdx.x * 16);
        if (threadIdx.x == 1) for (int j = 0; j < 4; ++j) { out[sh * 12 + j] = v[j]; out[sh * 12 + 4 + j] = w[j]; out[sh * 12 + 8 + j] = g[j]; }
    }
}
int main()
{
    float h[64], *d, *o, ho[48];
    for (int i = 0; i < 64; ++i) h[i] = i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int sh = 0; sh < 4; ++sh) { printf("shift %d: buffer", sh); for (int j = 0; j < 4; ++j) printf(" %g", ho[sh * 12 + j]); printf(" | buffer sc1"); for (int j = 0; j < 4; ++j) printf(" %g", ho[sh * 12 + 4 + j]); printf(" | global"); for (int j = 0; j < 4; ++j) printf(" %g", ho[sh * 12 + 8 + j]); printf("   (expect %d..)\n", 16 + sh); }
    return 0;
}
