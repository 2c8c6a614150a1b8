// pre3_update.hip -- the dense EKF update of update.m:27-56 on gfx950.
//
//   S = H P H' + R ; K = P H' inv(S) ; x += K (z-h) ; P -= K S K' ; P = 0.5(P+P') ; Jnorm rows/cols 4:7
//
// Formulation used here (mathematically identical, see DESIGN.md "update"):
//   HP   = H P                      ELL gather: H has <= 13 non-zeros per row          (k_ell_HP)
//   S    = HP H' + R                ELL gather                                          (k_ell_G)
//   S    = L L'  and  W = L^-1 [HP | nu]   one blocked right-looking sweep over the stacked matrix
//                                   [S ; HP' ; nu'] with 64-wide panels                 (k_chol_panel / k_chol_trail)
//   x   += W' (L^-1 nu)                                                                 (k_update_x, pre3_geom.hip)
//   P   -= W' W                     n x n x r MFMA contraction, the roofline-graded kernel (k_downdate)
//   P - W'W is computed with the same k-ordered fma chain for (i,j) and (j,i), so it is exactly symmetric
//   and the reference's 0.5*P+0.5*P' is the identity on it.
//
// Layout: P is ld x ld (ld multiple of 128, zero beyond n); W/HP are r_pad x ldw row-major, i.e. one
// length-n vector per measurement row, so both MFMA operands of W'W are read k-major/contiguous.
#include "pre3_internal.h"

namespace pre3 {

// ------------------------------------------------------------------------------------------------
// ELL gathers
// ------------------------------------------------------------------------------------------------
// dst[a][j] = sum_t val[a][t] * P[col[a][t]][j]   (a < r), 0 for padded rows; column `ld` <- nu[a]
template <typename T>
__global__ __launch_bounds__(256) void k_ell_HP(int r, int r_pad, const int32_t *__restrict__ row_col, const T *__restrict__ row_val,
                                                const double *__restrict__ row_nu, const T *__restrict__ P, int ld, T *__restrict__ dst,
                                                int ldw, int with_nu)
{
    int a = blockIdx.y;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldw || a >= r_pad) return;
    T out = (T)0;
    if (a < r) {
        if (j < ld) {
            T s = (T)0;
#pragma unroll
            for (int t = 0; t < ELLW; ++t) {
                T v = row_val[a * ELLW + t];
                int c = row_col[a * ELLW + t];
                s += v * P[(size_t)c * ld + j];
            }
            out = s;
        } else if (j == ld && with_nu) {
            out = (T)row_nu[a];
        }
    }
    dst[(size_t)a * ldw + j] = out;
}

// dst[a][b] = sum_t val[b][t] * HP[a][col[b][t]] + (R ? R[a][b] : add_identity*delta_ab); padding = identity
template <typename T>
__global__ __launch_bounds__(64) void k_ell_G(int r, int r_pad, const int32_t *__restrict__ row_col, const T *__restrict__ row_val,
                                              const T *__restrict__ HP, int ldw, T *__restrict__ dst, int ldg, int add_identity,
                                              const T *__restrict__ Rd)
{
    int a = blockIdx.y;
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= r_pad || a >= r_pad) return;
    T out;
    if (a < r && b < r) {
        T s = (T)0;
#pragma unroll
        for (int t = 0; t < ELLW; ++t) s += row_val[b * ELLW + t] * HP[(size_t)a * ldw + row_col[b * ELLW + t]];
        if (Rd) s += Rd[(size_t)a * r + b];
        else if (add_identity && a == b) s += (T)1;
        out = s;
    } else {
        out = (a == b) ? (T)1 : (T)0;
    }
    dst[(size_t)a * ldg + b] = out;
}

// ------------------------------------------------------------------------------------------------
// Blocked Cholesky of S fused with the forward solve W = L^-1 [HP | nu].
// The stacked matrix M = [S ; HP'] (rows: r_pad rows of S, then the ldw columns of HP as rows) is swept
// right-looking in NB=64 panels.  Panel step J:
//   k_chol_panel : every workgroup factors the (already updated) diagonal block M_JJ in LDS; workgroup b
//                  then solves its own 64-row block X <- M_bJ L_JJ^-T.   (S blocks below J, and all W strips)
//   k_chol_trail : M_bK -= M_bJ M_KJ'  for K > J                         (S lower tiles and all W strips)
// W strip c holds M(i,a) = W[a][c*64+i] (transposed storage), so its loads/stores are coalesced in i.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_chol_panel(T *__restrict__ S, int lds, T *__restrict__ W, int ldw, int J, int nrb,
                                                    int32_t *__restrict__ status)
{
    __shared__ T Ls[NB][NB + 1];
    __shared__ T Xs[NB][NB + 1];      // Xs[a][i]
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nS = nrb - J - 1;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, a = idx % NB;
        Ls[i][a] = S[(size_t)(J * NB + i) * lds + J * NB + a];
    }
    // load X early (independent of the factorisation)
    const bool isW = b > nS;
    const int c0 = isW ? (b - nS - 1) * NB : 0;
    if (b >= 1) {
        if (!isW) {
            int rb = J + b;
            for (int idx = tid; idx < NB * NB; idx += 256) {
                int i = idx / NB, a = idx % NB;
                Xs[a][i] = S[(size_t)(rb * NB + i) * lds + J * NB + a];
            }
        } else {
            for (int idx = tid; idx < NB * NB; idx += 256) {
                int a = idx / NB, i = idx % NB;
                Xs[a][i] = W[(size_t)(J * NB + a) * ldw + c0 + i];
            }
        }
    }
    __syncthreads();
    // --- potrf (lower) with deferred column scaling: after step c, column c holds l_ic * sqrt(piv_c)
    bool bad = false;
    for (int c = 0; c < NB; ++c) {
        T piv = Ls[c][c];
        if (!(piv > (T)0)) { bad = true; piv = (T)1; }
        T inv = (T)1 / piv;
        int mrem = NB - 1 - c;
        for (int idx = tid; idx < mrem * mrem; idx += 256) {
            int i = c + 1 + idx / mrem, j = c + 1 + idx % mrem;
            if (j <= i) Ls[i][j] -= Ls[i][c] * Ls[j][c] * inv;
        }
        __syncthreads();
    }
    if (bad && tid == 0 && b == 0) atomicExch(status, 1);
    // scale: L[i][c] = A[i][c] / sqrt(piv_c), L[c][c] = sqrt(piv_c)
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, c = idx % NB;
        if (i > c) { T p = Ls[c][c]; if (!(p > (T)0)) p = (T)1; Ls[i][c] = Ls[i][c] / sqrt(p); }
    }
    __syncthreads();
    if (tid < NB) { T p = Ls[tid][tid]; if (!(p > (T)0)) p = (T)1; Ls[tid][tid] = sqrt(p); }
    __syncthreads();
    if (b == 0) {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            int i = idx / NB, a = idx % NB;
            S[(size_t)(J * NB + i) * lds + J * NB + a] = (a <= i) ? Ls[i][a] : (T)0;
        }
        return;
    }
    // --- X <- X L^-T, right-looking with deferred scaling: after step c row Xs[c][:] holds x_c * L[c][c]
    for (int c = 0; c < NB; ++c) {
        T invd = (T)1 / Ls[c][c];
        int mrem = NB - 1 - c;
        for (int idx = tid; idx < mrem * NB; idx += 256) {
            int j = c + 1 + idx / NB, i = idx % NB;
            Xs[j][i] -= (Xs[c][i] * invd) * Ls[j][c];
        }
        __syncthreads();
    }
    if (!isW) {
        int rb = J + b;
        for (int idx = tid; idx < NB * NB; idx += 256) {
            int i = idx / NB, a = idx % NB;
            S[(size_t)(rb * NB + i) * lds + J * NB + a] = Xs[a][i] / Ls[a][a];
        }
    } else {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            int a = idx / NB, i = idx % NB;
            W[(size_t)(J * NB + a) * ldw + c0 + i] = Xs[a][i] / Ls[a][a];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_chol_trail(T *__restrict__ S, int lds, T *__restrict__ W, int ldw, int J, int nrb, int nW)
{
    __shared__ T As[NB][NB + 4];   // As[a][i]
    __shared__ T Bs[NB][NB + 4];   // Bs[a][j]
    const int tid = threadIdx.x;
    const int nK = nrb - J - 1;
    const int nSt = nK * (nK + 1) / 2;
    int idx = blockIdx.x;
    bool isW;
    int rb = 0, K, c0 = 0;
    if (idx < nSt) {
        // triangular decode: rows bb = 0..nK-1 (block J+1+bb), cols kk <= bb
        int bb = 0;
        while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
        int kk = idx - bb * (bb + 1) / 2;
        rb = J + 1 + bb; K = J + 1 + kk; isW = false;
    } else {
        int t = idx - nSt;
        c0 = (t % nW) * NB; K = J + 1 + t / nW; isW = true;
    }
    for (int e = tid; e < NB * NB; e += 256) {
        int j = e / NB, a = e % NB;
        Bs[a][j] = S[(size_t)(K * NB + j) * lds + J * NB + a];
    }
    if (!isW) {
        for (int e = tid; e < NB * NB; e += 256) {
            int i = e / NB, a = e % NB;
            As[a][i] = S[(size_t)(rb * NB + i) * lds + J * NB + a];
        }
    } else {
        for (int e = tid; e < NB * NB; e += 256) {
            int a = e / NB, i = e % NB;
            As[a][i] = W[(size_t)(J * NB + a) * ldw + c0 + i];
        }
    }
    __syncthreads();
    const int ti = (tid & 15) * 4, tj = (tid >> 4) * 4;
    T acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = (T)0;
#pragma unroll 8
    for (int a = 0; a < NB; ++a) {
        T av[4], bv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { av[p] = As[a][ti + p]; bv[p] = Bs[a][tj + p]; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] += av[p] * bv[q];
    }
    if (!isW) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(size_t)(rb * NB + ti + p) * lds + K * NB + tj + q] -= acc[p][q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) W[(size_t)(K * NB + tj + q) * ldw + c0 + ti + p] -= acc[p][q];
    }
}

// K' = L^-T W  (so that K = W' L^-1 ... = P H' inv(S)); slow back substitution, one lane per state row.
// Only the stateless drop-in returns K (no caller in the reference uses it).  Kt: r_pad x ldw.
template <typename T>
__global__ __launch_bounds__(256) void k_gain(int n, int r, const T *__restrict__ L, int lds, const T *__restrict__ W, int ldw,
                                              T *__restrict__ Kt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int a = r - 1; a >= 0; --a) {
        T s = W[(size_t)a * ldw + i];
        for (int b = a + 1; b < r; ++b) s -= L[(size_t)b * lds + a] * Kt[(size_t)b * ldw + i];
        Kt[(size_t)a * ldw + i] = s / L[(size_t)a * lds + a];
    }
}

// ------------------------------------------------------------------------------------------------
// K9 covariance down-date  P <- P - W'W  on the matrix cores.
// fp32: v_mfma_f32_32x32x2_f32 (exact f32 fma chain), fp64: v_mfma_f64_16x16x4_f64.
// Workgroup = 4 waves in 2x2, each wave owns a 64x64 sub-tile of a 128x128 tile of P; W is staged
// k-major through LDS (BK rows x 128 contiguous columns per operand), double-buffered.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int BLK = 32, KS = 2, NREG = 16;
    typedef float acc_t __attribute__((ext_vector_type(16)));
    static __device__ inline void mma(float a, float b, acc_t &c) { c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
    static __device__ inline int col(int lane) { return lane & 31; }
    static __device__ inline int kk(int lane) { return lane >> 5; }
};
template <> struct Mfma<double> {
    static constexpr int BLK = 16, KS = 4, NREG = 4;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ inline void mma(double a, double b, acc_t &c) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
    static __device__ inline int col(int lane) { return lane & 15; }
    static __device__ inline int kk(int lane) { return lane >> 4; }
};

template <typename T, int BK>
__global__ __launch_bounds__(256) void k_downdate(T *__restrict__ P, int ld, const T *__restrict__ W, int ldw, int r_pad)
{
    using M = Mfma<T>;
    constexpr int NBLK = 64 / M::BLK;                 // MFMA blocks per wave-tile dimension
    constexpr int VEC = 16 / sizeof(T);               // elements per 16-byte access
    constexpr int ROWV = TILE / VEC;                  // 16-byte vectors per staged row
    constexpr int NLD = (BK * ROWV) / 256;            // 16-byte loads per thread per operand per stage
    static_assert((BK * ROWV) % 256 == 0, "stage must divide over the workgroup");
    typedef T vec_t __attribute__((ext_vector_type(VEC)));

    __shared__ __attribute__((aligned(16))) T sA[2][BK][TILE];
    __shared__ __attribute__((aligned(16))) T sB[2][BK][TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int I0 = blockIdx.y * TILE, J0 = blockIdx.x * TILE;

    typename M::acc_t acc[NBLK][NBLK];
#pragma unroll
    for (int p = 0; p < NBLK; ++p)
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
            for (int e = 0; e < M::NREG; ++e) acc[p][q][e] = (T)0;

    vec_t ra[NLD], rb[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            int v = tid + l * 256;
            int kr = v / ROWV, cv = (v % ROWV) * VEC;
            ra[l] = *reinterpret_cast<const vec_t *>(W + (size_t)(k0 + kr) * ldw + I0 + cv);
            rb[l] = *reinterpret_cast<const vec_t *>(W + (size_t)(k0 + kr) * ldw + J0 + cv);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            int v = tid + l * 256;
            int kr = v / ROWV, cv = (v % ROWV) * VEC;
            *reinterpret_cast<vec_t *>(&sA[buf][kr][cv]) = ra[l];
            *reinterpret_cast<vec_t *>(&sB[buf][kr][cv]) = rb[l];
        }
    };

    const int nstage = r_pad / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) gload((s + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / M::KS; ++ks) {
            const int krow = ks * M::KS + M::kk(lane);
            T av[NBLK], bv[NBLK];
#pragma unroll
            for (int p = 0; p < NBLK; ++p) {
                av[p] = sA[buf][krow][wi * 64 + p * M::BLK + M::col(lane)];
                bv[p] = sB[buf][krow][wj * 64 + p * M::BLK + M::col(lane)];
            }
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int q = 0; q < NBLK; ++q) M::mma(av[p], bv[q], acc[p][q]);
        }
        if (s + 1 < nstage) sstore(buf ^ 1);
        __syncthreads();
    }
    // epilogue: P <- P - acc
#pragma unroll
    for (int p = 0; p < NBLK; ++p)
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
            for (int e = 0; e < M::NREG; ++e) {
                int row = I0 + wi * 64 + p * M::BLK + M::row(lane, e);
                int col = J0 + wj * 64 + q * M::BLK + M::col(lane);
                size_t o = (size_t)row * ld + col;
                P[o] = P[o] - acc[p][q][e];
            }
}

// synthetic W for the roofline probe
template <typename T>
__global__ void k_fill_w(T *W, size_t count, float scale)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t h = (uint32_t)(i * 2654435761u) ^ 0x9e3779b9u;
    h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
    W[i] = (T)(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#define DISPATCH_T(c, expr_f64, expr_f32) do { if ((c)->dtype == PRE3_F64) { expr_f64; } else { expr_f32; } } while (0)

int launch_ell_HP(pre3_ctx *c, int r, void *dst, bool with_nu)
{
    int r_pad = round_up(r, NB);
    dim3 g(ceil_div(c->ldw, 256), r_pad), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_ell_HP<double>, g, b, 0, c->stream, r, r_pad, c->row_col, (const double *)c->row_val, c->row_nu,
                           (const double *)c->P, c->ld, (double *)dst, c->ldw, with_nu ? 1 : 0),
        hipLaunchKernelGGL(k_ell_HP<float>, g, b, 0, c->stream, r, r_pad, c->row_col, (const float *)c->row_val, c->row_nu,
                           (const float *)c->P, c->ld, (float *)dst, c->ldw, with_nu ? 1 : 0));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_ell_G(pre3_ctx *c, int r, const void *HPsrc, void *dst, int ldg, int add_identity, const void *Rdense)
{
    int r_pad = round_up(r, NB);
    dim3 g(ceil_div(r_pad, 64), r_pad), b(64);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_ell_G<double>, g, b, 0, c->stream, r, r_pad, c->row_col, (const double *)c->row_val, (const double *)HPsrc,
                           c->ldw, (double *)dst, ldg, add_identity, (const double *)Rdense),
        hipLaunchKernelGGL(k_ell_G<float>, g, b, 0, c->stream, r, r_pad, c->row_col, (const float *)c->row_val, (const float *)HPsrc,
                           c->ldw, (float *)dst, ldg, add_identity, (const float *)Rdense));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

static int launch_chol_solve(pre3_ctx *c, int r_pad)
{
    int nrb = r_pad / NB, nW = c->ldw / NB;
    for (int J = 0; J < nrb; ++J) {
        int nS = nrb - J - 1;
        dim3 gA(1 + nS + nW), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_chol_panel<double>, gA, b, 0, c->stream, (double *)c->Smat, r_pad, (double *)c->W, c->ldw, J, nrb, c->stats + 6),
            hipLaunchKernelGGL(k_chol_panel<float>, gA, b, 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W, c->ldw, J, nrb, c->stats + 6));
        if (nS > 0) {
            dim3 gB(nS * (nS + 1) / 2 + nS * nW);
            DISPATCH_T(c,
                hipLaunchKernelGGL(k_chol_trail<double>, gB, b, 0, c->stream, (double *)c->Smat, r_pad, (double *)c->W, c->ldw, J, nrb, nW),
                hipLaunchKernelGGL(k_chol_trail<float>, gB, b, 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W, c->ldw, J, nrb, nW));
        }
    }
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_downdate(pre3_ctx *c, int r, const void *W)
{
    int r_pad = round_up(r, NB);
    dim3 g(c->ld / TILE, c->ld / TILE), b(256);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->kt.enabled) {
        if ((size_t)c->kt.used + 2 > c->kt.ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; PRE3_HIP(hipEventCreate(&e)); c->kt.ev.push_back(e); }
        }
        e0 = c->kt.ev[c->kt.used]; e1 = c->kt.ev[c->kt.used + 1];
        c->kt.used += 2;
        PRE3_HIP(hipEventRecord(e0, c->stream));
    }
    DISPATCH_T(c,
        hipLaunchKernelGGL((k_downdate<double, 8>), g, b, 0, c->stream, (double *)c->P, c->ld, (const double *)W, c->ldw, r_pad),
        hipLaunchKernelGGL((k_downdate<float, 16>), g, b, 0, c->stream, (float *)c->P, c->ld, (const float *)W, c->ldw, r_pad));
    if (c->kt.enabled) {
        PRE3_HIP(hipEventRecord(e1, c->stream));
        c->kt.flops += 2.0 * c->n * (double)c->n * r;                    // F_K9 = 2 n^2 r
        c->kt.bytes += 2.0 * c->n * (double)c->n * c->esz + 2.0 * c->n * (double)r * c->esz;
    }
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_fill_w(pre3_ctx *c, int r_pad)
{
    size_t count = (size_t)r_pad * c->ldw;
    dim3 g((unsigned)((count + 255) / 256)), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_fill_w<double>, g, b, 0, c->stream, (double *)c->W, count, 1e-3f),
        hipLaunchKernelGGL(k_fill_w<float>, g, b, 0, c->stream, (float *)c->W, count, 1e-3f));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_update_x(pre3_ctx *c, int which_prior, int r);   // pre3_geom.hip

// rows already in c->row_* (r rows).  which_prior selects x prior; P currently holds the prior covariance.
int run_update(pre3_ctx *c, int which_prior, int r, bool dense_R, void *Kt_out_dev)
{
    if (r == 0) {   // update.m:50-55: x_k_k = x_km1_k, p_k_k = p_km1_k
        if (which_prior == PRE3_X_K_KM1) PRE3_HIP(hipMemcpyAsync(c->x_kk, c->x_km1, sizeof(double) * c->n, hipMemcpyDeviceToDevice, c->stream));
        return PRE3_OK;
    }
    int r_pad = round_up(r, NB);
    PRE3_CHECK(r_pad <= c->rcap, PRE3_E_ARG, "update with %d rows exceeds the context capacity %d", r, c->rcap);
    PRE3_TRY(launch_ell_HP(c, r, c->W, true));
    PRE3_TRY(launch_ell_G(c, r, c->W, c->Smat, r_pad, 1, dense_R ? c->Rdense : nullptr));
    PRE3_TRY(launch_chol_solve(c, r_pad));
    PRE3_TRY(launch_update_x(c, which_prior, r));
    PRE3_TRY(launch_downdate(c, r, c->W));
    PRE3_TRY(launch_jnorm(c, 0));
    if (Kt_out_dev) {
        dim3 g(ceil_div(c->n, 256)), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_gain<double>, g, b, 0, c->stream, c->n, r, (const double *)c->Smat, r_pad, (const double *)c->W, c->ldw, (double *)Kt_out_dev),
            hipLaunchKernelGGL(k_gain<float>, g, b, 0, c->stream, c->n, r, (const float *)c->Smat, r_pad, (const float *)c->W, c->ldw, (float *)Kt_out_dev));
        PRE3_HIP(hipGetLastError());
    }
    return PRE3_OK;
}

}  // namespace pre3
