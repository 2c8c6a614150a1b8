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
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "pre3_internal.h"
#include "pre3_geomdev.h"
#include "pre3_chain.h"
#include "pre3_chain_async.h"
#ifndef PRE3_CHAIN_ASYNC_F64
#define PRE3_CHAIN_ASYNC_F64 0      // the same for fp64: works, but the compiler gathers the lookahead's 64 broadcast reads in front of the factor wave's arithmetic whatever fences sit between them (254 spilled registers) -- fp64 keeps the lock-step chain
#endif
#ifndef PRE3_CHAIN_ASYNC
#define PRE3_CHAIN_ASYNC 1          // fp32 panel chains outside the persistent kernel (k_chol_step, k_hi_fused) on the flag-driven form; 0: the lock-step chain of rounds 2-5
#endif
#include "pre3_cholp.h"

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

// The same for ALL measured rows, with the ELL rows built on the fly from the per-landmark Jacobians (k_build_rows' work:
// row 2s+c of measurement s = [Hc(c,:) | Hl(c,:)] at columns [0..6 | off..off+d-1], nu = z - h); block column 0 also
// stores the rows for the kernels that follow (H*P*H', scoring, the update).  One kernel boundary less per step.
template <typename T>
__global__ __launch_bounds__(256) void k_ell_HP_build(int m, int r_pad, const int32_t *__restrict__ meas, const int32_t *__restrict__ lm_type,
                                                      const int32_t *__restrict__ lm_off, const double *__restrict__ Hc,
                                                      const double *__restrict__ Hl, const double *__restrict__ z, const double *__restrict__ h,
                                                      int32_t *__restrict__ row_col, T *__restrict__ row_val, double *__restrict__ row_nu,
                                                      const T *__restrict__ P, int ld, T *__restrict__ dst, int ldw, const int32_t *__restrict__ need, int need_tag,
                                                      int ny_build, InnovRide ir, const int32_t *__restrict__ sel = nullptr, InboxRide ib = InboxRide{})
{
    // the last block row (ib.n16 > 0): the hypothesis table's pull from the pinned inbox rides here -- the scorer behind this launch is its first reader
    // (pre3_step_predicted / pre3_ransac with a caller's table: it was a 5-us launch of its own in front of this one)
    if (ib.n16 > 0 && blockIdx.y == gridDim.y - 1) { if (blockIdx.x == 0) inbox_pull_block(ib); return; }
    // rows of blocks beyond the build's own: the S_i pass of search_IC_matches.m:33-44 rides here (pre3_step; it only needs what the
    // prediction's launch left behind, like this kernel)
    if ((int)blockIdx.y >= ny_build) { innov_ride_block<T>(ir, ((int)blockIdx.y - ny_build) * gridDim.x + blockIdx.x); return; }
    // need != nullptr (a rank's slice of a sharded RANSAC round): only the measurements that slice's hypotheses draw (need[s] == tag of
    // this round: no clearing between rounds) are multiplied out
    if (need != nullptr && (int)blockIdx.y < m && need[blockIdx.y] != need_tag) return;
    // blockIdx.y = measurement s (rows 2s and 2s+1 share their 13 P rows: loaded once); four consecutive columns per lane:
    // 16-byte (fp32) loads of the gathered P rows, 16-byte stores
    typedef T v4_t __attribute__((ext_vector_type(4)));
    const int sIdx = blockIdx.y;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (2 * sIdx >= r_pad) return;
    v4_t out0 = { (T)0, (T)0, (T)0, (T)0 }, out1 = out0;
    if (sIdx < m) {
        // sel != nullptr (an update of a subset -- the rescue stage's HI inliers, an LI update after a sliced RANSAC round): row pair sIdx is
        // measurement sel[sIdx], m is the subset's size (launch_ell_HP_build_sel: k_build_rows + k_ell_HP in one launch)
        const int i = meas[sel ? sel[sIdx] : sIdx];
        const int d = lm_type[i] == PRE3_INVDEPTH ? 6 : 3, off = lm_off[i];
        T v0[13], v1[13]; int cc[13];
#pragma unroll
        for (int t = 0; t < 7; ++t) { cc[t] = t; v0[t] = (T)Hc[14 * i + t]; v1[t] = (T)Hc[14 * i + 7 + t]; }
#pragma unroll
        for (int t = 0; t < 6; ++t) { cc[7 + t] = t < d ? off + t : 0; v0[7 + t] = t < d ? (T)Hl[12 * i + t] : (T)0; v1[7 + t] = t < d ? (T)Hl[12 * i + 6 + t] : (T)0; }
        const double nu0 = z[2 * i] - h[2 * i], nu1 = z[2 * i + 1] - h[2 * i + 1];
        if (blockIdx.x == 0) {
            if (threadIdx.x < 2 * ELLW) {
                const int t = threadIdx.x & (ELLW - 1), c = threadIdx.x >> 4;
                int cv = 0; T vt = (T)0;
#pragma unroll
                for (int u = 0; u < 13; ++u) if (u == t) { cv = cc[u]; vt = c ? v1[u] : v0[u]; }
                row_col[(2 * sIdx + c) * ELLW + t] = cv; row_val[(2 * sIdx + c) * ELLW + t] = vt;
            }
            if (threadIdx.x == 0) { row_nu[2 * sIdx] = nu0; row_nu[2 * sIdx + 1] = nu1; }
        }
        if (j < ld) {                                   // ld is a multiple of 128: the whole quad is inside
            v4_t pv[13];
#pragma unroll
            for (int t = 0; t < 13; ++t) pv[t] = *reinterpret_cast<const v4_t *>(P + (size_t)cc[t] * ld + j);
#pragma unroll
            for (int t = 0; t < 13; ++t) { out0 += v0[t] * pv[t]; out1 += v1[t] * pv[t]; }
        } else if (j == ld) {
            out0[0] = (T)nu0; out1[0] = (T)nu1;
        }
    } else if (blockIdx.x == 0) {
        if (threadIdx.x < 2 * ELLW) { row_col[2 * sIdx * ELLW + threadIdx.x] = 0; row_val[2 * sIdx * ELLW + threadIdx.x] = (T)0; }
        if (threadIdx.x == 0) { row_nu[2 * sIdx] = 0; row_nu[2 * sIdx + 1] = 0; }
    }
    if (j < ldw) {
        *reinterpret_cast<v4_t *>(dst + (size_t)(2 * sIdx) * ldw + j) = out0;
        *reinterpret_cast<v4_t *>(dst + (size_t)(2 * sIdx + 1) * ldw + j) = out1;
    }
}

// The same with MB measurements per workgroup (round 6): the seven pose rows of P -- 7 of the 13 rows every measurement gathers -- are read once per
// workgroup and column quad instead of once per measurement (35 instead of 64 MB through the L2s at N = 500), the measurements' coefficient rows
// come from an LDS table built once per workgroup, and all MB x 6 landmark-row loads of a thread are in flight together.  Same 13-term chain in the
// same order per entry: bit-identical to k_ell_HP_build (tests/test_gpu_variants.py, PRE3_HP_MB=0 restores that kernel).
#ifndef PRE3_PEND_KB
#define PRE3_PEND_KB 8
#endif
template <typename T, int MB, bool PEND = false>
__global__ __launch_bounds__(256) void k_ell_HP_build_mb(int m, int r_pad, const int32_t *__restrict__ meas, const int32_t *__restrict__ lm_type,
                                                         const int32_t *__restrict__ lm_off, const double *__restrict__ Hc,
                                                         const double *__restrict__ Hl, const double *__restrict__ z, const double *__restrict__ h,
                                                         int32_t *__restrict__ row_col, T *__restrict__ row_val, double *__restrict__ row_nu,
                                                         const T *__restrict__ P, int ld, T *__restrict__ dst, int ldw, const int32_t *__restrict__ need, int need_tag,
                                                         int ny_build, InnovRide ir, const int32_t *__restrict__ sel = nullptr, InboxRide ib = InboxRide{}, PendW pw = PendW{})
{
    if (ib.n16 > 0 && blockIdx.y == gridDim.y - 1) { if (blockIdx.x == 0) inbox_pull_block(ib); return; }
    if ((int)blockIdx.y >= ny_build) { innov_ride_block<T>(ir, ((int)blockIdx.y - ny_build) * gridDim.x + blockIdx.x); return; }
    typedef T v4_t __attribute__((ext_vector_type(4)));
    constexpr int PEND_MAX = 2 * NB;               // rows of a pending HI update (two panels)
    __shared__ __attribute__((aligned(16))) float gp[PEND ? MB * 2 * PEND_MAX : 1];     // PendW: gp[k][2 mi + c] = row c of measurement mi times row k of W~
    __shared__ T cv[MB][2][16];                    // coefficient rows of measurement s0 + mi: [c][t], t = 0..6 pose, 7..12 landmark
    __shared__ int ccs[MB][16];                    // their columns (= rows of P)
    __shared__ int state[MB];                      // 0: nothing to do (beyond r_pad, or not needed by this slice), 1: zero rows (padding), 2: a measurement
    __shared__ double nus[MB][2];
    const int s0 = blockIdx.y * MB, tid = threadIdx.x;
    if (tid < MB * 16) {
        const int mi = tid >> 4, t = tid & 15, sIdx = s0 + mi;
        int st = 0, cvv = 0; T a0 = (T)0, a1 = (T)0;
        if (2 * sIdx < r_pad && !(need != nullptr && sIdx < m && need[sIdx] != need_tag)) {
            st = 1;
            if (sIdx < m) {
                st = 2;
                const int i = meas[sel ? sel[sIdx] : sIdx];
                const int d = lm_type[i] == PRE3_INVDEPTH ? 6 : 3, off = lm_off[i];
                if (t < 7) { cvv = t; a0 = (T)Hc[14 * i + t]; a1 = (T)Hc[14 * i + 7 + t]; }
                else if (t < 13) { const int u = t - 7; cvv = u < d ? off + u : 0; a0 = u < d ? (T)Hl[12 * i + u] : (T)0; a1 = u < d ? (T)Hl[12 * i + 6 + u] : (T)0; }
                if (t == 13) { nus[mi][0] = z[2 * i] - h[2 * i]; nus[mi][1] = z[2 * i + 1] - h[2 * i + 1]; }
            } else if (t == 13) { nus[mi][0] = 0; nus[mi][1] = 0; }
        }
        cv[mi][0][t] = a0; cv[mi][1][t] = a1; ccs[mi][t] = cvv;
        if (t == 0) state[mi] = st;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < MB * 2 * ELLW) {
        // the rows for the kernels that follow (H*P*H', scoring, the update), as k_ell_HP_build's block column 0 leaves them
        const int mi = tid / (2 * ELLW), c = (tid / ELLW) & 1, t = tid & (ELLW - 1), sIdx = s0 + mi;
        if (state[mi] != 0) {
            row_col[(2 * sIdx + c) * ELLW + t] = t < 13 ? ccs[mi][t] : 0;
            row_val[(2 * sIdx + c) * ELLW + t] = t < 13 ? cv[mi][c][t] : (T)0;
            if (t == 0) row_nu[2 * sIdx + c] = nus[mi][c];
        }
    }
    const int j = (blockIdx.x * blockDim.x + tid) * 4;
    const int prow = PEND ? (pw.rows < PEND_MAX ? pw.rows : PEND_MAX) : 0;
    int any = 0;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) any |= state[mi] == 2;
    constexpr int GB = PEND ? 2 : (MB < 4 ? MB : 4);      // measurements whose landmark rows are in flight together (the pending form keeps 2 MB quads of results)
    v4_t pp[7], lp[GB][6];
    const bool inside = j < ld;                     // ld is a multiple of 128: the whole quad is inside
    auto load_lp = [&](const int h) {
#pragma unroll
        for (int u = 0; u < GB; ++u)
            if (state[h + u] == 2) {
#pragma unroll
                for (int t = 0; t < 6; ++t) lp[u][t] = *reinterpret_cast<const v4_t *>(P + (size_t)ccs[h + u][7 + t] * ld + j);
            }
    };
    if (inside && any) {                            // (the gathers of P are on their way while the pending rows' coefficients are worked out)
#pragma unroll
        for (int t = 0; t < 7; ++t) pp[t] = *reinterpret_cast<const v4_t *>(P + (size_t)t * ld + j);
        load_lp(0);
    }
    // P stands for P - W~'W~ (PRE3_OPT_PEND_HI): H*P loses (H W~') W~.  Eight rows of W~ per round trip, the next eight on their way while the last
    // are used (a workgroup has few neighbours on its CU to hide an L2 latency behind), the first eight already while the coefficients are worked out.
    constexpr int KB = PRE3_PEND_KB;
    const bool corr = PEND && prow > 0 && (int)(blockIdx.x * blockDim.x * 4) < ld;      // (block-uniform; the block column behind ld holds nu only)
    const bool corr_t = corr && inside && any;
    v4_t wa[KB], wb[KB];
    auto loadw = [&](v4_t (&w)[KB], const int k0) {
#pragma unroll
        for (int u = 0; u < KB; ++u) { const int k = k0 + u < prow ? k0 + u : prow - 1; w[u] = *reinterpret_cast<const v4_t *>(reinterpret_cast<const T *>(pw.W) + (size_t)k * pw.ldw + j); }
    };
    v4_t out[MB][2];
    auto applyw = [&](const v4_t (&w)[KB], const int k0) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (k0 + u < prow) {
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) { out[mi][0] -= (T)gp[(k0 + u) * (2 * MB) + 2 * mi] * w[u]; out[mi][1] -= (T)gp[(k0 + u) * (2 * MB) + 2 * mi + 1] * w[u]; }
            }
        }
    };
    if (corr) {
        // the MB x 2 x rows coefficients gp[k][2 mi + c] (the whole workgroup, before anybody leaves): thirteen scattered reads of W~ feed both rows of a measurement
        for (int e = tid; e < MB * prow; e += 256) {
            const int mi = e % MB, k = e / MB;
            float g0 = 0.f, g1 = 0.f;
            if (state[mi] == 2) {
                const float *wr = pw.W + (size_t)k * pw.ldw;
#pragma unroll
                for (int t = 0; t < 13; ++t) { const float wv = wr[ccs[mi][t]]; g0 += (float)cv[mi][0][t] * wv; g1 += (float)cv[mi][1][t] * wv; }
            }
            gp[k * (2 * MB) + 2 * mi] = g0; gp[k * (2 * MB) + 2 * mi + 1] = g1;
        }
        __syncthreads();
    }
    if (j >= ldw) return;
#pragma unroll
    for (int h = 0; h < MB; h += GB) {
        if (h > 0 && inside && any) load_lp(h);
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int mi = h + u;
            v4_t out0 = { (T)0, (T)0, (T)0, (T)0 }, out1 = out0;
            if (state[mi] == 2) {
                if (inside) {
#pragma unroll
                    for (int t = 0; t < 7; ++t) { out0 += cv[mi][0][t] * pp[t]; out1 += cv[mi][1][t] * pp[t]; }
#pragma unroll
                    for (int t = 0; t < 6; ++t) { out0 += cv[mi][0][7 + t] * lp[u][t]; out1 += cv[mi][1][7 + t] * lp[u][t]; }
                } else if (j == ld) {
                    out0[0] = (T)nus[mi][0]; out1[0] = (T)nus[mi][1];
                }
            }
            out[mi][0] = out0; out[mi][1] = out1;
        }
    }
    if (corr_t) {
        asm volatile("" ::: "memory");                       // (the rows of W~ are not fetched while the gathers of P hold the registers: 222 against 160)
        loadw(wa, 0);
        for (int k0 = 0; k0 < prow; k0 += 2 * KB) {
            if (k0 + KB < prow) loadw(wb, k0 + KB);
            applyw(wa, k0);
            if (k0 + 2 * KB < prow) loadw(wa, k0 + 2 * KB);
            if (k0 + KB < prow) applyw(wb, k0 + KB);
        }
    }
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
        const int sIdx = s0 + mi;
        if (state[mi] == 0) continue;
        *reinterpret_cast<v4_t *>(dst + (size_t)(2 * sIdx) * ldw + j) = out[mi][0];
        *reinterpret_cast<v4_t *>(dst + (size_t)(2 * sIdx + 1) * ldw + j) = out[mi][1];
    }
}

// dst[a][b] = sum_t val[b][t] * HP[a][col[b][t]] + (R ? R[a][b] : add_identity*delta_ab); padding = identity
template <typename T>
__global__ __launch_bounds__(64) void k_ell_G(int r, int r_pad, const int32_t *__restrict__ row_col, const T *__restrict__ row_val,
                                              const T *__restrict__ HP, int ldw, T *__restrict__ dst, int ldg, int add_identity,
                                              const T *__restrict__ Rd, int lower_only)
{
    int a = blockIdx.y;
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= r_pad || a >= r_pad) return;
    T out;
    if (lower_only && (int)(blockIdx.x * blockDim.x) > a) {
        out = (T)0;                   // strictly above the diagonal: every reader takes (max, min) -- no gathers, defined contents
    } else if (a < r && b < r) {
        T s = (T)0;
#pragma unroll
        for (int t = 0; t < ELLW; ++t) s = ell_fma(row_val[b * ELLW + t], HP[(size_t)a * ldw + row_col[b * ELLW + t]], s);
        if (Rd) s += Rd[(size_t)a * r + b];
        else if (add_identity && a == b) s += (T)1;
        out = s;
    } else {
        out = (a == b) ? (T)1 : (T)0;
    }
    dst[(size_t)a * ldg + b] = out;
}

// H*P*H' for a slice of a sharded RANSAC round: only the entries among the 2k rows of each hypothesis of [lo, hi) (the scorer reads nothing
// else of G), one thread per (hypothesis, pair); the arithmetic of k_ell_G entry by entry, so the values are the full build's.
template <typename T>
__global__ __launch_bounds__(256) void k_ell_G_hyp(const int32_t *__restrict__ hyp, int k, int lo, int hi, const int32_t *__restrict__ row_col,
                                                   const T *__restrict__ row_val, const T *__restrict__ HP, int ldw, T *__restrict__ dst, int ldg)
{
    const int np = k * (2 * k + 1);                       // pairs (i >= j) of 2k rows
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (hi - lo) * np) return;
    const int h = lo + g / np, p = g - (g / np) * np;
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= p) ++i;
    const int j = p - i * (i + 1) / 2;
    const int ri = 2 * hyp[h * k + (i >> 1)] + (i & 1), rj = 2 * hyp[h * k + (j >> 1)] + (j & 1);
    const int a = ri > rj ? ri : rj, b = ri > rj ? rj : ri;
    T s = (T)0;
#pragma unroll
    for (int t = 0; t < ELLW; ++t) s = ell_fma(row_val[b * ELLW + t], HP[(size_t)a * ldw + row_col[b * ELLW + t]], s);
    dst[(size_t)a * ldg + b] = s;
}

// LI rows are a subset of the measured rows whose H*P and H*P*H' already exist (computed once per step for RANSAC):
// gather them instead of recomputing.  sel[s] = measurement index of selected landmark s; row a = 2*sel[a/2] + (a&1).
// ------------------------------------------------------------------------------------------------
// Blocked Cholesky of S fused with the forward solve W = L^-1 [HP | nu].
// The stacked matrix M = [S ; HP'] (rows: r_pad rows of S, then the ldw columns of HP as rows) is swept
// right-looking in NB=64 panels.  Panel step J:
//   k_chol_panel : every workgroup factors the (already updated) diagonal block M_JJ in LDS; workgroup b
//                  then solves its own 64-row block X <- M_bJ L_JJ^-T.   (S blocks below J, and all W strips)
//   k_chol_trail : M_bK -= M_bJ M_KJ'  for K > J                         (S lower tiles and all W strips)
// W strip c holds M(i,a) = W[a][c*64+i] (transposed storage), so its loads/stores are coalesced in i.
// ------------------------------------------------------------------------------------------------

// A solved 64-row block of a W strip leaves LDS (X[a][i] = W[J*64 + a][c0 + i]): the f32 rows and, fp32 with k_downdate_b3 in use (Wp), this
// strip's share of the bf16 planes of row block J, for the down-date and for the next launch's pending update.  Strip = half a 128-column
// block: fragments 2*half, 2*half+1 of 4 stages; the nu strip lives in an extra column block.
template <typename T, typename XT>
__device__ __forceinline__ void chol_store_w_strip(const XT &X, T *__restrict__ W, int ldw, int J, int c0, void *__restrict__ Wp, int nst_total, int ld)
{
    const int tid = threadIdx.x;
    typedef T st4_t __attribute__((ext_vector_type(4)));
    for (int idx = tid; idx < NB * NB / 4; idx += CH_NTH) {
        const int a2 = idx >> 4, i = (idx & 15) * 4;
        *reinterpret_cast<st4_t *>(W + (size_t)(J * NB + a2) * ldw + c0 + i) = st4_t{ X[a2][i], X[a2][i + 1], X[a2][i + 2], X[a2][i + 3] };
    }
    if constexpr (sizeof(T) == 4) {
        if (Wp != nullptr && c0 < ld + NB) {
            bf16x8_t *base = static_cast<bf16x8_t *>(Wp) + ((size_t)(c0 >> 7) * nst_total + 4 * J) * B3_GRAN + ((c0 >> 6) & 1) * 128;
            for (int idx = tid; idx < 512; idx += CH_NTH) {
                const int q = idx >> 7, fl = (idx >> 6) & 1, l = idx & 63, r = l & 31, h = l >> 5;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = X[16 * q + 8 * h + j][fl * 32 + r];
                b3_split_store(x, base + (size_t)q * B3_GRAN + fl * 64 + l);
            }
        }
    }
}

// PRO: the workgroup first applies the update of panel J-1 to its own blocks (diagonal block and X), i.e. the K = J
// column of the trailing update, so that the launch of panel J does not have to wait for a separate trail kernel.
// PREBUILT: the raw diagonal block (Ls) and the workgroup's X block (Xs) are already in LDS (k_hi_fused computes them there): no global loads,
// and nobody reads the raw diagonal block from global memory, so workgroup 0 need not wait before it stores L_JJ over it.
template <typename T, bool PRO, bool EARLY = false, bool PREBUILT = false>
__device__ __forceinline__ void chol_panel_body(ChSmem<T> &sm, T *__restrict__ S, int lds, T *__restrict__ W, int ldw, int J, int nrb,
                                                int32_t *__restrict__ status, const int b, unsigned int *__restrict__ arrive, unsigned int target,
                                                void *__restrict__ Wp = nullptr, int nst_total = 0, int ld = 0, void *__restrict__ Sp = nullptr, int sp_stride = 0,
                                                int rows_last = 0 /* > 0: real rows of the LAST panel (the rest of it is padding) */,
                                                int npend = 1 /* planes form: the updates of panels J-npend .. J-1 are pending for this column (the trailing launches sweep several panels at a time) */)
{
    PROBE_STAMP(0);
    auto &Ls = sm.Ls; auto &Xs = sm.Xs;
    const int tid = threadIdx.x;
    const int nS = nrb - J - 1;
    const bool isW = b > nS;
    const int c0 = isW ? (b - nS - 1) * NB : 0;
    // Ten waves.  0-3: D workers (tile (w>>1, w&1) of the diagonal block), 4-7: X workers (the same tiles of the workgroup's X block),
    // 8: factor wave, 9: z wave.  A workgroup's waves are dealt to the CU's four SIMDs cyclically (w, w+4, w+8 share one): every SIMD
    // gets one D and one X worker -- the prologue's and the chain's MFMA work, and the matrix pipe is per SIMD -- and the two chain waves
    // sit beside workers whose chain work is MFMA only, so their vector instructions issue unhindered.
    const int wave = tid >> 6, lane = tid & 63;
    const int role = wave == 8 ? 0 : wave == 9 ? 1 : 2;        // 0: factor wave, 1: z wave, 2: worker
    const bool worker = role >= 2, xside = wave >= 4 && wave < 8, hasX = b >= 1;
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    else if (role == 1) __builtin_amdgcn_s_setprio(2);
    const int wv = wave & 3;                                   // worker tile
    const int wl = tid & 255;                                  // lane id within the D (or X) worker group: 256 lanes load one 64x64 block
    typedef int frag_t __attribute__((ext_vector_type(4)));      // 8 bf16
    const bool planes = sizeof(T) == 4 && Sp != nullptr;       // fp32 with the bf16-split down-date: the pending update multiplies on the bf16 MFMA too
    frag_t fA[4][3], fB[4][3];                                 // pending update of this wave's tile: A operand (rows) and B operand (columns)
    const int w0 = (wv >> 1) * 32, w1 = (wv & 1) * 32, fa = wv >> 1, fb = wv & 1;
    const bool tile_live = worker && (xside ? hasX : wv != 1);  // (the D tile above the diagonal is never read)
    if (!PREBUILT && worker) {
        // every global load of the wave is issued before its first LDS store: the raw 64x64 block (D workers: the diagonal block,
        // X workers: the workgroup's own block) and, for the pending update of panel J-1, the operands of THIS wave's tile
        T g[16], gp[16];
        const int lr = wl >> 6, lc = wl & 63;
        if (!xside) {
#pragma unroll
            for (int t = 0; t < 16; ++t) g[t] = S[(size_t)(J * NB + lr + 4 * t) * lds + J * NB + lc];
        } else if (hasX) {
            if (!isW) {
                const int rb = J + b;
#pragma unroll
                for (int t = 0; t < 16; ++t) g[t] = S[(size_t)(rb * NB + lr + 4 * t) * lds + J * NB + lc];
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) g[t] = W[(size_t)(J * NB + lr + 4 * t) * ldw + c0 + lc];
            }
        }
        if (PRO && planes) {
            // operands as bf16 planes (written by the store epilogues of launch J-1), this wave's fragments straight into registers.
            // A operand: rows w0.. of B = M(J, J-1).  B operand: D tile -> rows w1.. of B;  X tile -> the own block's rows / columns w1..
            if (tile_live && J > 0) {                    // (panel 0 has no pending update: pacc stays 0)
                const int Jp = J - npend;                 // (several pending panels: the oldest first, as the sweep would have applied them)
                const frag_t *Bp = static_cast<const frag_t *>(Sp) + ((size_t)J * sp_stride + Jp) * B3_SGRAN + lane;
                const frag_t *Op; int ostage, oplane;
                if (!xside) { Op = Bp + fb * 64; ostage = 384; oplane = 128; }
                else if (!isW) { Op = static_cast<const frag_t *>(Sp) + ((size_t)(J + b) * sp_stride + Jp) * B3_SGRAN + fb * 64 + lane; ostage = 384; oplane = 128; }
                else { Op = static_cast<const frag_t *>(Wp) + ((size_t)(c0 >> 7) * nst_total + 4 * Jp) * B3_GRAN + (2 * ((c0 >> 6) & 1) + fb) * 64 + lane; ostage = B3_GRAN; oplane = 256; }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) { fA[q][pl] = Bp[q * 384 + pl * 128 + fa * 64]; fB[q][pl] = Op[q * ostage + pl * oplane]; }
            }
        } else if (PRO && J > 0) {
            // operands of the pending update through LDS: D workers bring B = M(J, J-1), X workers the workgroup's own M(b, J-1)
            if (!xside) {
#pragma unroll
                for (int t = 0; t < 16; ++t) gp[t] = S[(size_t)(J * NB + lr + 4 * t) * lds + (J - 1) * NB + lc];
            } else if (hasX) {
                if (!isW) {
                    const int rb = J + b;
#pragma unroll
                    for (int t = 0; t < 16; ++t) gp[t] = S[(size_t)(rb * NB + lr + 4 * t) * lds + (J - 1) * NB + lc];
                } else {
#pragma unroll
                    for (int t = 0; t < 16; ++t) gp[t] = W[(size_t)((J - 1) * NB + lr + 4 * t) * ldw + c0 + lc];
                }
            }
        }
        if (!xside) {
#pragma unroll
            for (int t = 0; t < 16; ++t) Ls[lr + 4 * t][lc] = g[t];
        } else if (hasX) {
            if (!isW) {
#pragma unroll
                for (int t = 0; t < 16; ++t) Xs[lc][lr + 4 * t] = g[t];        // S block: (i = lr+4t, a = lc) -> Xs[a][i]
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) Xs[lr + 4 * t][lc] = g[t];        // W strip: (a = lr+4t, i = lc)
            }
        }
        if (PRO && !planes && J > 0) {
            if (!xside) {
#pragma unroll
                for (int t = 0; t < 16; ++t) sm.Bs[lr + 4 * t][lc] = gp[t];                      // Bs[j][a]
            } else if (hasX) {
#pragma unroll
                for (int t = 0; t < 16; ++t) sm.As[lr + 4 * t][lc] = gp[t];                      // As: [i][a] (S) or [a][i] (W)
            }
        }
    }
    __syncthreads();
    PROBE_STAMP(4);
    // every workgroup reads the raw diagonal block; workgroup 0 overwrites it with L_JJ at the end and must not do so
    // before all of them have it (they normally start together, but nothing guarantees that for very large grids)
    if (!PREBUILT && tid == 0 && b != 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    typename ChW<T>::acc_t acc[ChW<T>::NBLK][ChW<T>::NBLK];        // this worker wave's tile of D or X, in MFMA accumulator layout
    bool acc_loaded = false;
    if (PRO && planes) {
        if constexpr (sizeof(T) == 4) {
            if (tile_live) {
                // six bf16 products per f32 product (see k_downdate_b3), K = 64 = 4 k-steps: 24 MFMAs of 32 cycles per wave.  The 32x32
                // result has the accumulator layout the chain keeps its tile in: tile = raw block - product, no LDS round trip.
                f32x16_t pacc;
#pragma unroll
                for (int e = 0; e < 16; ++e) pacc[e] = 0.f;
#define PRO_MMA(px, py) pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[q][px]), __builtin_bit_cast(bf16x8_t, fB[q][py]), pacc, 0, 0, 0)
                if (J > 0) {
                    // (several pending panels: the older ones' updates of an S block below the diagonal used to come from the trailing sweep, whose tile has
                    //  the two factors in the other operand roles -- its six plane products in that sweep's order, so that the sum rounds as there)
                    if (npend > 1 && xside && !isW) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { PRO_MMA(0, 0); PRO_MMA(1, 0); PRO_MMA(0, 1); PRO_MMA(1, 1); PRO_MMA(2, 0); PRO_MMA(0, 2); }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { PRO_MMA(0, 0); PRO_MMA(0, 1); PRO_MMA(1, 0); PRO_MMA(1, 1); PRO_MMA(0, 2); PRO_MMA(2, 0); }
                    }
                }
#undef PRO_MMA
                const int lrow = 4 * (lane >> 5), lcol = lane & 31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = w0 + (e & 3) + 8 * (e >> 2) + lrow, c = w1 + lcol;
                    acc[0][0][e] = (xside ? Xs[r][c] : Ls[r][c]) - pacc[e];
                }
                for (int u = 1; u < npend; ++u) {
                    // ... then the next pending panel's: its fragments now (the registers are free again), the same six products -- in the sweep's order for
                    // all but the newest panel, whose update was always this launch's --, subtracted from the rounded difference
                    const int Jp = J - npend + u;
                    const frag_t *Bp = static_cast<const frag_t *>(Sp) + ((size_t)J * sp_stride + Jp) * B3_SGRAN + lane;
                    const frag_t *Op; int ostage, oplane;
                    if (!xside) { Op = Bp + fb * 64; ostage = 384; oplane = 128; }
                    else if (!isW) { Op = static_cast<const frag_t *>(Sp) + ((size_t)(J + b) * sp_stride + Jp) * B3_SGRAN + fb * 64 + lane; ostage = 384; oplane = 128; }
                    else { Op = static_cast<const frag_t *>(Wp) + ((size_t)(c0 >> 7) * nst_total + 4 * Jp) * B3_GRAN + (2 * ((c0 >> 6) & 1) + fb) * 64 + lane; ostage = B3_GRAN; oplane = 256; }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) { fA[q][pl] = Bp[q * 384 + pl * 128 + fa * 64]; fB[q][pl] = Op[q * ostage + pl * oplane]; }
#pragma unroll
                    for (int e = 0; e < 16; ++e) pacc[e] = 0.f;
#define PRO_MMA(px, py) pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[q][px]), __builtin_bit_cast(bf16x8_t, fB[q][py]), pacc, 0, 0, 0)
                    if (u < npend - 1 && xside && !isW) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { PRO_MMA(0, 0); PRO_MMA(1, 0); PRO_MMA(0, 1); PRO_MMA(1, 1); PRO_MMA(2, 0); PRO_MMA(0, 2); }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { PRO_MMA(0, 0); PRO_MMA(0, 1); PRO_MMA(1, 0); PRO_MMA(1, 1); PRO_MMA(0, 2); PRO_MMA(2, 0); }
                    }
#undef PRO_MMA
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][0][e] = acc[0][0][e] - pacc[e];
                }
                if (PRE3_CHAIN_ASYNC && !xside) {
                    // the flag-driven chain's D workers read their tiles from Ls, in their own block layout (pre3_chain_async.h): the updated tile goes
                    // back where the raw one came from (each wave its own tile: no barrier)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Ls[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol] = acc[0][0][e];
                }
            }
            acc_loaded = true;
        }
    } else if (PRO) {
        if (tile_live) {
            using M = Mfma<T>;
            constexpr int NBLK = 32 / M::BLK;
            typename M::acc_t pacc[NBLK][NBLK];
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int q = 0; q < NBLK; ++q)
#pragma unroll
                    for (int e = 0; e < M::NREG; ++e) pacc[p][q][e] = (T)0;
            // tile(r, c) -= sum_k A[r][k] B[c][k]: A = rows w0.. of Bs (= M(J, J-1));  B = rows w1.. of Bs (D tile), of As[i][k] (S block: the
            // own block's rows) or the columns of As[k][i] (W strip)
#pragma unroll 4
            for (int k0 = 0; k0 < (J > 0 ? NB : 0); k0 += M::KS) {
                const int kx = k0 + M::kk(lane);
                T av[NBLK], bv[NBLK];
#pragma unroll
                for (int p = 0; p < NBLK; ++p) {
                    av[p] = sm.Bs[w0 + p * M::BLK + M::col(lane)][kx];
                    bv[p] = !xside ? sm.Bs[w1 + p * M::BLK + M::col(lane)][kx] : !isW ? sm.As[w1 + p * M::BLK + M::col(lane)][kx] : sm.As[kx][w1 + p * M::BLK + M::col(lane)];
                }
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int q = 0; q < NBLK; ++q) {
                        if (sizeof(T) == 8 && !xside && w0 == w1 && q > p) continue;      // (fp64: the 16 x 16 block above the diagonal is never read)
                        M::mma(av[p], bv[q], pacc[p][q]);
                    }
            }
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int q = 0; q < NBLK; ++q)
#pragma unroll
                    for (int e = 0; e < M::NREG; ++e) {
                        const int r = w0 + p * M::BLK + M::row(lane, e), c = w1 + q * M::BLK + M::col(lane);
                        acc[p][q][e] = (xside ? Xs[r][c] : Ls[r][c]) - pacc[p][q][e];
                        if (((sizeof(T) == 4 && PRE3_CHAIN_ASYNC) || (sizeof(T) == 8 && PRE3_CHAIN_ASYNC_F64)) && !xside) Ls[r][c] = acc[p][q][e];      // (the flag-driven chain's D workers read Ls)
                    }
        }
        acc_loaded = true;
        __syncthreads();                            // As (an operand of the products above) is the chain's hand-off buffer from here on
    }
    PROBE_STAMP(1);
    bool bad = false;
    // (the last panel of an update whose row count is not a multiple of 64 stops behind its last real sub-panel; PRE3_CHOL_EARLY=0 at the host: never)
    const int nsp_eff = (rows_last > 0 && J == nrb - 1) ? (rows_last + CH_MB - 1) / CH_MB : CH_NSP;
    // (EARLY is a KERNEL variant: the step-skipping form's uniform branches cost a full chain ~6 % -- measured in the persistent kernel, 14.35 k ->
    // 15.3 k cycles per panel -- and both forms in one kernel spill; the host launches it for a last panel that really has padding)
    if constexpr ((sizeof(T) == 4 && PRE3_CHAIN_ASYNC) || (sizeof(T) == 8 && PRE3_CHAIN_ASYNC_F64)) {
        // round 6: fp32 takes the flag-driven chain (no barrier per pipeline step; the step-skipping form costs it nothing, so EARLY or not is one code).
        // worker_init: the X workers keep the tile the prologue left in `acc`; the D workers' went back to Ls above.
        chol_chain_async<T, false, false>(sm, acc, acc_loaded, hasX, bad, ChaNoSide{},
                                           [](typename ChW<T>::acc_t (&)[ChW<T>::NBLK][ChW<T>::NBLK], bool, int) {}, ChaFromLs{},
                                           EARLY ? __builtin_amdgcn_readfirstlane(nsp_eff) : CH_NSP);
    } else {
        if constexpr (EARLY) chol_chain<T, false, true>(sm, acc, acc_loaded, hasX, bad, [](int) {}, __builtin_amdgcn_readfirstlane(nsp_eff));
        else chol_chain<T, false, false>(sm, acc, acc_loaded, hasX, bad, [](int) {});
    }
    PROBE_STAMP(2);
    if (bad && (tid & 63) == 0 && b == 0) atomicExch(status, 1);
    if (b == 0) {
        if (!PREBUILT && tid == 0) {
            bounded_wait(arrive, target, status + 1);        // status + 1 = stats[7], the wait guard (a give-up is reported as PRE3_E_HIP)
        }
        __syncthreads();
        for (int idx = tid; idx < NB * NB; idx += CH_NTH) {
            int i = idx >> 6, a2 = idx & 63;
            S[(size_t)(J * NB + i) * lds + J * NB + a2] = a2 <= i ? Ls[i][a2] : (T)0;
        }
        return;
    }
    if (!isW) {
        int rb = J + b;
        typedef T st4_t __attribute__((ext_vector_type(4)));
        for (int idx = tid; idx < NB * NB / 4; idx += CH_NTH) {             // 16- / 32-byte stores: a quarter of the store instructions
            const int i = idx >> 4, a2 = (idx & 15) * 4;
            *reinterpret_cast<st4_t *>(S + (size_t)(rb * NB + i) * lds + J * NB + a2) = st4_t{ Xs[a2][i], Xs[a2 + 1][i], Xs[a2 + 2][i], Xs[a2 + 3][i] };
        }
        if constexpr (sizeof(T) == 4) {
            if (Sp != nullptr) {           // the next launches' pending updates read this block as bf16 planes: [k-step][plane][32-row half][lane]
                bf16x8_t *base = static_cast<bf16x8_t *>(Sp) + ((size_t)rb * sp_stride + J) * B3_SGRAN;
                for (int idx = tid; idx < 512; idx += CH_NTH) {
                    const int q = idx >> 7, fl = (idx >> 6) & 1, l = idx & 63, r = l & 31, h = l >> 5;
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = Xs[16 * q + 8 * h + j][fl * 32 + r];
                    b3_split_store<128>(x, base + q * 384 + fl * 64 + l);
                }
            }
        }
    } else {
        chol_store_w_strip<T>(Xs, W, ldw, J, c0, (Sp != nullptr || J == nrb - 1) ? Wp : nullptr, nst_total, ld);
    }
    PROBE_STAMP(3);
}

// One 64x64 tile of the trailing update with its operands as bf16 planes (written by the store epilogues of the launch that solved panel J):
// fragments straight from global memory, six bf16 products per f32 product, no LDS and no barrier -- a third of the matrix-pipe time of
// the f32 form.  256 threads.
// J2 >= 0 (round 5): the updates of the panels J .. J2 in one pass over the tile -- ((C - A_J B_J') - A_J+1 B_J+1') - .., each product rounded and
// subtracted as the one-panel-per-launch sweep does, so the bits are the same; the tile is read and written once instead of once per panel.
__device__ __forceinline__ void chol_trail_b3_tile(float *__restrict__ S, int lds, float *__restrict__ W, int ldw, int J, bool isW, int rb, int K, int c0,
                                       const void *__restrict__ Wp, int nst_total, const void *__restrict__ Sp, int sp_stride, int J2 = -1)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w0 = (wave >> 1) * 32, w1 = (wave & 1) * 32;
    typedef int frag_t __attribute__((ext_vector_type(4)));
    const int fa = wave >> 1, fb = wave & 1, lrow = 4 * (lane >> 5), lcol = lane & 31;
    float cvv[16];
    float *Cbase = !isW ? S + (size_t)(rb * NB + w0) * lds + K * NB + w1 + lcol : W + (size_t)(K * NB + w0) * ldw + c0 + w1 + lcol;
    const int cld = !isW ? lds : ldw;
    for (int Jp = J; Jp <= (J2 >= 0 ? J2 : J); ++Jp) {
        const int pass = Jp - J;
        const frag_t *Ap, *Bp;
        int bstage, bplane;
        if (!isW) {
            Ap = static_cast<const frag_t *>(Sp) + ((size_t)rb * sp_stride + Jp) * B3_SGRAN + fa * 64 + lane;
            Bp = static_cast<const frag_t *>(Sp) + ((size_t)K * sp_stride + Jp) * B3_SGRAN + fb * 64 + lane; bstage = 384; bplane = 128;
        } else {
            Ap = static_cast<const frag_t *>(Sp) + ((size_t)K * sp_stride + Jp) * B3_SGRAN + fa * 64 + lane;
            Bp = static_cast<const frag_t *>(Wp) + ((size_t)(c0 >> 7) * nst_total + 4 * Jp) * B3_GRAN + (2 * ((c0 >> 6) & 1) + fb) * 64 + lane; bstage = B3_GRAN; bplane = 256;
        }
        frag_t fA[4][3], fB[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { fA[q][pl] = Ap[q * 384 + pl * 128]; fB[q][pl] = Bp[q * bstage + pl * bplane]; }
        if (pass == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) cvv[e] = Cbase[(size_t)((e & 3) + 8 * (e >> 2) + lrow) * cld];
        }
        f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#define TR_MMA(px, py) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[q][px]), __builtin_bit_cast(bf16x8_t, fB[q][py]), acc, 0, 0, 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) { TR_MMA(0, 0); TR_MMA(0, 1); TR_MMA(1, 0); TR_MMA(1, 1); TR_MMA(0, 2); TR_MMA(2, 0); }
#undef TR_MMA
#pragma unroll
        for (int e = 0; e < 16; ++e) cvv[e] = cvv[e] - acc[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) Cbase[(size_t)((e & 3) + 8 * (e >> 2) + lrow) * cld] = cvv[e];
}

// tile number -> (S tile (rb, K) | W tile (K, c0)) of the update of panel J applied to column blocks >= K0
__device__ __forceinline__ void chol_trail_decode(int idx, int K0, int nrb, int nW, bool &isW, int &rb, int &K, int &c0)
{
    const int nK = nrb - K0, nSt = nK * (nK + 1) / 2;
    rb = 0; c0 = 0;
    if (idx < nSt) {
        int bb = 0;
        while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
        rb = K0 + bb; K = K0 + idx - bb * (bb + 1) / 2; isW = false;
    } else {
        const int t = idx - nSt;
        c0 = (t % nW) * NB; K = K0 + t / nW; isW = true;
    }
}

// The trailing update as a launch of its own (no LDS: 8 workgroups per CU instead of the 2 that k_chol_step's 75 KB allow): used when
// the update has far more tiles than the chip has slots (large r and n), where it -- not the panel's chain -- sets the pace.
__global__ __launch_bounds__(256) void k_chol_trail_b3(float *__restrict__ S, int lds, float *__restrict__ W, int ldw, int J, int K0, int nrb, int nW,
                                                       const void *__restrict__ Wp, int nst_total, const void *__restrict__ Sp, int sp_stride, int J2)
{
    bool isW; int rb, K, c0;
    chol_trail_decode(blockIdx.x, K0, nrb, nW, isW, rb, K, c0);
    chol_trail_b3_tile(S, lds, W, ldw, J, isW, rb, K, c0, Wp, nst_total, Sp, sp_stride, J2);
}

// The same sweep with the W part in 128 x 128 super-tiles (round 5): a wave owns a 64 x 64 tile -- four 32 x 32 accumulators that share two A and
// two B fragment sets per k-step, i.e. half the operand bytes per MFMA of the 32 x 32-per-wave form, whose launches were bound by the rate at
// which a CU takes operand fragments (twice the matrix-pipe time at N = 2000), not by the read-modify-write of W any more once four panels
// share a pass.  Every 32 x 32 tile sees the products of the one-panel sweep in its order: the same bits.  S tiles (a tenth of the work) keep
// the 64 x 64 form.  No LDS, no barrier.  (measured, N = 2000: the eight sweeps 759 -> 696 us; what is left is the fragments' latency at two waves
// per SIMD -- requesting a k-step ahead by hand spilled (741 us), parking the tile's own values in LDS did not raise the occupancy: the compiler
// hoists every load the register budget allows.  k_chol_trail_b3l below stages the operands through LDS instead: 635 us.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_chol_trail_b3w(float *__restrict__ S, int lds, float *__restrict__ W, int ldw, int J, int K0, int nrb, int nW,
                                                        const void *__restrict__ Wp, int nst_total, const void *__restrict__ Sp, int sp_stride, int J2)
{
    const int nK = nrb - K0, nSt = nK * (nK + 1) / 2;
    if ((int)blockIdx.x < nSt) {
        bool isW; int rb, K, c0;
        chol_trail_decode(blockIdx.x, K0, nrb, nW, isW, rb, K, c0);
        chol_trail_b3_tile(S, lds, W, ldw, J, false, rb, K, 0, Wp, nst_total, Sp, sp_stride, J2);
        return;
    }
    typedef int frag_t __attribute__((ext_vector_type(4)));
    const int t = blockIdx.x - nSt, nWp = (nW + 1) >> 1;
    const int kp = t / nWp, wp = t - kp * nWp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = K0 + 2 * kp + (wave >> 1), cb = 2 * wp + (wave & 1), c0 = cb * NB;
    if (K >= nrb || cb >= nW) return;
    const int lrow = 4 * (lane >> 5), lcol = lane & 31;
    float *Cbase = W + (size_t)(K * NB) * ldw + c0 + lcol;
    float cvv[2][2][16];
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int e = 0; e < 16; ++e) cvv[fa][fb][e] = Cbase[(size_t)(32 * fa + (e & 3) + 8 * (e >> 2) + lrow) * ldw + 32 * fb];
    for (int Jp = J; Jp <= (J2 >= 0 ? J2 : J); ++Jp) {
        const frag_t *Ap = static_cast<const frag_t *>(Sp) + ((size_t)K * sp_stride + Jp) * B3_SGRAN + lane;
        const frag_t *Bp = static_cast<const frag_t *>(Wp) + ((size_t)(c0 >> 7) * nst_total + 4 * Jp) * B3_GRAN + (2 * ((c0 >> 6) & 1)) * 64 + lane;
        f32x16_t acc[2][2];
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[fa][fb][e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            frag_t fA[2][3], fB[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { fA[h][pl] = Ap[q * 384 + pl * 128 + h * 64]; fB[h][pl] = Bp[q * B3_GRAN + pl * 256 + h * 64]; }
            // (plane pairs outside, the four tiles inside: four independent accumulators between two products of the same tile)
#define TW_MMA(px, py) \
            _Pragma("unroll") for (int fa = 0; fa < 2; ++fa) \
                _Pragma("unroll") for (int fb = 0; fb < 2; ++fb) \
                    acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[fa][px]), __builtin_bit_cast(bf16x8_t, fB[fb][py]), acc[fa][fb], 0, 0, 0)
            TW_MMA(0, 0); TW_MMA(0, 1); TW_MMA(1, 0); TW_MMA(1, 1); TW_MMA(0, 2); TW_MMA(2, 0);
#undef TW_MMA
        }
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) cvv[fa][fb][e] = cvv[fa][fb][e] - acc[fa][fb][e];
    }
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int e = 0; e < 16; ++e) Cbase[(size_t)(32 * fa + (e & 3) + 8 * (e >> 2) + lrow) * ldw + 32 * fb] = cvv[fa][fb][e];
}

// The super-tile sweep with the operands staged through LDS (k_downdate_b3's form): per k-step the workgroup's four operand sub-blocks -- the
// S planes of its two row blocks, the W planes of its 128 columns: 24 KB -- come in ONCE by LDS-DMA (six 1-KB requests per wave, no registers
// held while they travel), three stages deep, and every wave takes its twelve fragments from LDS.  A quarter of the 32 x 32-per-wave form's
// operand traffic, and the fragments' latency is the pipeline's, not the wave's.  Same products per 32 x 32 tile in the same order.
// (measured, N = 2000: the eight sweeps of an update 759 us with 64 x 64 tiles -> 696 us (k_chol_trail_b3w) -> 635 us; step 3.99 -> 3.81 ms.  Two
// workgroups per CU and two stages of lookahead still leave the start of a workgroup -- its tile of W from HBM -- and part of the DMA latency exposed.)
constexpr int TL_STAGE = 1536, TL_NSLOT = 3;                // granules per stage: A0 384 | A1 384 | B 768
__global__ __launch_bounds__(256) void k_chol_trail_b3l(float *__restrict__ S, int lds, float *__restrict__ W, int ldw, int J, int K0, int nrb, int nW,
                                                        const void *__restrict__ Wp, int nst_total, const void *__restrict__ Sp, int sp_stride, int J2)
{
    const int nK = nrb - K0, nSt = nK * (nK + 1) / 2;
    if ((int)blockIdx.x < nSt) {
        bool isW; int rb, K, c0;
        chol_trail_decode(blockIdx.x, K0, nrb, nW, isW, rb, K, c0);
        chol_trail_b3_tile(S, lds, W, ldw, J, false, rb, K, 0, Wp, nst_total, Sp, sp_stride, J2);
        return;
    }
    typedef int frag_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
    frag_t *ops = reinterpret_cast<frag_t *>(tl_smem);
    const int t = blockIdx.x - nSt, nWp = (nW + 1) >> 1;
    const int kp = __builtin_amdgcn_readfirstlane(t / nWp), wp = __builtin_amdgcn_readfirstlane(t - kp * nWp);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wa = wave >> 1, wb = wave & 1;
    const int Kb0 = K0 + 2 * kp, Kb1 = Kb0 + 1 < nrb ? Kb0 + 1 : Kb0;          // (a missing second row block / column block is fetched as a copy of the first: the
    const bool vB1 = 2 * wp + 1 < nW;                                             //  DMA counts stay uniform; its waves skip the arithmetic)
    const int K = Kb0 + wa, cb = 2 * wp + wb, c0 = cb * NB;
    const bool mine = K < nrb && cb < nW;
    const int lrow = 4 * (lane >> 5), lcol = lane & 31;
    float *Cbase = W + (size_t)(K * NB) * ldw + c0 + lcol;
    float cvv[2][2][16];
    if (mine) {
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) cvv[fa][fb][e] = Cbase[(size_t)(32 * fa + (e & 3) + 8 * (e >> 2) + lrow) * ldw + 32 * fb];
    }
    // this wave's six DMA requests of a stage: request idx = wave + 4 u -> [0, 12): S planes (row block idx / 6, plane, half); [12, 24): W planes
    // (plane, 32-column fragment).  Both sources are linear in the step it = 4 (panel - J) + k-step: + 384 granules (S), + B3_GRAN (W).
    const frag_t *src[6]; int dst[6], stp[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int idx = wave + 4 * u;
        if (idx < 12) {
            const int a_ = idx / 6, r = idx - 6 * a_;
            src[u] = static_cast<const frag_t *>(Sp) + ((size_t)(a_ ? Kb1 : Kb0) * sp_stride + J) * B3_SGRAN + (r >> 1) * 128 + (r & 1) * 64 + lane;
            dst[u] = a_ * 384 + (r >> 1) * 128 + (r & 1) * 64; stp[u] = 384;
        } else {
            const int r = idx - 12, pl = r >> 2, f = r & 3, fs = (!vB1 && f >= 2) ? f - 2 : f;
            src[u] = static_cast<const frag_t *>(Wp) + ((size_t)wp * nst_total + 4 * J) * B3_GRAN + pl * 256 + fs * 64 + lane;
            dst[u] = 768 + pl * 256 + f * 64; stp[u] = B3_GRAN;
        }
    }
    const int nIt = 4 * ((J2 >= 0 ? J2 : J) - J + 1);
    auto issue = [&](int it) {
        frag_t *slot = ops + (it % TL_NSLOT) * TL_STAGE;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds(src[u] + (size_t)it * stp[u], (__attribute__((address_space(3))) void *)(slot + dst[u]), 16, 0, 0);
    };
    issue(0);
    issue(1);
    f32x16_t acc[2][2];
    for (int it = 0; it < nIt; ++it) {
        if (it + 1 < nIt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave's share of stage `it` has landed; the slot of stage it-1 is free
        if (it + 2 < nIt) issue(it + 2);
        if (!mine) continue;
        const frag_t *sl = ops + (it % TL_NSLOT) * TL_STAGE + lane;
        frag_t fA[2][3], fB[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { fA[h][pl] = sl[wa * 384 + pl * 128 + h * 64]; fB[h][pl] = sl[768 + pl * 256 + (2 * wb + h) * 64]; }
        if ((it & 3) == 0) {
#pragma unroll
            for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[fa][fb][e] = 0.f;
        }
#define TW_MMA(px, py) \
        _Pragma("unroll") for (int fa = 0; fa < 2; ++fa) \
            _Pragma("unroll") for (int fb = 0; fb < 2; ++fb) \
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[fa][px]), __builtin_bit_cast(bf16x8_t, fB[fb][py]), acc[fa][fb], 0, 0, 0)
        TW_MMA(0, 0); TW_MMA(0, 1); TW_MMA(1, 0); TW_MMA(1, 1); TW_MMA(0, 2); TW_MMA(2, 0);
#undef TW_MMA
        if ((it & 3) == 3) {
#pragma unroll
            for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) cvv[fa][fb][e] = cvv[fa][fb][e] - acc[fa][fb][e];
        }
    }
    if (mine) {
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) Cbase[(size_t)(32 * fa + (e & 3) + 8 * (e >> 2) + lrow) * ldw + 32 * fb] = cvv[fa][fb][e];
    }
}

// Trailing update on the matrix cores: one 64 x 64 tile  C -= A B'  (K = 64) per workgroup, 4 waves, each a
// 32 x 32 sub-tile.  S-type tiles keep (row, a) order in LDS and are read with a 65-float stride (conflict
// free); W-type tiles are stored k-major.  For W strips the operands are swapped so that the accumulator's
// lane index runs along i, the contiguous direction of W.
// J: the panel whose columns are the operands; K0: first column block to update (J+1 for the plain trailing update,
// J+2 when block J+1 is handled by the next panel's prologue); idx: tile number.  Threads >= 256 only keep the barrier.
template <typename T>
__device__ __forceinline__ void chol_trail_body(T (&As)[NB][NB + 1], T (&Bs)[NB][NB + 1], T *__restrict__ S, int lds, T *__restrict__ W, int ldw,
                                                int J, int K0, int nrb, int nW, int idx,
                                                const void *__restrict__ Wp = nullptr, int nst_total = 0, const void *__restrict__ Sp = nullptr, int sp_stride = 0)
{
    using M = Mfma<T>;
    constexpr int NBLK = 32 / M::BLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = tid < 256;
    const int nK = nrb - K0;
    const int nSt = nK * (nK + 1) / 2;
    bool isW;
    int rb = 0, K, c0 = 0;
    if (idx < nSt) {
        int bb = 0;
        while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
        int kk = idx - bb * (bb + 1) / 2;
        rb = K0 + bb; K = K0 + kk; isW = false;
    } else {
        int t = idx - nSt;
        c0 = (t % nW) * NB; K = K0 + t / nW; isW = true;
    }
    const int w0 = (wave >> 1) * 32, w1 = (wave & 1) * 32;
    if constexpr (sizeof(T) == 4) {
        if (Sp != nullptr) {
            if (active) chol_trail_b3_tile(S, lds, W, ldw, J, isW, rb, K, c0, Wp, nst_total, Sp, sp_stride);
            return;
        }
    }
    if (active) {
        // all global loads (both operand tiles and the 16 output elements this lane will update) are issued
        // before the first LDS store, so their latencies overlap
        T ga[16], gb[16];
        const int lr = tid >> 6, lc = tid & 63;
#pragma unroll
        for (int t = 0; t < 16; ++t) gb[t] = S[(size_t)(K * NB + lr + 4 * t) * lds + J * NB + lc];
        if (!isW) {
#pragma unroll
            for (int t = 0; t < 16; ++t) ga[t] = S[(size_t)(rb * NB + lr + 4 * t) * lds + J * NB + lc];
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) ga[t] = W[(size_t)(J * NB + lr + 4 * t) * ldw + c0 + lc];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) { Bs[lr + 4 * t][lc] = gb[t]; As[lr + 4 * t][lc] = ga[t]; }   // As: [i][a] (S) or [a][i] (W)
    }
    T cv[NBLK][NBLK][M::NREG];
    if (active) {
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) {
                    const int r_ = w0 + p * M::BLK + M::row(lane, e), c_ = w1 + q * M::BLK + M::col(lane);
                    cv[p][q][e] = !isW ? S[(size_t)(rb * NB + r_) * lds + K * NB + c_] : W[(size_t)(K * NB + r_) * ldw + c0 + c_];
                }
    }
    __syncthreads();
    if (!active) return;
    typename M::acc_t acc[NBLK][NBLK];
#pragma unroll
    for (int p = 0; p < NBLK; ++p)
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
            for (int e = 0; e < M::NREG; ++e) acc[p][q][e] = (T)0;
    if (!isW) {
        // acc rows -> i (A tile rows, offset w0), acc cols (lanes) -> j (B tile rows, offset w1)
#pragma unroll 4
        for (int k0 = 0; k0 < NB; k0 += M::KS) {
            const int k = k0 + M::kk(lane);
            T av[NBLK], bv[NBLK];
#pragma unroll
            for (int p = 0; p < NBLK; ++p) { av[p] = As[w0 + p * M::BLK + M::col(lane)][k]; bv[p] = Bs[w1 + p * M::BLK + M::col(lane)][k]; }
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int q = 0; q < NBLK; ++q) M::mma(av[p], bv[q], acc[p][q]);
        }
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) {
                    int i = w0 + p * M::BLK + M::row(lane, e), j = w1 + q * M::BLK + M::col(lane);
                    S[(size_t)(rb * NB + i) * lds + K * NB + j] = cv[p][q][e] - acc[p][q][e];
                }
    } else {
        // acc rows -> j (B tile rows, offset w0), acc cols (lanes) -> i (W columns, offset w1)
#pragma unroll 4
        for (int k0 = 0; k0 < NB; k0 += M::KS) {
            const int k = k0 + M::kk(lane);
            T av[NBLK], bv[NBLK];
#pragma unroll
            for (int p = 0; p < NBLK; ++p) { av[p] = Bs[w0 + p * M::BLK + M::col(lane)][k]; bv[p] = As[k][w1 + p * M::BLK + M::col(lane)]; }
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int q = 0; q < NBLK; ++q) M::mma(av[p], bv[q], acc[p][q]);
        }
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) {
                    int j = w0 + p * M::BLK + M::row(lane, e), i = w1 + q * M::BLK + M::col(lane);
                    W[(size_t)(K * NB + j) * ldw + c0 + i] = cv[p][q][e] - acc[p][q][e];
                }
    }
}


// One launch per panel (lookahead form): workgroups [0, nP) factor panel J -- first applying the K = J column of panel
// J-1's trailing update to their own blocks (PRO) -- while workgroups [nP, ...) apply the rest of panel J-1's trailing
// update (column blocks >= J+1), which nothing in this launch reads.  The dependent chain is then 1 launch per panel
// instead of 2, and the wide update runs in the shadow of the (latency-bound) panel.
template <typename T, bool EARLY = false>
__global__ __launch_bounds__(CH_NTH) void k_chol_step(T *__restrict__ S, int lds, T *__restrict__ W, int ldw, int J, int nrb, int nW,
                                                   int nP, int32_t *__restrict__ status, unsigned int *__restrict__ arrive, unsigned int target,
                                                   int nPT, void *__restrict__ Wp, int nst_total, int ncb, int ld_split, void *__restrict__ Sp, int sp_stride,
                                                   const int32_t *__restrict__ n_dev, int nS_max, int rows_last = 0, int npend = 1)
{
    __shared__ ChSmem<T> sm;
    int b = blockIdx.x;
    if (n_dev != nullptr) {
        // Panel 0 launched BEFORE the host knows the number of rows (LI update of a step: the count is still on its way through the mailbox):
        // the grid is sized for all measurements (nS_max S row blocks), the row count is read on the device (as k_gather_li in front of it
        // does), and the workgroups of S row blocks that do not exist arrive and leave.  The host polls the count while this launch runs and
        // sizes the launches of the panels >= 1 exactly; without it the stream sat idle for the mailbox round trip + a launch between the
        // gather and the first panel.  (Same kernel, same code as every other panel: the chain's ~35 KB of instructions are fetched cold once.)
        const int n = *n_dev;
        nrb = (2 * n + NB - 1) / NB; lds = nrb * NB;
        const int nS = nrb - 1;
        if (n <= 0 || (b > nS && b <= nS_max)) {      // no update at all / a row block beyond the selected rows: keep the arrival count the host assumed
            if (b != 0 && threadIdx.x == 0) atomicAdd(arrive, 1u);
            return;
        }
        if (b > nS_max) b -= nS_max - nS;             // the W strips follow the S row blocks that exist
        nP = nPT = 1 + nS + nW;
    }
#ifdef PRE3_PROBE
    // wall-clock (100 MHz) begin / end of every workgroup of the launch of panel 2: which workgroups are the launch's long pole
    struct RtStamp { int b, on; __device__ RtStamp(int b_, int on_) : b(b_), on(on_) { stamp(0); }
                     __device__ void stamp(int j) { __builtin_amdgcn_sched_barrier(0); if (on && threadIdx.x == 0 && b < 1024) { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); g_k9rt[b * 4 + j] = t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); g_k9rt[b * 4 + 2 + j] = t; } __builtin_amdgcn_sched_barrier(0); }
                     __device__ ~RtStamp() { stamp(1); } } rt_stamp(b, J == 2 && g_rt_on);
#endif
    if (b >= nPT) {
        // riders (fp32, k_downdate_b3 in use): row block J-1 of W is final since the previous launch; its bf16 planes are produced here,
        // in the shadow of the panel, so that only the last row block is left for the split launch in front of the down-date
        if constexpr (sizeof(T) == 4) {
            const int idx = b - nPT;
            if (threadIdx.x < 256) b3_split_block(W, ldw, static_cast<bf16x8_t *>(Wp), nst_total, idx % ncb, 4 * (J - 1) + idx / ncb, threadIdx.x);
        }
        return;
    }
    if (b < nP) {
        chol_panel_body<T, true, EARLY>(sm, S, lds, W, ldw, J, nrb, status, b, arrive, target, Wp, nst_total, ld_split, Sp, sp_stride, rows_last, npend);     // (J == 0: no pending update, skipped at run time)
    } else {
        chol_trail_body<T>(sm.As, sm.Bs, S, lds, W, ldw, J - 1, J + 1, nrb, nW, b - nP, Wp, nst_total, Sp, sp_stride);
    }
}


// ------------------------------------------------------------------------------------------------
// rescue_hi_inliers.m:44-47 + ekf_update_hi_inliers.m:45-58 up to the down-date, for at most 32 rescued landmarks (64 rows), in ONE launch
// that reads nothing from the host: every workgroup (0: the diagonal block; 1..: the 64-column strips of [HP | nu]) collects the HI list from
// the gate's flags, builds the measurement rows, multiplies out its own block of H*P and the whole of S = H*P*H' + I in LDS, and runs the
// one-panel chain on them (chol_panel_body, PREBUILT).  Replaces k_collect_hi -> [host poll] -> k_ell_HP_build -> k_ell_G -> k_chol_step.
// The arithmetic is theirs term for term (H*P: the 13-term fma chain of k_ell_HP_build; S: k_ell_G's sum over a row's non-zeros of H*P at
// those columns), so the factor, W and its planes are the same bits.  Workgroup 0 also does k_collect_hi's bookkeeping (hi_meas, sel_rows,
// stats[5], the mailbox) and the rows for later readers.  stats[8] <- 1 iff the update was done here (1 <= count <= 32): the speculative
// down-date behind this launch runs on that word; count == 0: nothing to do, the next prediction's Jnorm pass gets the identity;
// count > 32: the host takes the general path once it has polled the count.
// ------------------------------------------------------------------------------------------------
constexpr int HF_MAXL = 32, HF_MAXL2 = 2 * HF_MAXL, HF_NU = 7 + 6 * HF_MAXL, HF_TS = HF_NU + 2;
struct HiFused {
    int m; const int32_t *meas, *lm_ic, *lm_li, *lm_hi, *lm_type, *lm_off; const double *Hc, *Hl, *z, *h;
    int32_t *hi_meas, *sel_rows, *stats, *mail; int seq;
    int32_t *row_col; float *row_val; double *row_nu;
    const float *P; int ld; float *S; float *W; int ldw; void *Wp; int nst_total; void *Sp; int sp_stride;
    double *params;
    int max_l;                                    // landmarks this launch may update with: 64 (two panels) when the context's row capacity holds them, else 32
    int deal_min = 1 << 30;                       // one panel: S dealt from this many landmarks on
    unsigned long long *sx = nullptr;             // two panels: S dealt over the workgroups -- [128][128] (sequence number, value) pairs (hf_S_dealt); nullptr: every workgroup builds all of S
    int n = 0; double *x = nullptr; unsigned *xflag = nullptr;      // PRE3_OPT_PEND_HI (xflag != nullptr): no launch follows this one -- the strips finish the state themselves (hf_x_update)
};
struct HfSmem {
    int list[HF_MAXL2]; int cnt;
    int rc[2 * NB][13]; float rv[2 * NB][13]; float nu[2 * NB];
    int ucol[7 + 6 * HF_MAXL2];
    union {
        float T[NB][HF_TS];                       // T[a][k] = (H*P)(row0 + a, column k of the block's list): the columns S needs
        struct { float S11[NB][NB + 1], H0[NB][NB + 1], H1[NB][NB + 1]; } two;       // two panels: the blocks that wait for the second chain
    };
};

// T <- (H*P)(rows row0 .. row0+nrow-1, [pose columns 0..6 | the columns of landmarks lm0 .. lm0+nlm-1 of the list]).  The two image rows of a
// landmark read the same 13 rows of P: one gather feeds both fma chains (each chain k_ell_HP_build's, term for term).
// TRI (a diagonal block of S: rows row0.. are the landmarks lm0.. themselves): S is built on and below the diagonal only, which reads T(row, .) at the
// pose columns and at the columns of landmarks up to the row's own -- the rest of T is neither computed nor read.
template <bool TRI>
__device__ __forceinline__ void hf_T_block(HfSmem &hf, const HiFused &a, const int row0, const int nrow, const int lm0, const int nlm)
{
    const int nU = 7 + 6 * nlm, np = nrow >> 1;
    // A thread keeps ONE column k of T and walks down the row pairs that read it (TRI: the pairs from the column's own landmark on): the seven pose
    // rows of P at that column are the same for every pair and are read once, which leaves six scattered reads per item instead of thirteen.
    // (What bounds this loop is the rate at which a CU takes scattered 4-byte reads -- every workgroup of the launch reads the same lines of P --,
    //  not their latency: round 5's first form, one (pair, column) item per thread and iteration with all thirteen reads, 26.5 us for the launch at
    //  32 landmarks; four items in flight per lane 28.6 us; reading P(col, rc[t]) instead -- P is symmetric to the bit here -- 33.8 us: the lanes
    //  of a wave then read 64 different rows, where consecutive columns of ONE row share lines.  With a thread's pairs batched -- four, or all
    //  eleven, pairs' reads in flight at once -- T took 5.4 / 6.2 us against 5.8: not latency either.  What it is: fifty workgroups ask for the
    //  same ~8 k lines of P at the same time, ~3 k line requests per L2 channel.)  The fma chains are k_ell_HP_build's, term for term, whatever
    //  the order the values were fetched in.
    const int G = nU >= CH_NTH ? 1 : CH_NTH / nU;             // row-pair classes: thread (k, g) takes the pairs p = g (mod G)
    for (int idx = threadIdx.x; idx < nU * G; idx += CH_NTH) {
        const int g = idx / nU, k = idx - g * nU;
        const int jk = k < 7 ? 0 : (k - 7) / 6;                // TRI: the first pair that reads this column
        const float *pc = a.P + (k < 7 ? k : hf.ucol[7 + 6 * lm0 + k - 7]);
        float pose[7];
        if (g < np) {
#pragma unroll
            for (int t = 0; t < 7; ++t) pose[t] = pc[(size_t)t * a.ld];      // (a row's first seven entries are the pose columns 0..6 themselves)
        }
        for (int p = g; p < np; p += G) {
            if (TRI && p < jk) continue;
            const int row = row0 + 2 * p;
            float lv[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) lv[t] = pc[(size_t)hf.rc[row][7 + t] * a.ld];
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int t = 0; t < 13; ++t) {
                const float pv = t < 7 ? pose[t] : lv[t - 7];
                s0 = fmaf(hf.rv[row][t], pv, s0); s1 = fmaf(hf.rv[row + 1][t], pv, s1);
            }
            hf.T[2 * p][k] = s0; hf.T[2 * p + 1][k] = s1;
        }
    }
}
// (H*P*H')(ra, rb) out of T (k_ell_G: the sum over the non-zeros of row rb of H*P(ra, .) there); T holds rows row0.. and the landmarks of rows col0..
__device__ __forceinline__ float hf_S_entry(const HfSmem &hf, const int ra, const int rb, const int row0, const int col0)
{
    const int kb = 7 + 6 * ((rb - col0) >> 1);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 13; ++t) s = fmaf(hf.rv[rb][t], hf.T[ra - row0][t < 7 ? t : kb + t - 7], s);
    return s;
}
// rows row0 .. row0+63 of this workgroup's 64 columns of [H*P | nu] (k_ell_HP_build's 13-term chain; rows >= r are zero)
template <typename XT>
__device__ __forceinline__ void hf_own_block(const HfSmem &hf, const HiFused &a, XT &X, const int row0, const int r, const int c0)
{
    // (a thread's column is the same in every iteration -- CH_NTH is a multiple of 64 --: the pose rows of P at that column are read once)
    static_assert(CH_NTH % 64 == 0, "hf_own_block: one column per thread");
    const int i = threadIdx.x & 63, j = c0 + i;
    const float *pc = a.P + j;
    float pose[7];
    if (j < a.ld && row0 + 2 * (int)(threadIdx.x >> 6) < r) {
#pragma unroll
        for (int t = 0; t < 7; ++t) pose[t] = pc[(size_t)t * a.ld];
    }
    for (int idx = threadIdx.x; idx < NB * NB / 2; idx += CH_NTH) {
        const int p = idx >> 6, row = row0 + 2 * p;
        float s0 = 0.f, s1 = 0.f;
        if (row < r) {
            if (j < a.ld) {
                float lv[6];
#pragma unroll
                for (int t = 0; t < 6; ++t) lv[t] = pc[(size_t)hf.rc[row][7 + t] * a.ld];
#pragma unroll
                for (int t = 0; t < 13; ++t) {
                    const float pv = t < 7 ? pose[t] : lv[t - 7];
                    s0 = fmaf(hf.rv[row][t], pv, s0); s1 = fmaf(hf.rv[row + 1][t], pv, s1);
                }
            } else if (j == a.ld) { s0 = hf.nu[row]; s1 = hf.nu[row + 1]; }
        }
        X[2 * p][i] = s0; X[2 * p + 1][i] = s1;
    }
}
// one 32x32 tile of a 64-deep product on the f32 matrix cores (an exact fma chain over k): acc(r, c) += sum_k A(r, k) B(c, k); this lane feeds
// row / column (lane & 31) and k = k0 + (lane >> 5); the accumulator layout is the chain's (Mfma<float>::row / col)
template <typename FA, typename FB>
__device__ __forceinline__ void hf_mma64(f32x16_t &acc, FA &&A, FB &&B, const int lane)
{
    const int cl = lane & 31, kh = lane >> 5;
#pragma unroll 8
    for (int k0 = 0; k0 < NB; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A(cl, k0 + kh), B(cl, k0 + kh), acc, 0, 0, 0);
}

#ifdef PRE3_PROBE
static __device__ unsigned long long g_hf[16];                  // wall-clock stamps (100 MHz) of workgroup 5 of the last k_hi_fused launch (tools/probe_hi_fused.py)
#define HF_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 5) g_hf[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HF_STAMP(k)
#endif
// Two panels (33 .. 64 rescued landmarks): S = H*P*H' + I dealt over the launch's workgroups instead of built by every one of them (three blocks of T,
// 24 us at 64 landmarks: fifty workgroups asking for the same lines of P).  Workgroup b takes the row pairs p = b, b + G, .. of the list: T(rows of p,
// [pose | landmarks 0 .. p]) -- one column per thread, thirteen reads --, then S(ra, rb) for rb <= ra by hf_S_entry's chain, out as write-through
// (sequence number, value) pairs; then everybody collects the lower triangle with sc1 loads, re-reading what is not there yet (no fences, no counters:
// round 5's form of this, with a release / acquire per workgroup, cost 10-11 us).  A small map has fewer workgroups than pairs: they take several, one after the other.  The same fma chains as hf_T_block / hf_S_entry: the same bits.
// Leaves Ls = S00, Bs = S10 (padding rows zero), hf.two.S11 = S11 (identity padding), as the redundant form does.
__device__ __forceinline__ void hf_S_dealt(HfSmem &hf, ChSmem<float> &sm, const HiFused &a, const int b, const int cnt, int32_t *status_wait, const bool two = true)
{
    const int tid = threadIdx.x, r = 2 * cnt, G = gridDim.x;
    constexpr int TW = 7 + 6 * HF_MAXL2 + 1;                  // 392
    float (*T2)[TW] = reinterpret_cast<float (*)[TW]>(&hf.T[0][0]);      // [2][392] inside T's 64 x 202 floats
    static_assert(2 * TW <= NB * HF_TS, "hf_S_dealt: T2 fits T");
    for (int p = b; p < cnt; p += G) {                        // (block-uniform: the barriers inside are reached by the whole workgroup)
        const int nU = 7 + 6 * (p + 1), row = 2 * p;
        for (int k = tid; k < nU; k += CH_NTH) {
            const float *pc = a.P + (k < 7 ? k : hf.ucol[k]);
            float pv[13];
#pragma unroll
            for (int t = 0; t < 7; ++t) pv[t] = pc[(size_t)t * a.ld];
#pragma unroll
            for (int t = 0; t < 6; ++t) pv[7 + t] = pc[(size_t)hf.rc[row][7 + t] * a.ld];
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int t = 0; t < 13; ++t) { s0 = fmaf(hf.rv[row][t], pv[t], s0); s1 = fmaf(hf.rv[row + 1][t], pv[t], s1); }
            T2[0][k] = s0; T2[1][k] = s1;
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * (2 * p + 2); idx += CH_NTH) {
            const int rr = idx / (2 * p + 2), rb = idx - rr * (2 * p + 2), ra = 2 * p + rr;
            if (rb > ra) continue;
            const int kb = 7 + 6 * (rb >> 1);
            float sv = 0.f;
#pragma unroll
            for (int t = 0; t < 13; ++t) sv = fmaf(hf.rv[rb][t], T2[rr][t < 7 ? t : kb + t - 7], sv);
            if (ra == rb) sv += 1.f;
            const unsigned long long pr = ((unsigned long long)(unsigned)a.seq << 32) | (unsigned long long)__float_as_uint(sv);
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(a.sx + (size_t)ra * (2 * NB) + rb), "v"(pr) : "memory");
        }
        __syncthreads();                                       // (T2 is rewritten by the next pair)
    }
    __syncthreads();                                           // (T2 is dead: S11 takes its place)
    // collect: the lower triangle of the r real rows; four pairs in flight per thread and pass
    const int ntri = r * (r + 1) / 2;
    bool gave_up = false;
    for (int e0 = tid * 4; e0 < ntri; e0 += CH_NTH * 4) {
        int ra[4], rb[4]; unsigned long long pr[4]; bool have[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u < ntri ? e0 + u : ntri - 1;
            int row_ = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
            while (row_ * (row_ + 1) / 2 > e) --row_;
            while ((row_ + 1) * (row_ + 2) / 2 <= e) ++row_;
            ra[u] = row_; rb[u] = e - row_ * (row_ + 1) / 2; have[u] = false; pr[u] = 0;
        }
        for (int spin = 0; spin < (1 << 20); ++spin) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (!have[u]) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(pr[u]) : "v"(a.sx + (size_t)ra[u] * (2 * NB) + rb[u]) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool all = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) { have[u] = have[u] || (unsigned)(pr[u] >> 32) == (unsigned)a.seq; all = all && have[u]; }
            if (all) break;
            if (spin == (1 << 20) - 1) gave_up = true;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e0 + u >= ntri) continue;
            const float v = __uint_as_float((unsigned)pr[u]);
            if (ra[u] < NB) sm.Ls[ra[u]][rb[u]] = v;
            else if (rb[u] < NB) sm.Bs[ra[u] - NB][rb[u]] = v;
            else hf.two.S11[ra[u] - NB][rb[u] - NB] = v;
        }
    }
    if (gave_up) atomicExch(status_wait, 1);
    // what the triangle does not hold: zeros above the diagonals, S10's padding rows, S11's (one panel: S00's) identity padding
    for (int idx = tid; idx < NB * NB; idx += CH_NTH) {
        const int i = idx >> 6, j = idx & 63;
        if (two) {
            if (j > i) { sm.Ls[i][j] = 0.f; hf.two.S11[i][j] = 0.f; }
            if (NB + i >= r) { sm.Bs[i][j] = 0.f; if (j <= i) hf.two.S11[i][j] = i == j ? 1.f : 0.f; }
        } else {
            if (j > i) sm.Ls[i][j] = 0.f;
            else if (i >= r) sm.Ls[i][j] = i == j ? 1.f : 0.f;
        }
    }
    __syncthreads();
}

// PRE3_OPT_PEND_HI: x_k_k <- x_k_k + W~'(L^-1 nu) and update.m:42-46's normalisation (params[16..], [96..]) at the end of k_hi_fused itself -- the down-date
// launch that used to carry this as its riders (update_x_block) is not sent.  Strip b owns the columns 64 (b - 1) ..: the entries of x it updates; L^-1 nu is
// column ld, the last strip's, which raises a flag behind its stores (all workgroups of the launch are resident: 50 of them, one per CU).  The sums are
// update_x_block's, term for term (sixteen chains over the rows a = g mod 16, then the chains in order, the prior last).
__device__ __forceinline__ void hf_x_update(const HiFused &a, const int b, const int r, const bool two, ChSmem<float> &sm, HfSmem &hf)
{
    // No release / acquire fences (an agent-scope release writes the whole L2 back: 18 us per launch when every workgroup did one) and one hop only: the
    // last strip sends L^-1 nu as write-through (sequence number, value) pairs, one 8-byte store per row; a reader polls the pair it needs with sc1
    // loads until the number is this launch's.  A workgroup's own rows of W~ are still in LDS (what chol_store_w_strip has just read: Xs, and H0 for
    // the first of two panels).
    const int tid = threadIdx.x, b_nu = 1 + a.ld / NB;
    auto wrow = [&](const int k, const int ci) -> float { return two ? (k < NB ? hf.two.H0[k][ci] : sm.Xs[k - NB][ci]) : sm.Xs[k][ci]; };
    unsigned long long *ybuf = reinterpret_cast<unsigned long long *>(a.xflag);          // [128] pairs
    if (b == b_nu) {
        for (int k = tid; k < r; k += CH_NTH) {
            const unsigned long long pr = ((unsigned long long)(unsigned)a.seq << 32) | (unsigned long long)__float_as_uint(wrow(k, 0));
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(ybuf + k), "v"(pr) : "memory");
        }
        return;                                                // (column block ld / 64 holds no entry of the state)
    }
    const int blk = b - 1;
    if (b < 1 || blk * NB >= a.n) return;                      // workgroup 0 (no strip) and the strips behind the state's last entry
    double *scratch = reinterpret_cast<double *>(&sm.Ls[0][0]);                         // (the factor is out; >= 16 * 64 + 4 + 128 doubles)
    double (*red)[64] = reinterpret_cast<double (*)[64]>(scratch);
    double *q = scratch + 16 * 64, *ys = scratch + 16 * 64 + 4;
    __syncthreads();                                           // (everybody is done with Ls)
    for (int k = tid; k < r; k += CH_NTH) {
        unsigned long long pr = 0;
        bool gave_up = true;
        for (int spin = 0; spin < (1 << 20); ++spin) {
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pr) : "v"(ybuf + k) : "memory");
            if ((unsigned)(pr >> 32) == (unsigned)a.seq) { gave_up = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (gave_up) atomicExch(a.stats + 7, 1);
        ys[k] = (double)__uint_as_float((unsigned)pr);
    }
    __syncthreads();
    const int ci = tid & 63, rg = tid >> 6;
    const int i = blk * 64 + ci;
    if (rg < 4) {
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
            double sc = 0;
            for (int k = rg + 4 * u; k < r; k += 16) sc = fma((double)wrow(k, ci), ys[k], sc);
            red[rg + 4 * u][ci] = sc;
        }
    }
    __syncthreads();
    double sx = 0;
    if (rg == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) sx += red[g][ci];
        if (i < a.n) sx += a.x[i];
    }
    if (blk == 0) {
        if (rg == 0 && i >= 3 && i < 7) q[i - 3] = sx;
        __syncthreads();
        if (rg == 0 && i == 0) { double Jn[16]; d_normjac(q, Jn); for (int t = 0; t < 16; ++t) { a.params[16 + t] = Jn[t]; a.params[96 + t] = Jn[t]; } }
        if (rg == 0 && i >= 3 && i < 7) sx = sx / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    }
    if (rg == 0 && i < a.n) a.x[i] = sx;
}

__global__ __launch_bounds__(CH_NTH) void k_hi_fused(HiFused a)
{
    HF_STAMP(0);
    __shared__ ChSmem<float> sm;
    __shared__ HfSmem hf;
    const int tid = threadIdx.x, b = blockIdx.x;
    // ---- the HI list (rescue_hi_inliers.m:44-46: ic && !li && hi), in measurement order, and the rows of its landmarks (k_build_rows / k_ell_HP_build:
    //      row 2s+c of HI measurement s = [Hc(c,:) | Hl(c,:)] at columns [0..6 | off..off+d-1], nu = z - h).  A lane takes one measurement: its
    //      landmark's flags are one load level behind meas[j], and a rescued landmark's Hc / Hl / z / h / type / off are fetched on the SAME level, before
    //      the ballot prefix has told the lane its position in the list (round 5: the rows used to be two more dependent load levels behind the list)
    __shared__ int s_wc[CH_NTH / 64];
    const int lane = tid & 63, wv = tid >> 6;
    int cnt = 0;
    for (int j0 = 0; j0 < a.m; j0 += CH_NTH) {
        const int j = j0 + tid;
        int in = 0, i = 0;
        if (j < a.m) {
            i = a.meas[j];
            const int f_ic = a.lm_ic[i], f_li = a.lm_li[i], f_hi = a.lm_hi[i];
            in = (f_ic == 1 && f_li == 0) ? f_hi : 0;
            if (b == 0) a.hi_meas[j] = in;
        }
        double hc[14], hl[12], zz[2] = { 0, 0 }, hh[2] = { 0, 0 };
        int ty = 0, of = 0;
        if (in) {
#pragma unroll
            for (int t = 0; t < 14; ++t) hc[t] = a.Hc[14 * i + t];
#pragma unroll
            for (int t = 0; t < 12; ++t) hl[t] = a.Hl[12 * i + t];
            zz[0] = a.z[2 * i]; zz[1] = a.z[2 * i + 1]; hh[0] = a.h[2 * i]; hh[1] = a.h[2 * i + 1];
            ty = a.lm_type[i]; of = a.lm_off[i];
        }
        const unsigned long long mask = __ballot(in != 0);
        if (lane == 0) s_wc[wv] = __popcll(mask);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < CH_NTH / 64; ++w) { const int cw = s_wc[w]; if (w < wv) woff += cw; tot += cw; }
        if (in) {
            const int pos = cnt + woff + __popcll(mask & ((1ull << lane) - 1ull));
            if (b == 0) a.sel_rows[pos] = j;
            if (pos < HF_MAXL2) {
                hf.list[pos] = j;
                const int d = ty == PRE3_INVDEPTH ? 6 : 3;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = 2 * pos + c;
#pragma unroll
                    for (int t = 0; t < 7; ++t) { hf.rc[row][t] = t; hf.rv[row][t] = (float)hc[7 * c + t]; }
#pragma unroll
                    for (int t = 0; t < 6; ++t) { hf.rc[row][7 + t] = t < d ? of + t : 0; hf.rv[row][7 + t] = t < d ? (float)hl[6 * c + t] : 0.f; }
                    const double nu = zz[c] - hh[c];
                    hf.nu[row] = (float)nu;
                    if (b == 0) a.row_nu[row] = nu;         // (for whoever reads the rows behind this launch; the general path rebuilds them)
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) hf.ucol[7 + 6 * pos + t] = t < d ? of + t : 0;
            }
        }
        cnt += tot;
        __syncthreads();                                   // (s_wc is rewritten by the next pass)
    }
    HF_STAMP(1);
    const int r = 2 * cnt;
    const bool here = cnt >= 1 && cnt <= a.max_l;
    if (b == 0 && tid == 0) {
        a.stats[5] = cnt; a.stats[8] = here ? 1 : 0;
        if (cnt == 0) for (int t = 0; t < 16; ++t) a.params[96 + t] = (t % 5 == 0) ? 1.0 : 0.0;      // no update: the pending Jnorm pass is the identity
        a.mail[5] = cnt;
        a.mail[6] = a.stats[6]; a.mail[7] = a.stats[7];       // (the device's error words, as k_collect_hi publishes them)
        __threadfence_system();
        __hip_atomic_store(&a.mail[9], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!here) return;
    const bool two = cnt > HF_MAXL;                    // 33 .. 64 landmarks: two panels
    const int r_pad = two ? 2 * NB : NB;
    // the padding rows of the last panel: zero rows of H against identity rows of S
    if (tid < r_pad) {
        const int row = tid;
        if (row >= r) {
#pragma unroll
            for (int t = 0; t < 13; ++t) { hf.rc[row][t] = 0; hf.rv[row][t] = 0.f; }
            hf.nu[row] = 0.f;
            if (b == 0) a.row_nu[row] = 0;
        }
        if (row < 7) hf.ucol[row] = row;
    }
    __syncthreads();
    if (b == 0) {                                  // the rows in their ELL form for whoever reads them after this launch
        for (int idx = tid; idx < r_pad * ELLW; idx += CH_NTH) {
            const int row = idx >> 4, t = idx & 15;
            a.row_col[idx] = t < 13 ? hf.rc[row][t] : 0; a.row_val[idx] = t < 13 ? hf.rv[row][t] : 0.f;
        }
    }
    if (!two) {
        // ---- T = (H*P) at the columns of the selected rows, and this workgroup's own block of [H*P | nu]
        HF_STAMP(2);
        if (a.sx != nullptr && cnt >= a.deal_min) {
            // (round 6) S dealt over the workgroups, as in the two-panel case: from ~20 landmarks on it is cheaper than fifty copies of T
            if (b >= 1) hf_own_block(hf, a, sm.Xs, 0, r, (b - 1) * NB);
            hf_S_dealt(hf, sm, a, b, cnt, a.stats + 7, false);
            HF_STAMP(3); HF_STAMP(4);
        } else {
        hf_T_block<true>(hf, a, 0, r, 0, cnt);
        HF_STAMP(3);
        if (b >= 1) hf_own_block(hf, a, sm.Xs, 0, r, (b - 1) * NB);
        __syncthreads();
        HF_STAMP(4);
        // ---- S = H*P*H' + I on and below the diagonal, identity padding
        for (int idx = tid; idx < NB * NB; idx += CH_NTH) {
            const int ra = idx >> 6, rb = idx & 63;
            float s = 0.f;
            if (rb <= ra) {
                if (ra < r) { s = hf_S_entry(hf, ra, rb, 0, 0); if (ra == rb) s += 1.f; }
                else s = ra == rb ? 1.f : 0.f;
            }
            sm.Ls[ra][rb] = s;
        }
        __syncthreads();
        }
        HF_STAMP(5);
        chol_panel_body<float, false, true, true>(sm, a.S, NB, a.W, a.ldw, 0, 1, a.stats + 6, b, nullptr, 0u, a.Wp, a.nst_total, a.ld, a.Sp, a.sp_stride, r);
        HF_STAMP(6);
        if (a.xflag != nullptr) hf_x_update(a, b, r, false, sm, hf);
        return;
    }
    // ---- two panels (33 .. 64 landmarks), every workgroup for itself as above.  S = [S00 . ; S10 S11], [H*P | nu] = [H0 ; H1] (this workgroup's
    //      64 columns).  First chain: S00 = L00 L00' with the identity as right-hand side, which leaves M0 = L00^-1; then, on the f32 matrix cores,
    //      L10 = S10 M0', W0 = M0 H0, S11 - L10 L10', H1 - L10 W0; second chain: L11 and W1 = L11^-1 (H1 - L10 W0).  Workgroup 0 owns no strip: it
    //      keeps L in S (stride 128) and reports a non-positive pivot of either chain.
    {
        const int r1 = r - NB, c0 = (b - 1) * NB;
        const bool hasX = b >= 1;
        const int wave = tid >> 6, lane = tid & 63, wv = wave & 3, w0 = (wv >> 1) * 32, w1 = (wv & 1) * 32;
        const int lrow = 4 * (lane >> 5), lcol = lane & 31;
        auto &Ls = sm.Ls; auto &Xs = sm.Xs; auto &Bs = sm.Bs;
        if (a.sx != nullptr) hf_S_dealt(hf, sm, a, b, cnt, a.stats + 7);
        else {
            hf_T_block<true>(hf, a, 0, NB, 0, HF_MAXL);
            __syncthreads();
            for (int idx = tid; idx < NB * NB; idx += CH_NTH) {
                const int ra = idx >> 6, rb = idx & 63;
                float s = 0.f;
                if (rb <= ra) { s = hf_S_entry(hf, ra, rb, 0, 0); if (ra == rb) s += 1.f; }
                Ls[ra][rb] = s;
            }
            __syncthreads();
            hf_T_block<false>(hf, a, NB, r1, 0, HF_MAXL);
            __syncthreads();
            for (int idx = tid; idx < NB * NB; idx += CH_NTH) {            // Bs[i][k] = S10(i, k); padding rows are zero
                const int i = idx >> 6, rb = idx & 63;
                Bs[i][rb] = NB + i < r ? hf_S_entry(hf, NB + i, rb, NB, 0) : 0.f;
            }
            __syncthreads();
            hf_T_block<true>(hf, a, NB, r1, HF_MAXL, cnt - HF_MAXL);
            __syncthreads();
            float s11[(NB * NB + CH_NTH - 1) / CH_NTH];                    // (S11 takes T's place: through registers, across a barrier)
    #pragma unroll
            for (int q = 0; q < (NB * NB + CH_NTH - 1) / CH_NTH; ++q) {
                const int idx = tid + q * CH_NTH, i = (idx >> 6) & 63, j = idx & 63;
                float s = 0.f;
                if (j <= i) {
                    if (NB + i < r) { s = hf_S_entry(hf, NB + i, NB + j, NB, NB); if (i == j) s += 1.f; }
                    else s = i == j ? 1.f : 0.f;
                }
                s11[q] = s;
            }
            __syncthreads();
    #pragma unroll
            for (int q = 0; q < (NB * NB + CH_NTH - 1) / CH_NTH; ++q) {
                const int idx = tid + q * CH_NTH;
                if (idx < NB * NB) hf.two.S11[idx >> 6][idx & 63] = s11[q];
            }
        }
        if (hasX) { hf_own_block(hf, a, hf.two.H0, 0, r, c0); hf_own_block(hf, a, hf.two.H1, NB, r, c0); }
        for (int idx = tid; idx < NB * NB; idx += CH_NTH) Xs[idx >> 6][idx & 63] = (idx >> 6) == (idx & 63) ? 1.f : 0.f;
        __syncthreads();
        {
            typename ChW<float>::acc_t acc[ChW<float>::NBLK][ChW<float>::NBLK];
            bool bad = false;
            if constexpr (PRE3_CHAIN_ASYNC) chol_chain_async<float, false, true>(sm, acc, false, true, bad);       // (X = I: its tile above the diagonal stays zero)
            else chol_chain<float, false, false>(sm, acc, false, true, bad, [](int) {});
            if (bad && (wave == 8 || wave == 9) && lane == 0 && b == 0) atomicExch(a.stats + 6, 1);
        }
        // (the chain ends behind a barrier)  Ls = L00 (lower triangle), Xs[a][k] = M0(a, k)
        if (b == 0) {
            for (int idx = tid; idx < NB * NB; idx += CH_NTH) {
                const int i = idx >> 6, a2 = idx & 63;
                a.S[(size_t)i * (2 * NB) + a2] = a2 <= i ? Ls[i][a2] : 0.f;
                a.S[(size_t)i * (2 * NB) + NB + a2] = 0.f;
            }
        }
        f32x16_t pa;
#pragma unroll
        for (int e = 0; e < 16; ++e) pa[e] = 0.f;
        if (wave < 4) hf_mma64(pa, [&](int c, int k) { return Bs[w0 + c][k]; }, [&](int c, int k) { return Xs[w1 + c][k]; }, lane);                       // L10(i, a) = sum_k S10(i, k) M0(a, k)
        else if (wave < 8 && hasX) hf_mma64(pa, [&](int c, int k) { return Xs[w0 + c][k]; }, [&](int c, int k) { return hf.two.H0[k][w1 + c]; }, lane);      // W0(a, c) = sum_k M0(a, k) H0(k, c)
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int e = 0; e < 16; ++e) Bs[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol] = pa[e];
        } else if (wave < 8 && hasX) {
#pragma unroll
            for (int e = 0; e < 16; ++e) hf.two.H0[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol] = pa[e];
        }
        __syncthreads();
        // the second chain's blocks: D = S11 - L10 L10' over L00 (the tile above the diagonal is dead), X = H1 - L10 W0 over M0
        if (wave < 4) {
            if (wv != 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) pa[e] = hf.two.S11[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol];
                hf_mma64(pa, [&](int c, int k) { return -Bs[w0 + c][k]; }, [&](int c, int k) { return Bs[w1 + c][k]; }, lane);
#pragma unroll
                for (int e = 0; e < 16; ++e) Ls[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol] = pa[e];
            }
        } else if (wave < 8 && hasX) {
#pragma unroll
            for (int e = 0; e < 16; ++e) pa[e] = hf.two.H1[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol];
            hf_mma64(pa, [&](int c, int k) { return -Bs[w0 + c][k]; }, [&](int c, int k) { return hf.two.H0[k][w1 + c]; }, lane);
#pragma unroll
            for (int e = 0; e < 16; ++e) Xs[w0 + (e & 3) + 8 * (e >> 2) + lrow][w1 + lcol] = pa[e];
        }
        // row block 0 of W leaves (f32 rows + planes) while the second chain runs; workgroup 0: L10
        if (hasX) chol_store_w_strip<float>(hf.two.H0, a.W, a.ldw, 0, c0, a.Wp, a.nst_total, a.ld);
        else {
            for (int idx = tid; idx < NB * NB; idx += CH_NTH) a.S[(size_t)(NB + (idx >> 6)) * (2 * NB) + (idx & 63)] = Bs[idx >> 6][idx & 63];
        }
        __syncthreads();
        chol_panel_body<float, false, true, true>(sm, a.S, 2 * NB, a.W, a.ldw, 1, 2, a.stats + 6, b, nullptr, 0u, a.Wp, a.nst_total, a.ld, a.Sp, a.sp_stride, r1);
        if (a.xflag != nullptr) hf_x_update(a, b, r, true, sm, hf);
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
// K9 v3: symmetric, persistent.  Workgroups (2 per CU) pull 64x64 upper-triangle tiles (I <= J) from a
// monotonic global counter, so a workgroup in its load/store phase overlaps one in its MFMA phase on the
// same SIMDs and the tail is balanced at tile granularity.  4 waves per tile, each a 32x32 sub-tile (fp32: one
// v_mfma_f32_32x32x2 accumulator; fp64: 2x2 v_mfma_f64_16x16x4 accumulators).  W is staged k-major through
// LDS (BK rows x 64 contiguous columns per operand), double-buffered with register prefetch; the P tile is
// prefetched into registers before the k-loop.  Epilogue: new = P_IJ - acc goes to P_IJ and, transposed
// through a wave-private LDS patch, to P_JI: the lower triangle is never read and P is exactly symmetric.
template <typename T, int BK>
__global__ __launch_bounds__(256) void k_downdate(T *__restrict__ P, int ld, const T *__restrict__ W, int ldw, int r_pad,
                                                  const int2 *__restrict__ tiles, int tiles_stride, const int *__restrict__ tile_cnt,
                                                  unsigned int *__restrict__ ctr)
{
    using M = Mfma<T>;
    constexpr int TS = 64;
    constexpr int NBLK = 32 / M::BLK;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int ROWV = TS / VEC;
    constexpr int NLD = (BK * ROWV) / 256;
    static_assert((BK * ROWV) % 256 == 0 && NLD >= 1, "stage must divide over the workgroup");
    typedef T vec_t __attribute__((ext_vector_type(VEC)));

    __shared__ __attribute__((aligned(16))) T smem[4 * BK * TS];
    __shared__ unsigned int s_tile;
    T (*sA)[BK][TS] = reinterpret_cast<T (*)[BK][TS]>(smem);
    T (*sB)[BK][TS] = reinterpret_cast<T (*)[BK][TS]>(smem + 2 * BK * TS);
    T (*patch)[33] = reinterpret_cast<T (*)[33]>(smem + (threadIdx.x >> 6) * (32 * 33));
    static_assert(4 * 32 * 33 <= 4 * BK * TS, "patches must fit in the staging buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int nstage = r_pad / BK;

    // Tickets: 8 lists / 8 counters (one per XCD label blockIdx % 8: blocks are dealt round-robin over the XCDs,
    // so a list's tiles -- whole 4x4 super-tiles -- stay in one L2; speed only).  A workgroup drains its own
    // list only (the lists are balanced to within one tile); one contended word would serialise at ~88 tickets/us.
    auto next_ticket = [&]() -> unsigned int {
        const int x = blockIdx.x & 7;          // lists are balanced to within one tile: no stealing, one failing ticket per workgroup
        const unsigned int t = atomicAdd(&ctr[x], 1u);
        return t < (unsigned int)tile_cnt[x] ? (unsigned int)(x * tiles_stride) + t : 0xffffffffu;
    };
    vec_t ra[NLD], rb[NLD];
    auto gload = [&](int I0_, int J0_, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            int v = tid + l * 256;
            int kr = v / ROWV, cv = (v % ROWV) * VEC;
            ra[l] = *reinterpret_cast<const vec_t *>(W + (size_t)(k0 + kr) * ldw + I0_ + cv);
            rb[l] = *reinterpret_cast<const vec_t *>(W + (size_t)(k0 + kr) * ldw + J0_ + cv);
        }
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            int v = tid + l * 256;
            int kr = v / ROWV, cv = (v % ROWV) * VEC;
            *reinterpret_cast<vec_t *>(&sA[buf][kr][cv]) = ra[l];
            *reinterpret_cast<vec_t *>(&sB[buf][kr][cv]) = rb[l];
        }
    };
    // this lane's 16 entries of a P tile (accumulator layout)
    auto pload = [&](T (&dst)[NBLK][NBLK][M::NREG], int I0_, int J0_) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e)
                    dst[p][q][e] = P[(size_t)(I0_ + wi * 32 + p * M::BLK + M::row(lane, e)) * ld + J0_ + wj * 32 + q * M::BLK + M::col(lane)];
    };

#ifdef PRE3_PROBE
    if (tid == 0 && blockIdx.x == 0) { g_probe[8] = __builtin_amdgcn_s_memtime(); g_probe[9] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (tid == 0) s_tile = next_ticket();
    __syncthreads();
    unsigned int t = s_tile;
    if (t == 0xffffffffu) return;              // uniform: every wave of the workgroup leaves together
    int2 ij = tiles[t];
    T pv[NBLK][NBLK][M::NREG];
    gload(ij.x * TS, ij.y * TS, 0);
    pload(pv, ij.x * TS, ij.y * TS);
#ifdef PRE3_PROBE
    int probe_tile = 0;
#define K9_STAMP(k) do { if (tid == 0 && blockIdx.x < 64 && probe_tile < 8) g_k9[(blockIdx.x * 8 + probe_tile) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define K9_STAMP(k)
#endif
    for (;;) {
        K9_STAMP(0);
        const int I0 = ij.x * TS, J0 = ij.y * TS;
        typename M::acc_t acc[NBLK][NBLK];
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) acc[p][q][e] = (T)0;

        sstore(0);
        __syncthreads();
        K9_STAMP(1);
        for (int s = 0; s < nstage; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstage) gload(I0, J0, (s + 1) * BK);
            else if (tid == 0) s_tile = next_ticket();     // no loads behind it: its latency hides under the last stage
            // all operand fragments of the stage are read from LDS up front (registers are plentiful), so the
            // MFMAs issue back to back behind counted lgkmcnt waits instead of one LDS round trip per pair
            T av[BK / M::KS][NBLK], bv[BK / M::KS][NBLK];
#pragma unroll
            for (int ks = 0; ks < BK / M::KS; ++ks) {
                const int krow = ks * M::KS + M::kk(lane);
#pragma unroll
                for (int p = 0; p < NBLK; ++p) {
                    av[ks][p] = sA[buf][krow][wi * 32 + p * M::BLK + M::col(lane)];
                    bv[ks][p] = sB[buf][krow][wj * 32 + p * M::BLK + M::col(lane)];
                }
            }
#pragma unroll
            for (int ks = 0; ks < BK / M::KS; ++ks)
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int q = 0; q < NBLK; ++q) M::mma(av[ks][p], bv[ks][q], acc[p][q]);
            if (s + 1 < nstage) sstore(buf ^ 1);
            __syncthreads();
        }
        K9_STAMP(2);
        // ---- cross-tile prefetch: the next ticket is visible (barrier above); start its first stage and its P tile
        //      now, so their HBM/L2 latency overlaps this tile's epilogue instead of leaving the matrix cores idle
        const unsigned int tn = s_tile;
        const bool has_next = tn != 0xffffffffu;
        int2 ijn = ij;
        T pvn[NBLK][NBLK][M::NREG];
        if (has_next) {
            ijn = tiles[tn];
            gload(ijn.x * TS, ijn.y * TS, 0);
            pload(pvn, ijn.x * TS, ijn.y * TS);
        }
        // ---- epilogue (the staging buffers are dead: each wave owns a private [32][33] patch of them)
        const bool mirror = ij.x != ij.y;
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) {
                    const int lr = p * M::BLK + M::row(lane, e), lc = q * M::BLK + M::col(lane);
                    const T v = pv[p][q][e] - acc[p][q][e];
                    P[(size_t)(I0 + wi * 32 + lr) * ld + J0 + wj * 32 + lc] = v;
                    if (mirror) patch[lr][lc] = v;
                }
        if (mirror) {
            __builtin_amdgcn_s_waitcnt(0xc07f);      // wave-private patch: this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
            const int rr = lane & 31, half = lane >> 5;
#pragma unroll
            for (int cc = 0; cc < 32; cc += 2) {
                const int c = cc + half;
                P[(size_t)(J0 + wj * 32 + c) * ld + I0 + wi * 32 + rr] = patch[rr][c];
            }
        }
        K9_STAMP(3);
#ifdef PRE3_PROBE
        ++probe_tile;
        if (tid == 0 && blockIdx.x == 0) { g_probe[10] = __builtin_amdgcn_s_memtime(); g_probe[11] = __builtin_amdgcn_s_memrealtime(); }
#endif
        if (!has_next) return;
        ij = ijn;
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) pv[p][q][e] = pvn[p][q][e];
        __syncthreads();                       // every wave's patch reads are done before the buffers are restaged
    }
}

// LDS-DMA of one [BK][64] operand stage (both operands): granule v = tid + l*256 covers row kr = v / ROWV, columns
// (v % ROWV)*VEC ..; its LDS address is v*16 bytes into the stage image = wave-uniform base + lane*16, as the
// instruction requires (global_load_lds_dwordx4).
template <typename T, int BK, int NT = 256>
__device__ __forceinline__ void lds_dma_stage(const T *__restrict__ W, int ldw, int k0, int I0, int J0, T *sA, T *sB, int tid, int wave)
{
    constexpr int VEC = 16 / sizeof(T), ROWV = 64 / VEC, NLD = (BK * ROWV) / NT;
    static_assert((BK * ROWV) % NT == 0 && NLD >= 1, "stage must divide over the workgroup");
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
        const int v = tid + l * NT, kr = v / ROWV, cv = (v % ROWV) * VEC;
        const T *ga = W + (size_t)(k0 + kr) * ldw + I0 + cv;
        const T *gb = W + (size_t)(k0 + kr) * ldw + J0 + cv;
        T *la = sA + (size_t)(wave * 64 + l * NT) * VEC;            // lane 0's granule
        T *lb = sB + (size_t)(wave * 64 + l * NT) * VEC;
        __builtin_amdgcn_global_load_lds(ga, (__attribute__((address_space(3))) void *)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gb, (__attribute__((address_space(3))) void *)lb, 16, 0, 0);
    }
}

// K9, one-tile-per-workgroup form for launches whose whole tile list fits on the chip at once (n_tiles <= 5 per
// CU, e.g. n = 3013: 1176 tiles = 4.6 per CU).  Five 4-wave workgroups per CU (<= 160 KiB of LDS, <= 96 VGPRs)
// make the launch ONE round: a CU's matrix cores are shared by all of its tiles from start to end, so there is
// no second, half-empty round and load/epilogue phases of one workgroup hide behind the others' MFMAs.
// W is staged by LDS-DMA (global_load_lds_dwordx4: the [BK][64] image is lane-linear, 16 B per lane, so no
// VGPR round trip and no ds_write), double-buffered: stage s+1 is requested when stage s starts and waited for
// (vmcnt(0) + barrier) when it ends.  Same tile math and mirrored epilogue as k_downdate.
// update.m:36,42,48 for 64 states per workgroup: x_out = x_prior + W'y (y = column `ld` of W), the normalisation Jacobian
// at the un-normalised quaternion -> params[16..31], then the quaternion normalised.  Runs as extra workgroups of the K9
// launch (it only reads W), which removes one kernel boundary from the update chain.
template <typename T>
__device__ __forceinline__ void update_x_block(int blk, int n, int r, const T *__restrict__ W, int ldw, int ld,
                                               const double *__restrict__ x_prior, double *__restrict__ x_out, double *__restrict__ params,
                                               double *scratch /* >= 16*64+4 doubles of LDS: the tile path's staging buffer, no extra allocation */)
{
    // Sixteen chains, chain g summing the rows a = g (mod 16) in increasing order, then the chains in order 0 .. 15 and x_prior last: the same
    // sums, term for term, as k_update_x (pre3_geom.hip) and as the strips of the persistent factorisation (pre3_cholp.hip, x-update at their
    // end), so that x_k_k does not depend on which of the three computed it.  Four chains per thread here.
    double (*red)[64] = reinterpret_cast<double (*)[64]>(scratch);
    double *q = scratch + 16 * 64;
    const int ci = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blk * 64 + ci;                   // i < ldw always (ldw >= ld + 64 > n rounded up)
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {                  // chain rg + 4 u
        double sc = 0;
#pragma unroll 8
        for (int a = rg + 4 * u; a < r; a += 16) sc = fma((double)W[(size_t)a * ldw + i], (double)W[(size_t)a * ldw + ld], sc);
        red[rg + 4 * u][ci] = sc;
    }
    __syncthreads();
    double s = 0;
    if (rg == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][ci];
        if (i < n) s += x_prior[i];
    }
    if (blk == 0) {                 // block-uniform branch: every thread of the block reaches the barrier
        if (rg == 0 && i >= 3 && i < 7) q[i - 3] = s;
        __syncthreads();
        if (rg == 0 && i == 0) { double Jn[16]; d_normjac(q, Jn); for (int t = 0; t < 16; ++t) { params[16 + t] = Jn[t]; params[96 + t] = Jn[t]; } }
        if (rg == 0 && i >= 3 && i < 7) s = s / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    }
    if (rg == 0 && i < n) x_out[i] = s;
}

static inline int k9_write_through() { static const int v = getenv("PRE3_K9_WT") ? atoi(getenv("PRE3_K9_WT")) : 1; return v; }
struct XUpd { int n_tiles, n, r; const double *x_prior; double *x_out; double *params;
              const int32_t *gate;
              int nx, wt = 0; };        // (wt: P leaves as write-through stores, store_wt)  x-update workgroups in the launch (0: the state has been updated elsewhere -- the riders behind the tiles are all projection blocks)   // gate != nullptr (the speculative down-date behind k_hi_fused): run only if gate[8] == 1, with r = 2 * gate[5] rows

// waves_per_eu: the riders' fp64 geometry must not raise the register count of the tile path (5 workgroups per CU in fp32,
// 4 in fp64 -- every tile resident at once); if anything spills, it is the riders.
// NW = 8 (fp64 at small n, round 5): the tile's four 32 x 32 wave tiles are each held by TWO waves, one per half of every k-stage -- a CU that
// holds one tile (n = 1213: 210 tiles on 256 CUs) then has two waves per SIMD, and one's LDS reads / DMA waits hide behind the other's MFMAs
// (v_mfma_f64_16x16x4 takes 64 cycles: with one wave per SIMD the matrix pipe idled through every stage's LDS round trip and barrier).  The two
// halves meet in LDS behind the last stage, (first half) + (second half): one fixed order.
// write-through (sc0 sc1) store of one down-dated entry of P (see store_wt below: the tiles drain while the launch runs, not in its end-of-kernel release)
__device__ __forceinline__ void store1_wt(float *d, float v, bool wt)
{
    if (wt) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else *d = v;
}
__device__ __forceinline__ void store1_wt(double *d, double v, bool wt)
{
    if (wt) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else *d = v;
}
template <typename T, int BK, int NW = 4>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(5))) void k_downdate_1t(T *__restrict__ P, int ld, const T *__restrict__ W, int ldw, int r_pad,
                                                     const int2 *__restrict__ tiles, int gen_size, XUpd xu, ProjRide pr)
{
    using M = Mfma<T>;
    constexpr int TS = 64;
    constexpr int NBLK = 32 / M::BLK;
    constexpr int VEC = 16 / sizeof(T);               // elements per 16-byte LDS-DMA granule
    constexpr int ROWV = TS / VEC;                    // granules per staged row
    constexpr int NT = 64 * NW;
    constexpr int HS = BK / M::KS / 2;                // k-steps per half stage
    static_assert(NW == 4 || NW == 8, "four wave tiles, one or two waves each");
    constexpr int STG = BK * TS;                      // elements of one operand stage
    // epilogue patches: fp32 [32][33] (padded); fp64 [32][32] with the column XOR-ed by the row -- conflict-free both ways without the
    // padding, so that the patches fit the 32 KB of staging and five workgroups still share a CU's 160 KB
    constexpr int PS = sizeof(T) == 4 ? 33 : 32;
    constexpr int SMEM = 4 * STG > 4 * 32 * PS ? 4 * STG : 4 * 32 * PS;     // staging buffers, reused by the 4 epilogue patches
    __shared__ __attribute__((aligned(16))) T smem[SMEM];                   // [A0 | A1 | B0 | B1], each [BK][64]
    static_assert(sizeof(T) * SMEM >= sizeof(double) * (16 * 64 + 4), "the riders borrow the staging buffer");
    if ((int)blockIdx.x >= xu.n_tiles) {            // riders (launch_downdate adds these workgroups)
        if (NW > 4 && threadIdx.x >= 256) return;   // (the riders are four-wave workgroups)
        __builtin_amdgcn_s_setprio(3);              // short dependent chains: must not starve behind the MFMA waves sharing their SIMD
        const int nx = xu.nx, rb = blockIdx.x - xu.n_tiles;
        if (rb < nx) {                              // the state update
            update_x_block<T>(rb, xu.n, xu.r, W, ldw, ld, xu.x_prior, xu.x_out, xu.params, reinterpret_cast<double *>(smem));
            if (pr.n_blocks) ride_signal(pr.ctr);
        } else {                                    // the rescue's projection at the updated state, once x is complete
            proj_ride_block(pr, rb - nx);
        }
        return;
    }
    // Workgroups are dispatched in generations of one per CU; give each generation its own wave priority so the
    // tiles sharing a SIMD finish one after another (their epilogue/HBM phases then hide behind the next tile's
    // MFMAs) instead of all together at the end.  Priority is a speed hint only.
    {
        const int gen = blockIdx.x / gen_size;
        if (gen == 0) __builtin_amdgcn_s_setprio(3);
        else if (gen == 1) __builtin_amdgcn_s_setprio(2);
        else if (gen == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    T *patch = smem + ((threadIdx.x >> 6) & 3) * (32 * PS);
    auto pat = [&](int r, int c) -> T & { return patch[r * PS + (sizeof(T) == 4 ? c : (c ^ r))]; };
    const int2 ij = tiles[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PRE3_PROBE
#define RT_STAMP(k) do { if (tid == 0 && blockIdx.x < 2048) g_k9rt[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RT_STAMP(k)
#endif
    RT_STAMP(0);
    const int wi = (wave & 3) >> 1, wj = wave & 1, kh = wave >> 2;      // kh: the half of every stage this wave multiplies (NW == 8)
    const int I0 = ij.x * TS, J0 = ij.y * TS;
    const int nstage = r_pad / BK;
    typename M::acc_t acc[NBLK][NBLK];
#pragma unroll
    for (int p = 0; p < NBLK; ++p)
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
            for (int e = 0; e < M::NREG; ++e) acc[p][q][e] = (T)0;
    lds_dma_stage<T, BK, NT>(W, ldw, 0, I0, J0, smem, smem + 2 * STG, tid, wave);
    // this lane's 16 entries of the P tile: requested now, consumed in the epilogue (latency hidden by the k-loop)
    T pv[NBLK][NBLK][M::NREG];
    if (kh == 0) {
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e)
                    pv[p][q][e] = P[(size_t)(I0 + wi * 32 + p * M::BLK + M::row(lane, e)) * ld + J0 + wj * 32 + q * M::BLK + M::col(lane)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    RT_STAMP(1);
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage)       // buf^1 was last read in stage s-1: every wave is past that barrier
            lds_dma_stage<T, BK, NT>(W, ldw, (s + 1) * BK, I0, J0, smem + (buf ^ 1) * STG, smem + (2 + (buf ^ 1)) * STG, tid, wave);
        const T *sA = smem + buf * STG, *sB = smem + (2 + buf) * STG;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (NW == 8 && hf != kh) continue;
            T av[HS][NBLK], bv[HS][NBLK];
#pragma unroll
            for (int ks = 0; ks < HS; ++ks) {
                const int krow = (hf * HS + ks) * M::KS + M::kk(lane);
#pragma unroll
                for (int p = 0; p < NBLK; ++p) {
                    av[ks][p] = sA[krow * TS + wi * 32 + p * M::BLK + M::col(lane)];
                    bv[ks][p] = sB[krow * TS + wj * 32 + p * M::BLK + M::col(lane)];
                }
            }
#pragma unroll
            for (int ks = 0; ks < HS; ++ks)
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int q = 0; q < NBLK; ++q) M::mma(av[ks][p], bv[ks][q], acc[p][q]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA granules of stage s+1 have landed
        __syncthreads();
    }
    RT_STAMP(2);
    if (NW == 8) {
        // the second half's share goes through the wave tile's patch (the staging buffers are dead behind the last barrier) and is added in one
        // fixed order, (first half) + (second half); the epilogue is the first half's waves'
        if (kh == 1) {
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int q = 0; q < NBLK; ++q)
#pragma unroll
                    for (int e = 0; e < M::NREG; ++e) pat(p * M::BLK + M::row(lane, e), q * M::BLK + M::col(lane)) = acc[p][q][e];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int p = 0; p < NBLK; ++p)
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e) acc[p][q][e] += pat(p * M::BLK + M::row(lane, e), q * M::BLK + M::col(lane));
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();           // (this wave's reads of its patch are done before the mirror image goes through it)
    }
    const bool mirror = ij.x != ij.y;
#pragma unroll
    for (int p = 0; p < NBLK; ++p)
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
            for (int e = 0; e < M::NREG; ++e) {
                const int lr = p * M::BLK + M::row(lane, e), lc = q * M::BLK + M::col(lane);
                const size_t o = (size_t)(I0 + wi * 32 + lr) * ld + J0 + wj * 32 + lc;
                const T v = pv[p][q][e] - acc[p][q][e];
                store1_wt(P + o, v, xu.wt != 0);
                if (mirror) pat(lr, lc) = v;
            }
    if (mirror) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int rr = lane & 31, half = lane >> 5;
#pragma unroll
        for (int cc = 0; cc < 32; cc += 2) {
            const int c = cc + half;
            store1_wt(P + (size_t)(J0 + wj * 32 + c) * ld + I0 + wi * 32 + rr, pat(rr, c), xu.wt != 0);
        }
    }
    RT_STAMP(3);
}

// ------------------------------------------------------------------------------------------------
// K9 on the bf16 matrix cores (fp32 covariance path).  gfx950's f32-input MFMA runs at the vector rate, 1/16 of the
// bf16 MFMA; an fp32 value is the exact sum of three bf16 values (8 + 8 + 8 significand bits, round-to-nearest at each
// step), so W = a + b + c and  W'W = a'a + (a'b + b'a) + (a'c + c'a + b'b) + O(2^-27 |W|'|W|): six bf16 products with f32
// accumulation instead of one f32 product, 16/6 = 2.7x the f32 matrix rate at f32 accuracy (the dropped b'c, c'b, c'c
// terms are below the rounding of the f32 accumulation itself).
//   k_split_w      W (f32, k-major) -> three bf16 planes, stored in the order the tile kernel's LDS stage image has:
//                  block (cb, s) = 128 columns x 16 k = [plane 3][fragment 4][lane 64][8 bf16] (12 KB, contiguous), where
//                  lane (r = l & 31, h = l >> 5) of fragment f holds k = 16 s + 8 h + 0..7 of column 128 cb + 32 f + r:
//                  exactly the A / B operand of v_mfma_f32_32x32x16_bf16, so the tile kernel's staging is a linear
//                  LDS-DMA copy and its fragment reads are conflict-free ds_read_b128.
//   k_downdate_b3  one 128x128 tile of the upper triangle per workgroup (4 waves x 64x64), three-deep LDS-DMA ring of
//                  24 KB stages, 24 MFMAs per wave and stage; mirrored epilogue through wave-private LDS patches.
//                  Bytes per tile-stage: 24 KB for 128x128x16x6 MACs -- 64x64 tiles would need 4x the L2 bandwidth.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split_w(const float *__restrict__ W, int ldw, bf16x8_t *__restrict__ Wp, int nst_total, int st0)
{
    b3_split_block(W, ldw, Wp, nst_total, blockIdx.x, st0 + blockIdx.y, threadIdx.x);
}

// One operand of one stage into the ring.  TM = 128: the whole 768-granule block, copied linearly (3 instructions per wave).
// TM = 64: fragments 2*half, 2*half+1 of every plane (3 x 128 granules -> image [plane][fragment 2][lane]): one instruction in every
// wave + a second one in waves 0 and 1.
template <int TM>
__device__ __forceinline__ void b3_dma(const bf16x8_t *__restrict__ src, int half, bf16x8_t *dst, int tid, int wave)
{
    if (TM == 128) {
#define B3_NPL 3
#pragma unroll
        for (int l = 0; l < B3_NPL; ++l)
            __builtin_amdgcn_global_load_lds(src + l * 256 + tid, (__attribute__((address_space(3))) void *)(dst + l * 256 + wave * 64), 16, 0, 0);
    } else {
        __builtin_amdgcn_global_load_lds(src + (tid >> 7) * 256 + half * 128 + (tid & 127), (__attribute__((address_space(3))) void *)(dst + wave * 64), 16, 0, 0);
        if (wave < 2)
            __builtin_amdgcn_global_load_lds(src + 512 + half * 128 + tid, (__attribute__((address_space(3))) void *)(dst + 256 + wave * 64), 16, 0, 0);
    }
}

// wait until at most the DMA instructions of LEFT stages (this wave's share) are outstanding
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int TM, int LEFT>
__device__ __forceinline__ void b3_wait(int wave)
{
    if (TM == 128) vmwait<2 * B3_NPL * LEFT>();
    else if (wave < 2) vmwait<4 * LEFT>();
    else vmwait<2 * LEFT>();
}

// TM x TM tile at 64-column block (bi, bj): 4 waves x (TM/2 x TM/2), i.e. NB x NB accumulators of 32x32 per wave
// Write-through (sc0 sc1) stores of the down-dated P: the tiles drain to memory while the launch still runs instead of sitting dirty in the
// eight L2s until the end-of-kernel release writes them back (a ~6-us gap in front of the next launch at N = 500; pre3_cholp.hip: dd_store_wt).
typedef float store_f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float *d, float v, bool wt)
{
    if (wt) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else *d = v;
}
__device__ __forceinline__ void store_wt(store_f4_t *d, store_f4_t v, bool wt)
{
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else *d = v;
}

template <int TM>
__device__ __forceinline__ void b3_tile(float *__restrict__ P, int ld, const bf16x8_t *__restrict__ Wp, int nst_total, int nst, int bi, int bj,
                                        bf16x8_t *smem /* ring: [3][2][768 granules] */, bool wt = false)
{
    constexpr int NB = TM / 64, PL = (TM / 32) * 64;       // blocks per wave and dimension; granules per plane of an operand stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int R0 = bi * 64 + wi * (TM / 2), C0 = bj * 64 + wj * (TM / 2);
    const bool diag = bi == bj;
    const bf16x8_t *srcA = Wp + (size_t)(bi >> 1) * nst_total * B3_GRAN;
    const bf16x8_t *srcB = Wp + (size_t)(bj >> 1) * nst_total * B3_GRAN;
    const int hA = bi & 1, hB = bj & 1;
    const int lrow = 4 * (lane >> 5), lcol = lane & 31;                 // 32x32 accumulator: row = (e & 3) + 8 (e >> 2) + lrow, col = lcol
    auto ring = [&](int slot, int operand) { return smem + (slot * 2 + operand) * B3_GRAN; };
    // Register-level pipeline: the fragments of stage s+1 are read from the ring while the MFMAs of stage s run from registers, and
    // the ring runs two stages ahead of that (slot s%3 is free as soon as stage s sits in registers):
    //   top of stage s:  wait DMA(s+1) [DMA(s+2) may still fly], barrier -> issue DMA(s+3) into slot s%3 -> ds_read stage s+1 -> MFMAs(s)
    // so neither the LDS-DMA latency (about 1.5 stages) nor the ds_read latency stands in front of an MFMA.
    static_assert(B3_NBUF == 3, "the register pipeline wants a three-slot ring");
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < nst) {
            b3_dma<TM>(srcA + (size_t)d * B3_GRAN, hA, ring(d, 0), tid, wave);
            b3_dma<TM>(srcB + (size_t)d * B3_GRAN, hB, ring(d, 1), tid, wave);
        }
    b3_wait<TM, 2>(wave);                         // nst >= 4: three stages requested, the first one has landed
    __syncthreads();
#ifdef PRE3_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_k9rt[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 7) { g_probe[0] = __builtin_amdgcn_s_memtime(); g_probe[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
    f32x16_t acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragments are carried as 4 x i32 (the same 16 bytes): loop-carried <8 x bf16> values get legalised element by element
    typedef int frag_t __attribute__((ext_vector_type(4)));
    frag_t A[3][NB], B[3][NB], A2[3][NB], B2[3][NB];
#define B3_READ_FRAGS(slot, FA, FB) do { \
        const frag_t *sA_ = reinterpret_cast<const frag_t *>(ring(slot, 0)) + (NB * wi) * 64 + lane; \
        const frag_t *sB_ = reinterpret_cast<const frag_t *>(ring(slot, 1)) + (NB * wj) * 64 + lane; \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) { FA[p][i] = sA_[p * PL + i * 64]; FB[p][i] = sB_[p * PL + i * 64]; } \
    } while (0)
#define B3_MMA(FA, pa, FB, pb) \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) \
            _Pragma("unroll") for (int j = 0; j < NB; ++j) \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FA[pa][i]), __builtin_bit_cast(bf16x8_t, FB[pb][j]), acc[i][j], 0, 0, 0)
    // largest terms first; six of the nine partial products
#define B3_MFMAS(FA, FB) do { B3_MMA(FA, 0, FB, 0); B3_MMA(FA, 0, FB, 1); B3_MMA(FA, 1, FB, 0); B3_MMA(FA, 1, FB, 1); B3_MMA(FA, 0, FB, 2); B3_MMA(FA, 2, FB, 0); } while (0)
    // one stage: FA/FB hold stage s, GA/GB receive stage s+1 (ping-pong over two unrolled stages: no register copies).
    // LEFT: stages that may still be in flight once stage s+1 has landed; DMA: whether stage s+3 exists.  nst is a multiple of 4.
#define B3_STAGE(s_, FA, FB, GA, GB, LEFT, DMA) do { \
        const int nxt_ = buf == 2 ? 0 : buf + 1; \
        b3_wait<TM, LEFT>(wave);                                             /* stage s+1 has landed (this wave's granules) */ \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                  /* this wave's reads of stage s are in registers: its slot may be refilled */ \
        __builtin_amdgcn_s_barrier();                                        /* raw barrier: __syncthreads() would add vmcnt(0) for the DMAs still in flight */ \
        if (DMA) { \
            b3_dma<TM>(srcA + (size_t)((s_) + 3) * B3_GRAN, hA, ring(buf, 0), tid, wave); \
            b3_dma<TM>(srcB + (size_t)((s_) + 3) * B3_GRAN, hB, ring(buf, 1), tid, wave); \
        } \
        B3_READ_FRAGS(nxt_, GA, GB); \
        B3_MFMAS(FA, FB); \
        buf = nxt_; \
    } while (0)
    B3_READ_FRAGS(0, A, B);
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the loop's first MFMA must not inherit a wait that also covers the next reads
    int buf = 0;                                  // slot of stage s
    int s = 0;
    for (; s + 5 < nst; s += 2) {
        B3_STAGE(s, A, B, A2, B2, 1, true);
        B3_STAGE(s + 1, A2, B2, A, B, 1, true);
    }
    // the last four stages, straight-line: s = nst - 4
    B3_STAGE(s, A, B, A2, B2, 1, true);
    B3_STAGE(s + 1, A2, B2, A, B, 1, false);
    {   // stage nst-2, with the P tile requested behind its synchronisation: two stages of MFMAs cover the loads' latency, and from
        // here on only two fragment sets' worth of registers are live besides the accumulators
        const int nxt_ = buf == 2 ? 0 : buf + 1;
        b3_wait<TM, 0>(wave);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        B3_READ_FRAGS(nxt_, A2, B2);
        buf = nxt_;
    }
    float pv[NB][NB][16];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                pv[i][j][e] = P[(size_t)(R0 + i * 32 + (e & 3) + 8 * (e >> 2) + lrow) * ld + C0 + j * 32 + lcol];
    B3_MFMAS(A, B);
    B3_MFMAS(A2, B2);
#undef B3_STAGE
#undef B3_MFMAS
#undef B3_MMA
#undef B3_READ_FRAGS
    __syncthreads();                              // the patches below alias the ring
#ifdef PRE3_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_k9rt[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 7) { g_probe[1] = __builtin_amdgcn_s_memtime(); g_probe[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // epilogue: new = P - acc to (row, col) and, through a wave-private patch, to (col, row).  On a diagonal tile only the upper
    // triangle is written directly and its mirror image copied, so P stays exactly symmetric whatever the order of the six products.
    // Off the diagonal both images leave as 16-byte stores (a quarter of the store instructions: the tail of this kernel is store-issue
    // bound): the block goes through two wave-private LDS patches, one as is and one transposed ([32][36] floats each: rows 16-byte aligned).
    typedef float f4_t __attribute__((ext_vector_type(4)));
    float (*patch)[36] = reinterpret_cast<float (*)[36]>(reinterpret_cast<float *>(smem) + wave * (2 * 32 * 36));
    float (*patchT)[36] = patch + 32;
    static_assert(B3_NBUF * 2 * B3_GRAN * 16 >= 4 * 2 * 32 * 36 * 4, "patches must fit in the ring");
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int gi = NB * wi + i, gj = NB * wj + j;
            if (diag && gi > gj) continue;                               // below the diagonal: the mirror of another block
            const bool dblk = diag && gi == gj;
            const int r0 = R0 + i * 32, c0 = C0 + j * 32;
            if (dblk) {                                                   // diagonal 32x32 block: element-wise predicates
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = (e & 3) + 8 * (e >> 2) + lrow;
                    const float v = pv[i][j][e] - acc[i][j][e];
                    if (lr <= lcol) store_wt(P + (size_t)(r0 + lr) * ld + c0 + lcol, v, wt);
                    patch[lr][lcol] = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int rr = lane & 31, half = lane >> 5;
#pragma unroll
                for (int cc = 0; cc < 32; cc += 2) {
                    const int c = cc + half;
                    if (rr < c) store_wt(P + (size_t)(c0 + c) * ld + r0 + rr, patch[rr][c], wt);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {                             // e = 4g..4g+3: rows 8g + lrow + 0..3 of column lcol
                    f4_t v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { v[t] = pv[i][j][4 * g + t] - acc[i][j][4 * g + t]; patch[8 * g + lrow + t][lcol] = v[t]; }
                    *reinterpret_cast<f4_t *>(&patchT[lcol][8 * g + lrow]) = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int rr = lane >> 3, c4 = (lane & 7) * 4;            // 8 rows x 128 B per instruction
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = rr + 8 * it;
                    store_wt(reinterpret_cast<f4_t *>(P + (size_t)(r0 + row) * ld + c0 + c4), *reinterpret_cast<const f4_t *>(&patch[row][c4]), wt);
                    store_wt(reinterpret_cast<f4_t *>(P + (size_t)(c0 + row) * ld + r0 + c4), *reinterpret_cast<const f4_t *>(&patchT[row][c4]), wt);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
}

// tiles[b] = (bi | big << 16, bj) in 64-column block units: the launch holds whole rounds of 128x128 tiles (one per CU and round)
// and covers what is left over with 64x64 tiles, so that no CU ends up with a second large tile while the others idle.
__global__ __launch_bounds__(256) void k_downdate_b3(float *__restrict__ P, int ld, const bf16x8_t *__restrict__ Wp, int nst_total, int nst,
                                                      const float *__restrict__ W, int ldw, const int2 *__restrict__ tiles, XUpd xu, ProjRide pr)
{
    __shared__ __attribute__((aligned(16))) bf16x8_t smem[B3_NBUF * 2 * B3_GRAN];       // 24 KB per ring slot: [slot][A | B][plane][fragment][lane]
    static_assert(sizeof(smem) >= 4 * 32 * 33 * sizeof(float) && sizeof(smem) >= sizeof(double) * (16 * 64 + 4), "patches and riders borrow the ring");
    if (xu.gate != nullptr) {
        if (xu.gate[8] != 1) return;                            // the fused HI update in front found nothing to do (or too much: the host follows up)
        nst = 2 * xu.gate[5] > NB ? 8 : 4;                      // one panel of rows (four k-stages) up to 32 landmarks, two up to 64
    }
    if ((int)blockIdx.x >= xu.n_tiles) {            // riders, as in k_downdate_1t
        __builtin_amdgcn_s_setprio(3);
        const int nx = xu.nx, rb = blockIdx.x - xu.n_tiles;
        if (rb < nx) {
            update_x_block<float>(rb, xu.n, xu.gate != nullptr ? 2 * xu.gate[5] : xu.r, W, ldw, ld, xu.x_prior, xu.x_out, xu.params, reinterpret_cast<double *>(smem));
            if (pr.n_blocks) ride_signal(pr.ctr);
        } else {
            proj_ride_block(pr, rb - nx);
        }
        return;
    }
    const int2 t = tiles[blockIdx.x];
#ifdef PRE3_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
        g_k9rt[blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_k9hw[blockIdx.x] = (hw & 0xffff) | ((xcc & 0xf) << 16) | ((unsigned)(t.x >> 16) << 24);
    }
#endif
    if (t.x >> 16) b3_tile<128>(P, ld, Wp, nst_total, nst, t.x & 0xffff, t.y, smem, xu.wt != 0);
    else b3_tile<64>(P, ld, Wp, nst_total, nst, t.x, t.y, smem, xu.wt != 0);
#ifdef PRE3_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_k9rt[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
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
    PRE3_TRY(pend_flush(c));
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

static inline int hp_build_mb() { static const int v = getenv("PRE3_HP_MB") ? atoi(getenv("PRE3_HP_MB")) : 1; return v; }      // 0: one measurement per workgroup (k_ell_HP_build)

// the same for a subset: row pair a = measurement sel[a] (sel == nullptr: a), a < nsel; the padding up to r_pad is zero rows
int launch_ell_HP_build_sel(pre3_ctx *c, int nsel, const int32_t *sel_dev, void *dst)
{
    PRE3_TRY(pend_flush(c));
    const int r_pad = round_up(2 * nsel, NB);
    if (r_pad == 0) return PRE3_OK;
    const int gx = ceil_div(c->ldw / 4, 256), ny = r_pad / 2;
    InnovRide ir{};
    if (c->ride_innovation && c->N > 0) {            // pre3_step_all: S_i of every predicted landmark (+ the clearing of last frame's inlier flags) in this launch
        const int nb = ceil_div(c->N * 16, 256), rows = ceil_div(nb, gx);
        ir = InnovRide{ rows * gx, c->N, c->ld, (int)(c->flags_bytes / sizeof(int32_t)), c->lm.type, c->lm.off, c->lm.has_h, c->P, c->lm.Hc, c->lm.Hl,
                        c->lm.S, c->lm.has_S, (int32_t *)((unsigned char *)c->inbox_dev + c->off_flags) };
    }
    if (hp_build_mb() > 0) {
        constexpr int MBF = 4, MBD = 2;
        auto kd = k_ell_HP_build_mb<double, MBD>; auto kf = k_ell_HP_build_mb<float, MBF>;
        const int nyg = c->dtype == PRE3_F32 ? ceil_div(ny, MBF) : ceil_div(ny, MBD);
        dim3 g(gx, nyg + (ir.n_blocks ? ir.n_blocks / gx : 0)), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL(kd, g, b, 0, c->stream, nsel, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                               c->row_col, (double *)c->row_val, c->row_nu, (const double *)c->P, c->ld, (double *)dst, c->ldw, (const int32_t *)nullptr, 0, nyg, ir, sel_dev, InboxRide{}, PendW{}),
            hipLaunchKernelGGL(kf, g, b, 0, c->stream, nsel, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                               c->row_col, (float *)c->row_val, c->row_nu, (const float *)c->P, c->ld, (float *)dst, c->ldw, (const int32_t *)nullptr, 0, nyg, ir, sel_dev, InboxRide{}, PendW{}));
        PRE3_HIP(hipGetLastError());
        if (ir.n_blocks) { c->ride_innovation = false; c->innovated = true; }
        return PRE3_OK;
    }
    dim3 g(gx, ny + (ir.n_blocks ? ir.n_blocks / gx : 0)), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_ell_HP_build<double>, g, b, 0, c->stream, nsel, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                           c->row_col, (double *)c->row_val, c->row_nu, (const double *)c->P, c->ld, (double *)dst, c->ldw, (const int32_t *)nullptr, 0, ny, ir, sel_dev),
        hipLaunchKernelGGL(k_ell_HP_build<float>, g, b, 0, c->stream, nsel, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                           c->row_col, (float *)c->row_val, c->row_nu, (const float *)c->P, c->ld, (float *)dst, c->ldw, (const int32_t *)nullptr, 0, ny, ir, sel_dev));
    PRE3_HIP(hipGetLastError());
    if (ir.n_blocks) { c->ride_innovation = false; c->innovated = true; }
    return PRE3_OK;
}

// rows of all m measurements built and multiplied in one launch (replaces launch_build_rows_impl + launch_ell_HP)
int launch_ell_HP_build(pre3_ctx *c, void *dst, const int32_t *need, int need_tag, const InboxRide *ib_in)
{
    const InboxRide ib = ib_in ? *ib_in : InboxRide{};
    const int ib_rows = ib.n16 > 0 ? 1 : 0;
    const int r_pad = round_up(2 * c->m, NB);
    const int gx = ceil_div(c->ldw / 4, 256), ny = r_pad / 2;
    InnovRide ir{};
    if (c->ride_innovation && c->N > 0) {            // pre3_step: S_i of every predicted landmark (+ the clearing of last frame's inlier flags) in this launch
        const int nb = ceil_div(c->N * 16, 256), rows = ceil_div(nb, gx);
        ir = InnovRide{ rows * gx, c->N, c->ld, (int)(c->flags_bytes / sizeof(int32_t)), c->lm.type, c->lm.off, c->lm.has_h, c->P, c->lm.Hc, c->lm.Hl,
                        c->lm.S, c->lm.has_S, (int32_t *)((unsigned char *)c->inbox_dev + c->off_flags) };
    }
    // PRE3_OPT_PEND_HI: this launch (rows of ALL measurements at the predicted state, S_i riding) reads P - W~'W~; every other form runs behind the flush
    PendW pw = pend_args(c);
    if (pw.rows > 0 && !(hp_build_mb() > 0 && c->dtype == PRE3_F32 && need == nullptr)) { PRE3_TRY(pend_flush(c)); pw = PendW{}; }
    if (pw.rows > 0) { ir.pend_W = pw.W; ir.pend_ldw = pw.ldw; ir.pend_rows = pw.rows; }
    if (hp_build_mb() > 0 && need == nullptr) {      // (a sharded round's slice -- need != nullptr -- skips most measurements: one per workgroup, k_ell_HP_build, returns at once for those)
        constexpr int MBF = 4, MBD = 2;
        static const int mbp = getenv("PRE3_PEND_MB") ? atoi(getenv("PRE3_PEND_MB")) : 4;      // measurements per workgroup of the pending form: 4 (17.1 -> 16.5 us with their landmark rows gathered two at a time); 8 halves the loads of W~ but leaves 156 workgroups for 256 CUs (18.7 us), 2 doubles them (21.5 us)
        const int mbf = pw.rows > 0 ? (mbp == 8 ? 8 : 4) : MBF;
        auto kd = k_ell_HP_build_mb<double, MBD>; auto kf = pw.rows > 0 ? (mbf == 8 ? k_ell_HP_build_mb<float, 8, true> : k_ell_HP_build_mb<float, 4, true>) : k_ell_HP_build_mb<float, MBF, false>;
        const int nyg = c->dtype == PRE3_F32 ? ceil_div(ny, mbf) : ceil_div(ny, MBD);
        dim3 g(gx, nyg + (ir.n_blocks ? ir.n_blocks / gx : 0) + ib_rows), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL(kd, g, b, 0, c->stream, c->m, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                               c->row_col, (double *)c->row_val, c->row_nu, (const double *)c->P, c->ld, (double *)dst, c->ldw, need, need_tag, nyg, ir, (const int32_t *)nullptr, ib, PendW{}),
            hipLaunchKernelGGL(kf, g, b, 0, c->stream, c->m, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                               c->row_col, (float *)c->row_val, c->row_nu, (const float *)c->P, c->ld, (float *)dst, c->ldw, need, need_tag, nyg, ir, (const int32_t *)nullptr, ib, pw));
        PRE3_HIP(hipGetLastError());
        if (ir.n_blocks) { c->ride_innovation = false; c->innovated = true; }
        return PRE3_OK;
    }
    dim3 g(gx, ny + (ir.n_blocks ? ir.n_blocks / gx : 0) + ib_rows), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_ell_HP_build<double>, g, b, 0, c->stream, c->m, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                           c->row_col, (double *)c->row_val, c->row_nu, (const double *)c->P, c->ld, (double *)dst, c->ldw, need, need_tag, ny, ir, (const int32_t *)nullptr, ib),
        hipLaunchKernelGGL(k_ell_HP_build<float>, g, b, 0, c->stream, c->m, r_pad, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl, c->lm.z, c->lm.h,
                           c->row_col, (float *)c->row_val, c->row_nu, (const float *)c->P, c->ld, (float *)dst, c->ldw, need, need_tag, ny, ir, (const int32_t *)nullptr, ib));
    PRE3_HIP(hipGetLastError());
    if (ir.n_blocks) { c->ride_innovation = false; c->innovated = true; }
    return PRE3_OK;
}

int launch_ell_G(pre3_ctx *c, int r, const void *HPsrc, void *dst, int ldg, int add_identity, const void *Rdense, bool lower_only)
{
    int r_pad = round_up(r, NB);
    dim3 g(ceil_div(r_pad, 64), r_pad), b(64);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_ell_G<double>, g, b, 0, c->stream, r, r_pad, c->row_col, (const double *)c->row_val, (const double *)HPsrc,
                           c->ldw, (double *)dst, ldg, add_identity, (const double *)Rdense, lower_only ? 1 : 0),
        hipLaunchKernelGGL(k_ell_G<float>, g, b, 0, c->stream, r, r_pad, c->row_col, (const float *)c->row_val, (const float *)HPsrc,
                           c->ldw, (float *)dst, ldg, add_identity, (const float *)Rdense, lower_only ? 1 : 0));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_ell_G_hyp(pre3_ctx *c, int k, int lo, int hi, int ldg)
{
    if (hi <= lo) return PRE3_OK;
    dim3 g(ceil_div((hi - lo) * k * (2 * k + 1), 256)), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_ell_G_hyp<double>, g, b, 0, c->stream, c->hyp, k, lo, hi, c->row_col, (const double *)c->row_val, (const double *)c->HP, c->ldw, (double *)c->G, ldg),
        hipLaunchKernelGGL(k_ell_G_hyp<float>, g, b, 0, c->stream, c->hyp, k, lo, hi, c->row_col, (const float *)c->row_val, (const float *)c->HP, c->ldw, (float *)c->G, ldg));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

static int launch_chol_solve(pre3_ctx *c, int r_pad, bool first_done = false, bool predicted_prior = false, int r = -1 /* real rows, if known */, int which_prior = -1)
{
    int nrb = r_pad / NB, nW = c->ldw / NB;
    c->split_rows = 0;
    if (first_done && c->cholp_done) {             // pre3_update_li's speculative launch was the persistent form: everything is done
        c->cholp_done = false;
        c->split_rows = nrb * NB;                  // (c->dd_done: the groups its consumers have down-dated already)
        if (c->kt.pending) cholp_timing_rows(c, r > 0 ? r : r_pad);
        return PRE3_OK;
    }
    c->cholp_done = false; c->dd_done = 0; c->x_done = false;
    // (one panel is one launch in either form, and the lock-step form has no hand-off in it: 12.5 us against 17 -- taken for the rescue
    // stage's small updates; an update of the PREDICTED state keeps the persistent form at any size, because pre3_update_li's speculative
    // launch -- row count still on the device -- cannot choose, and the two ways into that update must compute the same thing)
    if (!first_done && (nrb >= 2 || predicted_prior) && cholp_usable(c, nrb)) return launch_cholp(c, nrb, nrb, r, which_prior);
    {
        const bool split = c->k9_b3 && c->dtype == PRE3_F32 && c->Wp != nullptr;
        static const int pro_env = getenv("PRE3_CHOL_PRO_B3") ? atoi(getenv("PRE3_CHOL_PRO_B3")) : 1;
        const bool pro_planes = split && pro_env && c->Sp != nullptr;       // pending updates on the bf16 MFMA as well
        static const int early_env = getenv("PRE3_CHOL_EARLY") ? atoi(getenv("PRE3_CHOL_EARLY")) : 1;
        const int rows_last = (early_env && r > 0 && r <= r_pad && r > r_pad - NB) ? r - (r_pad - NB) : 0;      // real rows of the last panel (0: unknown / switched off)
        // a trailing update with far more tiles than k_chol_step has slots (2 workgroups per CU: its 75 KB of LDS) goes out as a launch
        // of its own (no LDS, 8 workgroups per CU)
        static const int trail_split = getenv("PRE3_CHOL_TRAIL_SPLIT") ? atoi(getenv("PRE3_CHOL_TRAIL_SPLIT")) : 4;
        auto n_trail = [&](int J) { const int nK = J >= 1 ? nrb - J - 1 : 0; return nK * (nK + 1) / 2 + nK * nW; };      // tiles of panel J-1's update of the column blocks >= J+1
        auto own_trail_at = [&](int J) { return pro_planes && trail_split > 0 && n_trail(J) > trail_split * 2 * c->num_cus; };
        // Round 5: where the trailing update is a launch of its own it is bound by the read-modify-write of W's remaining rows (N = 2000: 4.8 GB per
        // update over 40 panels), so it sweeps a GROUP of p panels at a time -- behind the last panel J of a group: panels J-p+1 .. J on the column
        // blocks >= J+2, each tile read and written once -- and a panel applies what is still pending for its own column itself (the first panel of
        // a group: the whole group before it; the t-th: the t panels of its own group in front of it).  Every product is rounded and subtracted in
        // the order of the one-panel sweep: the same bits (PRE3_CHOL_TRAIL_P=1: that sweep).  The grouping ends with the first panel Jt of a group
        // from which on the update is small enough to ride in the panels' launches.
        static const int trail_p = std::min(4, std::max(1, getenv("PRE3_CHOL_TRAIL_P") ? atoi(getenv("PRE3_CHOL_TRAIL_P")) : 4));
        int Jt = 0;
        if (trail_p > 1 && own_trail_at(1)) { Jt = trail_p; while (Jt < nrb && own_trail_at(Jt)) Jt += trail_p; }
        for (int J = first_done ? 1 : 0; J < nrb; ++J) {
            const int nS = nrb - J - 1, nP = 1 + nS + nW;
            const bool paired = J >= 1 && J <= Jt;                          // this panel's column is brought up to date by the panel itself
            const int npend = !paired ? 1 : (J % trail_p == 0 ? trail_p : J % trail_p);
            const int nT = paired ? 0 : n_trail(J);
            const int ncb = (split && !pro_planes && J >= 1) ? c->ld / B3_T : 0;   // split riders: 4 stages x ncb column blocks of row block J-1
            const bool own_trail = !paired && own_trail_at(J);
            if (own_trail)
                hipLaunchKernelGGL(k_chol_trail_b3, dim3(nT), dim3(256), 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W, c->ldw, J - 1, J + 1, nrb, nW,
                                   c->Wp, c->rcap / B3_BK, c->Sp, c->rcap / NB, -1);
            const int nT_in = own_trail ? 0 : nT;
            dim3 g(nP + nT_in + 4 * ncb), bP(CH_NTH);
            c->chol_target += (unsigned)(nP - 1);                           // every non-diagonal workgroup of the panel arrives once
            const bool early = rows_last > 0 && rows_last <= NB - 8 && J == nrb - 1;       // at least one padded sub-panel in the last panel
            if (early) {
                DISPATCH_T(c,
                    hipLaunchKernelGGL((k_chol_step<double, true>), g, bP, 0, c->stream, (double *)c->Smat, r_pad, (double *)c->W, c->ldw, J, nrb, nW, nP, c->stats + 6, c->chol_arrive, c->chol_target,
                                       nP + nT_in, nullptr, 0, 0, 0, nullptr, 0, nullptr, 0, rows_last, 1),
                    hipLaunchKernelGGL((k_chol_step<float, true>), g, bP, 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W, c->ldw, J, nrb, nW, nP, c->stats + 6, c->chol_arrive, c->chol_target,
                                       nP + nT_in, split ? c->Wp : nullptr, c->rcap / B3_BK, ncb, c->ld, pro_planes ? c->Sp : nullptr, c->rcap / NB, nullptr, 0, rows_last, npend));
            } else {
            DISPATCH_T(c,
                hipLaunchKernelGGL(k_chol_step<double>, g, bP, 0, c->stream, (double *)c->Smat, r_pad, (double *)c->W, c->ldw, J, nrb, nW, nP, c->stats + 6, c->chol_arrive, c->chol_target,
                                   nP + nT_in, nullptr, 0, 0, 0, nullptr, 0, nullptr, 0, 0, 1),
                hipLaunchKernelGGL(k_chol_step<float>, g, bP, 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W, c->ldw, J, nrb, nW, nP, c->stats + 6, c->chol_arrive, c->chol_target,
                                   nP + nT_in, split ? c->Wp : nullptr, c->rcap / B3_BK, ncb, c->ld, pro_planes ? c->Sp : nullptr, c->rcap / NB, nullptr, 0, 0, npend));
            }
            if (paired && J % trail_p == trail_p - 1 && J + 2 <= nrb - 1) {
                // the group J-p+1 .. J on everything from column block J+2 on
                const int nK2 = nrb - (J + 2), nT2 = nK2 * (nK2 + 1) / 2 + nK2 * nW;
                static const int t128 = getenv("PRE3_CHOL_TRAIL_T128") ? atoi(getenv("PRE3_CHOL_TRAIL_T128")) : 2;      // W part in 128 x 128 super-tiles: 2 = operands staged through LDS, 1 = fragments straight from memory (0: 64 x 64 tiles)
                if (t128 == 2) {
                    static std::atomic<bool> attr{ false };
                    if (!attr.exchange(true)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_trail_b3l), hipFuncAttributeMaxDynamicSharedMemorySize, TL_NSLOT * TL_STAGE * 16);
                    hipLaunchKernelGGL(k_chol_trail_b3l, dim3(nK2 * (nK2 + 1) / 2 + ((nK2 + 1) / 2) * ((nW + 1) / 2)), dim3(256), TL_NSLOT * TL_STAGE * 16, c->stream, (float *)c->Smat, r_pad,
                                       (float *)c->W, c->ldw, J - trail_p + 1, J + 2, nrb, nW, c->Wp, c->rcap / B3_BK, c->Sp, c->rcap / NB, J);
                } else if (t128)
                    hipLaunchKernelGGL(k_chol_trail_b3w, dim3(nK2 * (nK2 + 1) / 2 + ((nK2 + 1) / 2) * ((nW + 1) / 2)), dim3(256), 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W,
                                       c->ldw, J - trail_p + 1, J + 2, nrb, nW, c->Wp, c->rcap / B3_BK, c->Sp, c->rcap / NB, J);
                else
                hipLaunchKernelGGL(k_chol_trail_b3, dim3(nT2), dim3(256), 0, c->stream, (float *)c->Smat, r_pad, (float *)c->W, c->ldw, J - trail_p + 1, J + 2, nrb, nW,
                                   c->Wp, c->rcap / B3_BK, c->Sp, c->rcap / NB, J);
            }
        }
        if (split) c->split_rows = nrb * NB;
        PRE3_HIP(hipGetLastError());
        return PRE3_OK;
    }
}

int launch_downdate(pre3_ctx *c, int r, const void *W, int which_prior)
{
    int r_pad = round_up(r, NB);
    // persistent grid: 2 workgroups per CU (or one per tile when there are fewer tiles); the tile counter is
    // monotonic across launches (each launch consumes grid + n_tiles tickets), reset long before it could wrap
    static const int wgs_per_cu = getenv("PRE3_K9_WGS") ? atoi(getenv("PRE3_K9_WGS")) : 3;
    const int gsz = std::min(c->n_tiles, wgs_per_cu * c->num_cus);
    dim3 g(gsz), b(256);
    static const int force = getenv("PRE3_K9_FORM") ? atoi(getenv("PRE3_K9_FORM")) : 0;     // 1: one-tile, 2: persistent (experiments)
    // all tiles resident at once: 5 workgroups/CU in fp32 (16.9 KB of LDS each), 4 in fp64 (33.8 KB)
    const bool one_tile = force == 1 || (force == 0 && c->n_tiles <= 5 * c->num_cus);
    // fp32: three-way bf16 split on the bf16 matrix cores (k_split_w + k_downdate_b3); PRE3_K9_B3=0 keeps the f32-MFMA forms
    const bool use_b3 = c->k9_b3 && c->dtype == PRE3_F32 && c->Wp != nullptr;
    if (!use_b3 && !one_tile && which_prior >= 0) PRE3_TRY(launch_update_x(c, which_prior, r));          // (also resets the ticket counters)
    if (!use_b3 && !one_tile) {                                    // ticket counters of the persistent form
        if (!c->tile_ctr_clean) PRE3_HIP(hipMemsetAsync(c->tile_ctr, 0, sizeof(unsigned int) * 8, c->stream));
        c->tile_ctr_clean = false;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // only launches in the matrix-bound regime are bracketed: the updates of the PREDICTED state (the LI updates, r_pad >= 128; and
    // pre3_bench_downdate).  The rescue stage's HI update -- a few dozen rows, at most a couple of panels early in a sequence -- is a
    // read-modify-write of P at HBM speed and would only dilute the figure, and with 'one in N' it would alias with the LI/HI alternation
    const bool timed = c->kt.enabled && c->dd_done == 0 && r_pad >= 2 * NB && which_prior != PRE3_X_K_K && (c->kt.seen++ % c->kt.every) == 0;
    if (timed) {
        if ((size_t)c->kt.used + 2 > c->kt.ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; PRE3_HIP(hipEventCreate(&e)); c->kt.ev.push_back(e); }
        }
        e0 = c->kt.ev[c->kt.used]; e1 = c->kt.ev[c->kt.used + 1];
        c->kt.used += 2;
        PRE3_HIP(hipEventRecord(e0, c->stream));
    }
    if (use_b3) {
        // k-stages of 16 rows: rows r .. r_pad-1 of W are zero (identity padding of S against zero rows of H*P), so the stages that hold nothing
        // else are skipped -- the tile loop wants an even count of at least four (r = 518: 34 stages instead of 36)
        const int nst = std::min(r_pad / B3_BK, std::max(4, 2 * ceil_div(r, 2 * B3_BK))), nst_total = c->rcap / B3_BK;
        const int st0 = c->split_rows / B3_BK;                         // row blocks the factorisation's riders have already split
        c->split_rows = 0;
        if (st0 < nst) {
            dim3 gs(c->ld / B3_T, nst - st0);
            hipLaunchKernelGGL(k_split_w, gs, b, 0, c->stream, (const float *)W, c->ldw, (bf16x8_t *)c->Wp, nst_total, st0);
        }
        // the persistent factorisation's consumers have down-dated the tiles of groups [0, dd_done) already (pre3_cholp.hip): what is left
        // -- nothing at N = 500 -- goes out as 64 x 64 tiles; the x-update and the rescue's projection ride in this launch either way
        const int2 *tiles = (const int2 *)c->tiles128;
        int n_tiles_launch = c->n_tiles128;
        if (c->dd_done > 0) {
            const int t0 = c->dd_tile_off[c->dd_done];
            tiles = (const int2 *)c->dd_tiles + t0;
            n_tiles_launch = c->dd_tile_off.back() - t0;
        }
        c->dd_done = 0;
        XUpd xu{ n_tiles_launch, c->n, r, which_prior == PRE3_X_K_K ? c->x_kk : c->x_km1, c->x_kk, c->pred_params, nullptr, 0 };
        xu.wt = k9_write_through();
        const bool x_done = c->x_done && which_prior >= 0;            // the factorisation's strips have computed x_k_k already
        c->x_done = false;
        const int nx = (which_prior >= 0 && !x_done) ? ceil_div(c->n, 64) : 0;
        xu.nx = nx;
        ProjRide pr{};
        if (which_prior >= 0 && c->ride_rescue_projection && c->N > 0) {
            if (n_tiles_launch + nx == 0) {                           // nothing left for this launch: the projection rides with the Jnorm pass
                c->proj_with_jnorm = true;
            } else {
                pr = make_proj_ride(c, PRE3_X_K_K, 0, 1, nx);
                c->rescue_projected = true;
            }
            c->ride_rescue_projection = false;
        }
        dim3 g1(n_tiles_launch + nx + pr.n_blocks);
        if (g1.x > 0) {
            hipLaunchKernelGGL(k_downdate_b3, g1, b, 0, c->stream, (float *)c->P, c->ld, (const bf16x8_t *)c->Wp, nst_total, nst, (const float *)W, c->ldw,
                               tiles, xu, pr);
        }
    } else if (one_tile) {
        XUpd xu{ c->n_tiles, c->n, r, which_prior == PRE3_X_K_K ? c->x_kk : c->x_km1, c->x_kk, c->pred_params, nullptr, 0 };
        xu.wt = k9_write_through();
        const int nx = which_prior >= 0 ? ceil_div(c->n, 64) : 0;
        xu.nx = nx;
        ProjRide pr{};
        if (nx > 0 && c->ride_rescue_projection && c->N > 0) {      // the rescue's projection (stale h kept: clear_first = 0) rides along
            pr = make_proj_ride(c, PRE3_X_K_K, 0, 1, nx);
            c->ride_rescue_projection = false; c->rescue_projected = true;
        }
        dim3 g1(c->n_tiles + nx + pr.n_blocks);
        // fp64 with few tiles per CU (configs[1]: 210 tiles on 256 CUs): two waves per wave tile, so that every SIMD has two waves to overlap
        static const int nw8_env = getenv("PRE3_K9_F64_NW8") ? atoi(getenv("PRE3_K9_F64_NW8")) : 3;      // (measured at n = 1213, r = 320: 0: 24.3 us; 1 (16-row stages, 8 waves): 30.5; 2 (32-row stages): 24.8; 3 (32-row stages, 8 waves): 22.5)
        if (c->dtype == PRE3_F64 && nw8_env == 1 && c->n_tiles <= 2 * c->num_cus) {
            hipLaunchKernelGGL((k_downdate_1t<double, 16, 8>), g1, dim3(512), 0, c->stream, (double *)c->P, c->ld, (const double *)W, c->ldw, r_pad, (const int2 *)c->tiles_flat, c->num_cus, xu, pr);
        } else if (c->dtype == PRE3_F64 && nw8_env == 2 && c->n_tiles <= 2 * c->num_cus && r_pad % 32 == 0) {
            hipLaunchKernelGGL((k_downdate_1t<double, 32, 4>), g1, dim3(256), 0, c->stream, (double *)c->P, c->ld, (const double *)W, c->ldw, r_pad, (const int2 *)c->tiles_flat, c->num_cus, xu, pr);
        } else if (c->dtype == PRE3_F64 && nw8_env == 3 && c->n_tiles <= 2 * c->num_cus && r_pad % 32 == 0) {
            hipLaunchKernelGGL((k_downdate_1t<double, 32, 8>), g1, dim3(512), 0, c->stream, (double *)c->P, c->ld, (const double *)W, c->ldw, r_pad, (const int2 *)c->tiles_flat, c->num_cus, xu, pr);
        } else
        DISPATCH_T(c,
            hipLaunchKernelGGL((k_downdate_1t<double, 16>), g1, b, 0, c->stream, (double *)c->P, c->ld, (const double *)W, c->ldw, r_pad, (const int2 *)c->tiles_flat, c->num_cus, xu, pr),
            hipLaunchKernelGGL((k_downdate_1t<float, 32>), g1, b, 0, c->stream, (float *)c->P, c->ld, (const float *)W, c->ldw, r_pad, (const int2 *)c->tiles_flat, c->num_cus, xu, pr));
    } else
    DISPATCH_T(c,
        hipLaunchKernelGGL((k_downdate<double, 32>), g, b, 0, c->stream, (double *)c->P, c->ld, (const double *)W, c->ldw, r_pad, (const int2 *)c->tiles, c->tiles_stride, c->tile_cnt, c->tile_ctr),
        hipLaunchKernelGGL((k_downdate<float, 32>), g, b, 0, c->stream, (float *)c->P, c->ld, (const float *)W, c->ldw, r_pad, (const int2 *)c->tiles, c->tiles_stride, c->tile_cnt, c->tile_ctr));
    if (timed) {
        PRE3_HIP(hipEventRecord(e1, c->stream));
        c->kt.flops += (double)c->n * ((double)c->n + 1.0) * r;          // SYRK count n(n+1)r; the survey's un-halved figure is 2 n^2 r
        c->kt.bytes += 1.5 * c->n * (double)c->n * c->esz + (double)c->n * (double)r * c->esz;
    }
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// rescue_hi_inliers.m:44-47 + ekf_update_hi_inliers.m:45-58 without the host: k_hi_fused (collection, rows, H*P, S, one-panel factorisation and
// solve for up to 32 rescued landmarks) and, right behind it, the down-date of that update, which runs only if k_hi_fused says it did the update
// (stats[8]) and takes its row count from the device.  seq: the mailbox sequence number the collection publishes.
bool hi_fused_usable(const pre3_ctx *c)
{
    static const int env = getenv("PRE3_HI_FUSED") ? atoi(getenv("PRE3_HI_FUSED")) : 1;
    return env != 0 && c->dtype == PRE3_F32 && c->k9_b3 && c->Wp != nullptr && c->Sp != nullptr && c->m > 0 && c->m <= 16384 && c->rcap >= NB && c->N > 0;
}

// landmarks k_hi_fused updates with on its own: 64 (two panels) when the row capacity holds them (PRE3_HI_FUSED_TWO=0: one panel, A/B)
int hi_fused_max(const pre3_ctx *c)
{
    static const int two_env = getenv("PRE3_HI_FUSED_TWO") ? atoi(getenv("PRE3_HI_FUSED_TWO")) : 1;
    return two_env != 0 && c->rcap >= 2 * NB ? HF_MAXL2 : HF_MAXL;
}

PendW pend_args(const pre3_ctx *c)
{
    if (c->pend_rows <= 0 || c->W_pend == nullptr) return PendW{};
    return PendW{ c->W_pend, c->ldw, c->pend_rows, c->Wp_pend, c->rcap / B3_BK };
}

// the pending HI down-date as the launch it would have been (tiles only: the x-update went out with k_hi_fused's launch pair)
int pend_flush(pre3_ctx *c)
{
    if (c->pend_rows <= 0) return PRE3_OK;
    const int rows = c->pend_rows;
    c->pend_rows = 0;
    XUpd xu{ c->n_tiles128, c->n, 0, c->x_kk, c->x_kk, c->pred_params, nullptr, 0 };
    xu.wt = k9_write_through();
    ProjRide pr{};
    hipLaunchKernelGGL(k_downdate_b3, dim3(c->n_tiles128), dim3(256), 0, c->stream, (float *)c->P, c->ld, (const bf16x8_t *)c->Wp_pend, c->rcap / B3_BK, rows > NB ? 8 : 4,
                       (const float *)c->W_pend, c->ldw, (const int2 *)c->tiles128, xu, pr);
    PRE3_HIP(hipGetLastError());
    // (H*P / S_i built while the rows were pending were built for P - W~'W~: they stay valid)
    return PRE3_OK;
}

int launch_hi_fused(pre3_ctx *c, int32_t seq)
{
    PRE3_TRY(pend_flush(c));                       // (k_hi_fused reads P)
    HiFused a{};
    a.m = c->m; a.meas = c->meas; a.lm_ic = c->lm.ic; a.lm_li = c->lm.li; a.lm_hi = c->lm.hi; a.lm_type = c->lm.type; a.lm_off = c->lm.off;
    a.Hc = c->lm.Hc; a.Hl = c->lm.Hl; a.z = c->lm.z; a.h = c->lm.h;
    a.hi_meas = c->hi_meas; a.sel_rows = c->sel_rows; a.stats = c->stats; a.mail = c->mail_dev; a.seq = seq;
    a.row_col = c->row_col; a.row_val = (float *)c->row_val; a.row_nu = c->row_nu;
    a.P = (const float *)c->P; a.ld = c->ld; a.S = (float *)c->Smat; a.W = (float *)c->W; a.ldw = c->ldw;
    a.Wp = c->Wp; a.nst_total = c->rcap / B3_BK; a.Sp = c->Sp; a.sp_stride = c->rcap / NB; a.params = c->pred_params;
    a.max_l = hi_fused_max(c);
    static const int deal_env = getenv("PRE3_HF_DEAL") ? atoi(getenv("PRE3_HF_DEAL")) : 1;      // 0: every workgroup builds all of S in the two-panel case too (round 5)
    // (the dealt S and the pending form's state update make workgroups of this launch wait for one another: only when all of them are resident together --
    //  one per CU: each declares more than half a CU's LDS)
    const bool co_resident = 1 + c->ldw / NB <= c->num_cus;
    a.sx = deal_env && co_resident ? c->hf_sx : nullptr;
    static const int deal_min = getenv("PRE3_HF_DEAL_MIN") ? atoi(getenv("PRE3_HF_DEAL_MIN")) : 18;      // measured, k_hi_fused at 4 .. 32 landmarks: dealt 14.4 14.7 15.7 16.3 17.0 17.5 19.1 us, every workgroup for itself 12.3 13.4 14.7 16.4 17.4 18.6 22.8
    a.deal_min = deal_min;
    // PRE3_OPT_PEND_HI: W~ and its planes go to buffers of their own (the next LI update's strips overwrite W / Wp), the launch behind this one carries
    // the x-update only, and P - W~'W~ stays pending (pre3_update_hi learns the row count; pend_flush / launch_cholp end it)
    const bool pend = c->pend_opt && c->W_pend != nullptr && c->Wp_pend != nullptr && c->hf_xy != nullptr && co_resident;
    if (pend) { a.W = c->W_pend; a.Wp = c->Wp_pend; a.n = c->n; a.x = c->x_kk; a.xflag = c->hf_xy; }
    c->hi_pend_launched = pend;
    hipLaunchKernelGGL(k_hi_fused, dim3(1 + c->ldw / NB), dim3(CH_NTH), 0, c->stream, a);
    if (pend) {                                    // (the strips finish the state themselves: hf_x_update)
        PRE3_HIP(hipGetLastError());
        c->split_rows = 0; c->dd_done = 0; c->x_done = false; c->cholp_done = false;
        return PRE3_OK;
    }
    // the down-date of that update (one or two panels: four or eight k-stages, read on the device), the x-update riding along; every workgroup leaves at once unless stats[8] == 1
    const int nx = ceil_div(c->n, 64);
    XUpd xu{ c->n_tiles128, c->n, 0, c->x_kk, c->x_kk, c->pred_params, c->stats, nx };
    xu.wt = k9_write_through();
    ProjRide pr{};
    hipLaunchKernelGGL(k_downdate_b3, dim3(c->n_tiles128 + nx), dim3(256), 0, c->stream, (float *)c->P, c->ld, (const bf16x8_t *)c->Wp, c->rcap / B3_BK, 4,
                       (const float *)c->W, c->ldw, (const int2 *)c->tiles128, xu, pr);
    PRE3_HIP(hipGetLastError());
    c->split_rows = 0; c->dd_done = 0; c->x_done = false; c->cholp_done = false;
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


// rows already in c->row_* (r rows).  which_prior selects x prior; P currently holds the prior covariance.
// Rows of H*P (with the nu column) and the block of H*P*H' (+I) that the selected measurements pick out of what RANSAC
// already multiplied out, in ONE launch.  nsel < 0: the count is read on the device (n_dev, written by k_ransac_select),
// so the launch can be issued before the host has polled it; nsel_max bounds the grid.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_li(int nsel, const int32_t *__restrict__ n_dev, int gxW, const int32_t *__restrict__ sel,
                                                   const T *__restrict__ HP, T *__restrict__ W, int ldw, const T *__restrict__ G, int ldg,
                                                   T *__restrict__ S, const int32_t *__restrict__ row_col, const T *__restrict__ row_val)
{
    const int r = 2 * (nsel >= 0 ? nsel : n_dev[0]);
    const int r_pad = (r + NB - 1) / NB * NB;
    const int a = blockIdx.y;
    if (a >= r_pad) return;
    if ((int)blockIdx.x < gxW) {
        const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (j >= ldw) return;
        typedef T v4_t __attribute__((ext_vector_type(4)));
        v4_t v = { (T)0, (T)0, (T)0, (T)0 };
        if (a < r) v = *reinterpret_cast<const v4_t *>(HP + (size_t)(2 * sel[a >> 1] + (a & 1)) * ldw + j);
        *reinterpret_cast<v4_t *>(W + (size_t)a * ldw + j) = v;
    } else {
        const int b = (blockIdx.x - gxW) * 256 + threadIdx.x;
        if (b >= r_pad) return;
        T out = (a == b) ? (T)1 : (T)0;
        // G holds its lower triangle (64-column granularity); sel is ascending, so b <= a maps to a lower entry; the factorisation
        // never reads S above the diagonal (kept at 0 / 1 there)
        if (a < r && b < r && b <= a) {
            const int ra = 2 * sel[a >> 1] + (a & 1), rb = 2 * sel[b >> 1] + (b & 1);
            if (G != nullptr) out += G[(size_t)ra * ldg + rb];
            else {                                              // k_ell_G's sum for this entry, here (no H*P*H' of all measured rows is built)
                T g = (T)0;
#pragma unroll
                for (int t = 0; t < ELLW; ++t) g = ell_fma(row_val[rb * ELLW + t], HP[(size_t)ra * ldw + row_col[rb * ELLW + t]], g);
                out += g;
            }
        }
        S[(size_t)a * r_pad + b] = out;
    }
}

int launch_gather_li(pre3_ctx *c, int nsel, int nsel_max, const int32_t *sel_dev, int ldg)
{
    const int r_pad = round_up(2 * (nsel >= 0 ? nsel : nsel_max), NB);
    if (r_pad == 0) return PRE3_OK;
    const int gxW = ceil_div(c->ldw / 4, 256);
    dim3 g(gxW + ceil_div(r_pad, 256), r_pad), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_gather_li<double>, g, b, 0, c->stream, nsel, c->stats + 4, gxW, sel_dev, (const double *)c->HP, (double *)c->W, c->ldw, c->g_valid ? (const double *)c->G : nullptr, ldg, (double *)c->Smat, c->row_col, (const double *)c->row_val),
        hipLaunchKernelGGL(k_gather_li<float>, g, b, 0, c->stream, nsel, c->stats + 4, gxW, sel_dev, (const float *)c->HP, (float *)c->W, c->ldw, c->g_valid ? (const float *)c->G : nullptr, ldg, (float *)c->Smat, c->row_col, (const float *)c->row_val));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// panel 0 of the factorisation of an update whose row count is still on the device (c->stats[4]); nsel_max bounds it
int launch_chol_first_spec(pre3_ctx *c, int nsel_max)
{
    const int nrb_max = round_up(2 * nsel_max, NB) / NB, nS_max = nrb_max - 1, nW = c->ldw / NB;
    if (nrb_max <= 0) return PRE3_OK;
    const bool split = c->k9_b3 && c->dtype == PRE3_F32 && c->Wp != nullptr;
    static const int pro_env = getenv("PRE3_CHOL_PRO_B3") ? atoi(getenv("PRE3_CHOL_PRO_B3")) : 1;
    const bool pro_planes = split && pro_env && c->Sp != nullptr;
    const int nP = 1 + nS_max + nW;
    c->chol_target += (unsigned)(nP - 1);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_chol_step<double>, dim3(nP), dim3(CH_NTH), 0, c->stream, (double *)c->Smat, 0, (double *)c->W, c->ldw, 0, 0, nW, nP, c->stats + 6,
                           c->chol_arrive, c->chol_target, nP, nullptr, 0, 0, 0, nullptr, 0, c->stats + 4, nS_max),
        hipLaunchKernelGGL(k_chol_step<float>, dim3(nP), dim3(CH_NTH), 0, c->stream, (float *)c->Smat, 0, (float *)c->W, c->ldw, 0, 0, nW, nP, c->stats + 6,
                           c->chol_arrive, c->chol_target, nP, split ? c->Wp : nullptr, c->rcap / B3_BK, 0, c->ld, pro_planes ? c->Sp : nullptr, c->rcap / NB, c->stats + 4, nS_max));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// prebuilt: W (H*P with the nu column) and Smat (S) are already in place (launch_gather_li); first_done: so is panel 0 (launch_chol_first_spec)
int run_update(pre3_ctx *c, int which_prior, int r, bool dense_R, void *Kt_out_dev, bool prebuilt, bool first_done, bool hp_built)
{
    if (r == 0) {   // update.m:50-55: x_k_k = x_km1_k, p_k_k = p_km1_k
        c->dd_done = 0; c->x_done = false; c->cholp_done = false;      // (a speculative persistent launch found no rows on the device either)
        c->jn_q_valid = false; c->proj_in_cholp = false;               // (... so its consumers left no rows 3..6 behind and its strips projected nothing)
        if (c->kt.pending) cholp_timing_rows(c, 0);
        if (which_prior == PRE3_X_K_KM1) PRE3_HIP(hipMemcpyAsync(c->x_kk, c->x_km1, sizeof(double) * c->n, hipMemcpyDeviceToDevice, c->stream));
        return PRE3_OK;
    }
    int r_pad = round_up(r, NB);
    PRE3_CHECK(r_pad <= c->rcap, PRE3_E_ARG, "update with %d rows exceeds the context capacity %d", r, c->rcap);
    PRE3_TRY(pend_flush(c));                       // (PRE3_OPT_PEND_HI: unless the speculative persistent launch's consumers have taken the pending rows already)
    if (!prebuilt) {
        if (!hp_built) PRE3_TRY(launch_ell_HP(c, r, c->W, true));           // (hp_built: launch_ell_HP_build_sel made the rows and W = H*P in one launch)
        PRE3_TRY(launch_ell_G(c, r, c->W, c->Smat, r_pad, 1, dense_R ? c->Rdense : nullptr));
    }
    // (a stateless update that wants K' back keeps the x-update where it was)
    PRE3_TRY(launch_chol_solve(c, r_pad, first_done, which_prior == PRE3_X_K_KM1, r, which_prior));
    if (c->tail_done && first_done && which_prior == PRE3_X_K_KM1) {
        // the speculative persistent launch has carried everything: factorisation, solve, x-update, rescue stage, the HI update of up to 32
        // landmarks, and ONE down-date of P for both updates (pre3_cholp.hip, CpTail).  The rows / columns 3..6 pass that is left
        // (update.m:42-46 of both updates, params[96..]) is pre3_update_hi's to schedule, once it knows how the rescue stage ended.
        c->dd_done = 0; c->x_done = false; c->split_rows = 0; c->ride_rescue_projection = false; c->proj_with_jnorm = false;
        return PRE3_OK;
    }
    PRE3_TRY(launch_downdate(c, r, c->W, which_prior));
    // update.m:42-46.  leave_jn_to_predict (pre3_step completing the previous step's HI update): the prediction's launch that follows carries it
    if (c->leave_jn_to_predict && !Kt_out_dev && !c->proj_with_jnorm) c->jn_pending = true;
    else PRE3_TRY(launch_jnorm(c, 0));
    if (Kt_out_dev) {
        dim3 g(ceil_div(c->n, 256)), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_gain<double>, g, b, 0, c->stream, c->n, r, (const double *)c->Smat, r_pad, (const double *)c->W, c->ldw, (double *)Kt_out_dev),
            hipLaunchKernelGGL(k_gain<float>, g, b, 0, c->stream, c->n, r, (const float *)c->Smat, r_pad, (const float *)c->W, c->ldw, (float *)Kt_out_dev));
        PRE3_HIP(hipGetLastError());
    }
    return PRE3_OK;
}

#ifdef PRE3_PROBE
extern "C" __attribute__((visibility("default"))) int pre3_debug_k9_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k9), sizeof(unsigned long long) * 64 * 8 * 4) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int pre3_debug_k9rt(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k9rt), sizeof(unsigned long long) * 2048 * 4) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int pre3_debug_rt(int on)
{
    static unsigned long long z[2048 * 4];
    if (on && hipMemcpyToSymbol(HIP_SYMBOL(g_k9rt), z, sizeof z) != hipSuccess) return -3;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rt_on), &on, sizeof on) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int pre3_debug_k9hw(unsigned int *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k9hw), sizeof(unsigned int) * 2048) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int pre3_debug_hf(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hf), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int pre3_debug_probe(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int pre3_debug_k9_clear(void)
{
    static unsigned long long z[64 * 8 * 4];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k9), z, sizeof z) == hipSuccess ? 0 : -3;
}
#endif

}  // namespace pre3
