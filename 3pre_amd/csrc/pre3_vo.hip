// pre3_vo.hip -- SURVEY 8(f)-4: the visual-odometry front end's 4-point 3D-3D RANSAC on the device.
//   vodometry_dr_ye.m:162-236  (adaptive count, winner, final fit, error statistics)
//   ransac_dr_ye.m:13-23,48-72 (point gathering from the range images, inlier radius, per-hypothesis support)
//   find_transform_matrix_dr_ye.m:8-41 (centroids, H = sum q2 q1', svd, V U', reflection handling)
//   R2e.m:21-23, R2q.m, Calculate_V_Omega_RANSAC_dr_ye.m:40-50 (Euler angles and the u = [T; q] the predict kernel consumes)
// One workgroup (one wave) per hypothesis: every lane solves the same 3x3 SVD (uniform control flow, no broadcast),
// then the lanes stride over the matched points; ballots give the inlier bit mask and the support.  All fp64, no
// contraction (the inlier test is a discontinuity; keep the operation order of the reference's loops).
#include <algorithm>
#include <vector>

#include "pre3_internal.h"

namespace pre3 {

struct VoOut {                 // device-side result block
    double rot[9], trans[3], euler[3], u[7];
    double error_mean, error_std, dist;
    int32_t sta, n_support, n_iterations, best, dist_ok, pad[3];
};

// svd of a 3x3 by one-sided Jacobi; returns U, sv, V with H = U diag(sv) V'
__device__ void vo_svd3(const double *H, double *U, double *sv, double *V)
{
#pragma clang fp contract(off)
    double A[9];
    for (int i = 0; i < 9; ++i) { A[i] = H[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; U[i] = 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { alpha += A[3 * i + p] * A[3 * i + p]; beta += A[3 * i + q] * A[3 * i + q]; gamma += A[3 * i + p] * A[3 * i + q]; }
            if (gamma == 0.0 || fabs(gamma) <= 2.2e-16 * sqrt(alpha * beta)) continue;
            rotated = 1;
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double ap = A[3 * i + p], aq = A[3 * i + q];
                A[3 * i + p] = c * ap - s * aq; A[3 * i + q] = s * ap + c * aq;
                const double vp = V[3 * i + p], vq = V[3 * i + q];
                V[3 * i + p] = c * vp - s * vq; V[3 * i + q] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    int ok[3];
    double big = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) { sv[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]); big = sv[j] > big ? sv[j] : big; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ok[j] = sv[j] > 1e-300 && sv[j] > 1e-18 * big;
        if (ok[j]) for (int i = 0; i < 3; ++i) U[3 * i + j] = A[3 * i + j] / sv[j];
    }
    const int nok = ok[0] + ok[1] + ok[2];
    if (nok == 2) {                       // orthonormal completion for a zero singular value
        const int j = !ok[0] ? 0 : (!ok[1] ? 1 : 2), a = (j + 1) % 3, b = (j + 2) % 3;
        const double ua[3] = { U[a], U[3 + a], U[6 + a] }, ub[3] = { U[b], U[3 + b], U[6 + b] };
        const double w[3] = { ua[1] * ub[2] - ua[2] * ub[1], ua[2] * ub[0] - ua[0] * ub[2], ua[0] * ub[1] - ua[1] * ub[0] };
        for (int i = 0; i < 3; ++i) U[3 * i + j] = w[i];
    } else if (nok < 2) {
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (nok == 1) {
            const int j = ok[0] ? 0 : (ok[1] ? 1 : 2), a = (j + 1) % 3, b = (j + 2) % 3;
            const double u[3] = { A[j] / sv[j], A[3 + j] / sv[j], A[6 + j] / sv[j] };
            const int m = fabs(u[0]) < fabs(u[1]) ? (fabs(u[0]) < fabs(u[2]) ? 0 : 2) : (fabs(u[1]) < fabs(u[2]) ? 1 : 2);
            double e[3] = { 0, 0, 0 }; e[m] = 1;
            double w[3] = { u[1] * e[2] - u[2] * e[1], u[2] * e[0] - u[0] * e[2], u[0] * e[1] - u[1] * e[0] };
            const double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            for (int i = 0; i < 3; ++i) w[i] /= nw;
            const double x[3] = { u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0] };
            for (int i = 0; i < 3; ++i) { U[3 * i + j] = u[i]; U[3 * i + a] = w[i]; U[3 * i + b] = x[i]; }
        }
    }
}

// find_transform_matrix_dr_ye.m:19-44 from the centroids and H
__device__ int vo_solve(const double *H, const double *ct1, const double *ct2, double *rot, double *trans)
{
#pragma clang fp contract(off)
    double U[9], sv[3], V[9], Xq[9];
    vo_svd3(H, U, sv, V);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Xq[3 * i + j] = V[3 * i] * U[3 * j] + V[3 * i + 1] * U[3 * j + 1] + V[3 * i + 2] * U[3 * j + 2];
    const double mdet = Xq[0] * (Xq[4] * Xq[8] - Xq[5] * Xq[7]) - Xq[1] * (Xq[3] * Xq[8] - Xq[5] * Xq[6]) + Xq[2] * (Xq[3] * Xq[7] - Xq[4] * Xq[6]);
    int state;
    if (round(mdet) == 1) state = 1;
    else if (round(mdet) == -1) {
        int zn = -1, cnt = 0;
        for (int j = 0; j < 3; ++j) if (fabs(sv[j]) < 0.00000000000001) { zn = j; ++cnt; }
        if (cnt == 1) {
            for (int i = 0; i < 3; ++i) V[3 * i + zn] = -V[3 * i + zn];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Xq[3 * i + j] = V[3 * i] * U[3 * j] + V[3 * i + 1] * U[3 * j + 1] + V[3 * i + 2] * U[3 * j + 2];
            state = 2;
        } else state = -1;
    } else state = 0;
    if (state >= 1) {
        for (int i = 0; i < 9; ++i) rot[i] = Xq[i];
        for (int i = 0; i < 3; ++i) trans[i] = ct1[i] - (rot[3 * i] * ct2[0] + rot[3 * i + 1] * ct2[1] + rot[3 * i + 2] * ct2[2]);
    } else {
        for (int i = 0; i < 9; ++i) rot[i] = H[i];
        trans[0] = trans[1] = trans[2] = 0;
    }
    return state;
}

// ransac_dr_ye.m:13-19 for one frame: pset(:,i) = [-x(ROW,COL); -y(ROW,COL); z(ROW,COL)]
__global__ void k_vo_gather(int rows, int cols, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                            int ldf, const double *__restrict__ frm, int K, int pnum, const double *__restrict__ sel, int sel_stride,
                            double *__restrict__ pset, int32_t *__restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pnum) return;
    const int k = (int)sel[(size_t)i * sel_stride] - 1;
    if (k < 0 || k >= K) { atomicOr(bad, 1); return; }
    const int COL = (int)round(frm[(size_t)ldf * k]), ROW = (int)round(frm[(size_t)ldf * k + 1]);
    if (ROW < 1 || ROW > rows || COL < 1 || COL > cols) { atomicOr(bad, 2); return; }
    const size_t o = (size_t)(COL - 1) * rows + (ROW - 1);
    pset[3 * i] = -x[o]; pset[3 * i + 1] = -y[o]; pset[3 * i + 2] = z[o];
}

// ransac_dr_ye.m:20-23 -- one wave
__global__ void k_vo_dist(int pnum, const double *__restrict__ pset2, VoOut *__restrict__ out)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    double mz = INFINITY;
    for (int k = lane; k < pnum; k += 64) {
        const double a = pset2[3 * k], b = pset2[3 * k + 1], c = pset2[3 * k + 2];
        const double nr = sqrt(c * c + b * b + a * a);
        if (nr > 0.4 && c < mz) mz = c;
    }
    for (int o = 32; o > 0; o >>= 1) { const double v = __shfl_xor(mz, o, 64); mz = v < mz ? v : mz; }
    int first = 0x7fffffff;
    for (int k = lane; k < pnum; k += 64) if (pset2[3 * k + 2] == mz && k < first) first = k;
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(first, o, 64); first = v < first ? v : first; }
    if (lane == 0) {
        const bool ok = mz < INFINITY && first < pnum;
        out->dist_ok = ok;
        out->dist = ok ? sqrt(pset2[3 * first] * pset2[3 * first] + pset2[3 * first + 1] * pset2[3 * first + 1] + pset2[3 * first + 2] * pset2[3 * first + 2]) : 0.0;
    }
}

// ransac_dr_ye.m:48-72 for hypothesis blockIdx.x
__global__ __launch_bounds__(64) void k_vo_score(int pnum, const double *__restrict__ pset1, const double *__restrict__ pset2,
                                                 const int32_t *__restrict__ draws, const VoOut *__restrict__ out, int words,
                                                 unsigned long long *__restrict__ masks, int32_t *__restrict__ cnum, int32_t *__restrict__ state)
{
#pragma clang fp contract(off)
    const int hyp = blockIdx.x, lane = threadIdx.x;
    double ct1[3] = { 0, 0, 0 }, ct2[3] = { 0, 0, 0 }, H[9], rot[9], tr[3];
    int d[4];
    for (int s = 0; s < 4; ++s) d[s] = draws[4 * hyp + s];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 3; ++i) { ct1[i] += pset1[3 * d[s] + i]; ct2[i] += pset2[3 * d[s] + i]; }
    for (int i = 0; i < 3; ++i) { ct1[i] /= 4; ct2[i] /= 4; }
    for (int i = 0; i < 9; ++i) H[i] = 0;
    for (int s = 0; s < 4; ++s) {
        double q1[3], q2[3];
        for (int i = 0; i < 3; ++i) { q1[i] = pset1[3 * d[s] + i] - ct1[i]; q2[i] = pset2[3 * d[s] + i] - ct2[i]; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[3 * i + j] += q2[i] * q1[j];
    }
    const int st = vo_solve(H, ct1, ct2, rot, tr);
    const double thr = 0.001 * out->dist;
    int cnt = 0;
    for (int w = 0; w < words; ++w) {
        const int k = w * 64 + lane;
        int in = 0;
        if (k < pnum) {
            double dd = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double v = rot[3 * i] * pset2[3 * k] + rot[3 * i + 1] * pset2[3 * k + 1] + rot[3 * i + 2] * pset2[3 * k + 2];
                v = v + tr[i];
                const double e = v - pset1[3 * k + i];
                dd = dd + e * e;
            }
            in = dd < thr;
        }
        const unsigned long long b = __ballot(in);
        if (lane == 0) masks[(size_t)hyp * words + w] = b;
        cnt += __popcll(b);
    }
    if (lane == 0) { cnum[hyp] = cnt; state[hyp] = st; }
}

// vodometry_dr_ye.m:185-236: winner = first maximum, adaptive count, final fit on its inliers, error statistics -- one wave
__global__ void k_vo_final(int pnum, int n_hyp, const double *__restrict__ pset1, const double *__restrict__ pset2,
                           const int32_t *__restrict__ cnum, int words, const unsigned long long *__restrict__ masks,
                           VoOut *__restrict__ out, int32_t *__restrict__ inl_out)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    int bc = -1, bi = 0x7fffffff;
    for (int i = lane; i < n_hyp; i += 64) { const int c = cnum[i]; if (c > bc) { bc = c; bi = i; } }
    for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_xor(bc, o, 64), oi = __shfl_xor(bi, o, 64);
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    // nIterations is overwritten at every strict improvement, so its final value belongs to the global maximum (:185-188)
    double nIter = n_hyp;
    if (bc > 0) nIter = 5 * ceil(log(0.01) / log(1 - pow((double)bc / pnum, 4)));
    const int n_it = (int)(nIter < n_hyp ? nIter : n_hyp);
    for (int k = lane; k < pnum; k += 64) inl_out[k] = bc >= 3 ? (int)((masks[(size_t)bi * words + (k >> 6)] >> (k & 63)) & 1ull) : 0;
    if (bc < 3) {                                                                          // :198-205
        if (lane == 0) { out->sta = 4; out->n_support = bc < 0 ? 0 : bc; out->n_iterations = n_it; out->best = bi;
                         for (int i = 0; i < 9; ++i) out->rot[i] = 0; for (int i = 0; i < 3; ++i) { out->trans[i] = 0; out->euler[i] = 0; }
                         out->error_mean = out->error_std = 0; out->u[0] = out->u[1] = out->u[2] = 0; out->u[3] = 1; out->u[4] = out->u[5] = out->u[6] = 0; }
        return;
    }
    const unsigned long long *mk = masks + (size_t)bi * words;
    // centroids, then H = sum q2 q1' over the inliers (lane-strided partial sums, butterfly-reduced)
    double s[6] = { 0, 0, 0, 0, 0, 0 };
    for (int k = lane; k < pnum; k += 64)
        if ((mk[k >> 6] >> (k & 63)) & 1ull) for (int i = 0; i < 3; ++i) { s[i] += pset1[3 * k + i]; s[3 + i] += pset2[3 * k + i]; }
    for (int o = 32; o > 0; o >>= 1) for (int i = 0; i < 6; ++i) s[i] += __shfl_xor(s[i], o, 64);
    double ct1[3], ct2[3], H[9];
    for (int i = 0; i < 3; ++i) { ct1[i] = s[i] / bc; ct2[i] = s[3 + i] / bc; }
    for (int i = 0; i < 9; ++i) H[i] = 0;
    for (int k = lane; k < pnum; k += 64)
        if ((mk[k >> 6] >> (k & 63)) & 1ull) {
            double q1[3], q2[3];
            for (int i = 0; i < 3; ++i) { q1[i] = pset1[3 * k + i] - ct1[i]; q2[i] = pset2[3 * k + i] - ct2[i]; }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[3 * i + j] += q2[i] * q1[j];
        }
    for (int o = 32; o > 0; o >>= 1) for (int i = 0; i < 9; ++i) H[i] += __shfl_xor(H[i], o, 64);
    double rot[9], tr[3];
    const int sta = vo_solve(H, ct1, ct2, rot, tr);
    // ErrorRANSAC_Norm, mean and (N-1)-normalised std (:222-225)
    double se = 0;
    for (int k = lane; k < pnum; k += 64)
        if ((mk[k >> 6] >> (k & 63)) & 1ull) {
            double s2 = 0;
            for (int i = 0; i < 3; ++i) {
                const double v = rot[3 * i] * pset2[3 * k] + rot[3 * i + 1] * pset2[3 * k + 1] + rot[3 * i + 2] * pset2[3 * k + 2] + tr[i] - pset1[3 * k + i];
                s2 += v * v;
            }
            se += sqrt(s2);
        }
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    const double mean = se / bc;
    double var = 0;
    for (int k = lane; k < pnum; k += 64)
        if ((mk[k >> 6] >> (k & 63)) & 1ull) {
            double s2 = 0;
            for (int i = 0; i < 3; ++i) {
                const double v = rot[3 * i] * pset2[3 * k] + rot[3 * i + 1] * pset2[3 * k + 1] + rot[3 * i + 2] * pset2[3 * k + 2] + tr[i] - pset1[3 * k + i];
                s2 += v * v;
            }
            const double e = sqrt(s2) - mean;
            var += e * e;
        }
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    if (lane == 0) {
        for (int i = 0; i < 9; ++i) out->rot[i] = rot[i];
        for (int i = 0; i < 3; ++i) out->trans[i] = tr[i];
        out->error_mean = mean; out->error_std = bc > 1 ? sqrt(var / (bc - 1)) : 0.0;
        out->sta = sta; out->n_support = bc; out->n_iterations = n_it; out->best = bi;
        out->euler[0] = out->euler[1] = out->euler[2] = 0;
        if (sta >= 1) { out->euler[0] = atan2(rot[7], rot[8]); out->euler[1] = asin(-rot[6]); out->euler[2] = atan2(rot[3], rot[0]); }   // R2e.m:21-23
        // Calculate_V_Omega_RANSAC_dr_ye.m:40-50: u = [T; R2q(R)], identity unless sta == 1
        double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, T3[3] = { 0, 0, 0 };
        if (sta == 1) { for (int i = 0; i < 9; ++i) R[i] = rot[i]; for (int i = 0; i < 3; ++i) T3[i] = tr[i]; }
        const double Tq = R[0] + R[4] + R[8] + 1;
        double a, b, c, d, S;
        if (Tq > 0.00000001) { S = 2 * sqrt(Tq); a = 0.25 * S; b = (R[5] - R[7]) / S; c = (R[6] - R[2]) / S; d = (R[1] - R[3]) / S; }
        else if (R[0] > R[4] && R[0] > R[8]) { S = 2 * sqrt(1.0 + R[0] - R[4] - R[8]); a = (R[5] - R[7]) / S; b = 0.25 * S; c = (R[1] + R[3]) / S; d = (R[6] + R[2]) / S; }
        else if (R[4] > R[8]) { S = 2 * sqrt(1.0 + R[4] - R[0] - R[8]); a = (R[6] - R[2]) / S; b = (R[1] + R[3]) / S; c = 0.25 * S; d = (R[5] + R[7]) / S; }
        else { S = 2 * sqrt(1.0 + R[8] - R[0] - R[4]); a = (R[1] - R[3]) / S; b = (R[6] + R[2]) / S; c = (R[5] + R[7]) / S; d = 0.25 * S; }
        out->u[0] = T3[0]; out->u[1] = T3[1]; out->u[2] = T3[2]; out->u[3] = a; out->u[4] = -b; out->u[5] = -c; out->u[6] = -d;
    }
}

struct DevMem {                    // pooled device scratch (pre3_match.hip): no hipMalloc / hipFree per call once warm
    void *p = nullptr;
    int slot = -1;
    ~DevMem() { scratch_release(slot, p); }
    int alloc(size_t bytes) { return scratch_acquire(bytes, &p, &slot); }
    template <typename T> T *as() { return (T *)p; }
};

static int vo_run(int pnum, const double *d_p1, const double *d_p2, int n_hyp, const int32_t *draws, int32_t *cnum_out, int32_t *state_out,
                  int32_t *inlier_out, pre3_vo_result *res, int reps, double *ms_out)
{
    const int words = ceil_div(pnum, 64);
    DevMem dd, dm, dc, ds, dout, dinl;
    PRE3_TRY(dd.alloc(sizeof(int32_t) * 4 * (size_t)n_hyp)); PRE3_TRY(dm.alloc(sizeof(unsigned long long) * (size_t)n_hyp * words));
    PRE3_TRY(dc.alloc(sizeof(int32_t) * n_hyp)); PRE3_TRY(ds.alloc(sizeof(int32_t) * n_hyp)); PRE3_TRY(dout.alloc(sizeof(VoOut)));
    PRE3_TRY(dinl.alloc(sizeof(int32_t) * pnum));
    PRE3_HIP(hipMemcpy(dd.p, draws, sizeof(int32_t) * 4 * (size_t)n_hyp, hipMemcpyHostToDevice));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_out) { PRE3_HIP(hipEventCreate(&e0)); PRE3_HIP(hipEventCreate(&e1)); PRE3_HIP(hipEventRecord(e0, 0)); }
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_vo_dist, dim3(1), dim3(64), 0, 0, pnum, d_p2, dout.as<VoOut>());
        hipLaunchKernelGGL(k_vo_score, dim3(n_hyp), dim3(64), 0, 0, pnum, d_p1, d_p2, dd.as<int32_t>(), dout.as<VoOut>(), words,
                           dm.as<unsigned long long>(), dc.as<int32_t>(), ds.as<int32_t>());
        hipLaunchKernelGGL(k_vo_final, dim3(1), dim3(64), 0, 0, pnum, n_hyp, d_p1, d_p2, dc.as<int32_t>(), words, dm.as<unsigned long long>(),
                           dout.as<VoOut>(), dinl.as<int32_t>());
    }
    int rc = PRE3_OK;
    if (hipGetLastError() != hipSuccess) { set_error("vo: kernel launch failed"); rc = PRE3_E_HIP; }
    if (ms_out && rc == PRE3_OK) {
        float ms = 0;
        if (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { set_error("vo: event timing failed"); rc = PRE3_E_HIP; }
        *ms_out = ms / reps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    PRE3_TRY(rc);
    PRE3_HIP(hipDeviceSynchronize());
    VoOut o;
    PRE3_HIP(hipMemcpy(&o, dout.p, sizeof o, hipMemcpyDeviceToHost));
    PRE3_CHECK(o.dist_ok, PRE3_E_NUMERIC, "vo: no matched point is farther than 0.4 m from the camera (ransac_dr_ye.m:21 has no minimum there)");
    if (cnum_out) PRE3_HIP(hipMemcpy(cnum_out, dc.p, sizeof(int32_t) * n_hyp, hipMemcpyDeviceToHost));
    if (state_out) PRE3_HIP(hipMemcpy(state_out, ds.p, sizeof(int32_t) * n_hyp, hipMemcpyDeviceToHost));
    if (inlier_out) PRE3_HIP(hipMemcpy(inlier_out, dinl.p, sizeof(int32_t) * pnum, hipMemcpyDeviceToHost));
    if (res) {
        memcpy(res->rot, o.rot, sizeof o.rot); memcpy(res->trans, o.trans, sizeof o.trans); memcpy(res->euler, o.euler, sizeof o.euler);
        memcpy(res->u, o.u, sizeof o.u);
        res->error_mean = o.error_mean; res->error_std = o.error_std; res->dist = o.dist;
        res->sta = o.sta; res->n_support = o.n_support; res->n_iterations = o.n_iterations; res->best = o.best;
    }
    return PRE3_OK;
}

static int vo_check(int device, int pnum, int n_hyp, const int32_t *draws)
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_error("no HIP device available (libpre3 has no CPU fallback)"); return PRE3_E_NODEVICE; }
    PRE3_CHECK(pnum >= 4, PRE3_E_ARG, "vo: number of points is smaller than 4: insufficient for ransac");      // ransac_dr_ye.m:5-11
    PRE3_CHECK(n_hyp >= 1 && draws, PRE3_E_ARG, "vo: no hypotheses");
    for (int i = 0; i < 4 * n_hyp; ++i) PRE3_CHECK(draws[i] >= 0 && draws[i] < pnum, PRE3_E_ARG, "vo: draws[%d]=%d is not a match position (pnum=%d)", i, draws[i], pnum);
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return PRE3_E_NODEVICE; }
    return PRE3_OK;
}

}  // namespace pre3

using namespace pre3;

extern "C" {

int pre3_vo_ransac(int device, int pnum, const double *pset1, const double *pset2, int n_hyp, const int32_t *draws, int32_t *cnum_out,
                   int32_t *state_out, int32_t *inlier_out, pre3_vo_result *res)
{
    PRE3_TRY(vo_check(device, pnum, n_hyp, draws));
    PRE3_CHECK(pset1 && pset2, PRE3_E_ARG, "pre3_vo_ransac: null point set");
    DevMem p1, p2;
    PRE3_TRY(p1.alloc(sizeof(double) * 3 * pnum)); PRE3_TRY(p2.alloc(sizeof(double) * 3 * pnum));
    PRE3_HIP(hipMemcpy(p1.p, pset1, sizeof(double) * 3 * pnum, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(p2.p, pset2, sizeof(double) * 3 * pnum, hipMemcpyHostToDevice));
    return vo_run(pnum, p1.as<double>(), p2.as<double>(), n_hyp, draws, cnum_out, state_out, inlier_out, res, 1, nullptr);
}

int pre3_vo_ransac_frames(int device, int rows, int cols, const double *x1, const double *y1, const double *z1, const double *x2,
                          const double *y2, const double *z2, int ldf, int K1, const double *frm1, int K2, const double *frm2, int pnum,
                          const double *match, int n_hyp, const int32_t *draws, double *pset1_out, double *pset2_out, int32_t *cnum_out,
                          int32_t *state_out, int32_t *inlier_out, pre3_vo_result *res)
{
    PRE3_TRY(vo_check(device, pnum, n_hyp, draws));
    PRE3_CHECK(rows > 0 && cols > 0 && x1 && y1 && z1 && x2 && y2 && z2 && frm1 && frm2 && match && ldf >= 2 && K1 > 0 && K2 > 0, PRE3_E_ARG,
               "pre3_vo_ransac_frames: bad arguments");
    const size_t img = sizeof(double) * (size_t)rows * cols;
    DevMem im, f1, f2, mt, p1, p2, bad;
    PRE3_TRY(im.alloc(6 * img)); PRE3_TRY(f1.alloc(sizeof(double) * (size_t)ldf * K1)); PRE3_TRY(f2.alloc(sizeof(double) * (size_t)ldf * K2));
    PRE3_TRY(mt.alloc(sizeof(double) * 2 * pnum)); PRE3_TRY(p1.alloc(sizeof(double) * 3 * pnum)); PRE3_TRY(p2.alloc(sizeof(double) * 3 * pnum));
    PRE3_TRY(bad.alloc(sizeof(int32_t)));
    const double *src[6] = { x1, y1, z1, x2, y2, z2 };
    for (int i = 0; i < 6; ++i) PRE3_HIP(hipMemcpy((char *)im.p + i * img, src[i], img, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(f1.p, frm1, sizeof(double) * (size_t)ldf * K1, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(f2.p, frm2, sizeof(double) * (size_t)ldf * K2, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(mt.p, match, sizeof(double) * 2 * pnum, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemset(bad.p, 0, sizeof(int32_t)));
    const double *I = im.as<double>();
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(k_vo_gather, dim3(ceil_div(pnum, 64)), dim3(64), 0, 0, rows, cols, I, I + n, I + 2 * n, ldf, f1.as<double>(), K1, pnum,
                       mt.as<double>(), 2, p1.as<double>(), bad.as<int32_t>());
    hipLaunchKernelGGL(k_vo_gather, dim3(ceil_div(pnum, 64)), dim3(64), 0, 0, rows, cols, I + 3 * n, I + 4 * n, I + 5 * n, ldf, f2.as<double>(), K2, pnum,
                       mt.as<double>() + 1, 2, p2.as<double>(), bad.as<int32_t>());
    PRE3_HIP(hipGetLastError());
    int32_t b = 0;
    PRE3_HIP(hipMemcpy(&b, bad.p, sizeof b, hipMemcpyDeviceToHost));
    PRE3_CHECK(b == 0, PRE3_E_ARG, "pre3_vo_ransac_frames: %s", (b & 1) ? "a match refers to a keypoint that does not exist" : "a keypoint rounds to a pixel outside the range image");
    if (pset1_out) PRE3_HIP(hipMemcpy(pset1_out, p1.p, sizeof(double) * 3 * pnum, hipMemcpyDeviceToHost));
    if (pset2_out) PRE3_HIP(hipMemcpy(pset2_out, p2.p, sizeof(double) * 3 * pnum, hipMemcpyDeviceToHost));
    return vo_run(pnum, p1.as<double>(), p2.as<double>(), n_hyp, draws, cnum_out, state_out, inlier_out, res, 1, nullptr);
}

// measurement only: kernels of one RANSAC (dist + score + final) on resident inputs, averaged over reps
int pre3_vo_bench(int device, int pnum, const double *pset1, const double *pset2, int n_hyp, const int32_t *draws, int reps, double *ms_per_call)
{
    PRE3_TRY(vo_check(device, pnum, n_hyp, draws));
    PRE3_CHECK(pset1 && pset2 && reps >= 1 && ms_per_call, PRE3_E_ARG, "pre3_vo_bench: bad arguments");
    DevMem p1, p2;
    PRE3_TRY(p1.alloc(sizeof(double) * 3 * pnum)); PRE3_TRY(p2.alloc(sizeof(double) * 3 * pnum));
    PRE3_HIP(hipMemcpy(p1.p, pset1, sizeof(double) * 3 * pnum, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(p2.p, pset2, sizeof(double) * 3 * pnum, hipMemcpyHostToDevice));
    PRE3_TRY(vo_run(pnum, p1.as<double>(), p2.as<double>(), n_hyp, draws, nullptr, nullptr, nullptr, nullptr, 3, nullptr));   // warm-up
    return vo_run(pnum, p1.as<double>(), p2.as<double>(), n_hyp, draws, nullptr, nullptr, nullptr, nullptr, reps, ms_per_call);
}

}  // extern "C"
