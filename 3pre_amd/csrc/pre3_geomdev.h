// pre3_geomdev.h -- device-side camera geometry shared by pre3_geom.hip and pre3_update.hip (the projection also rides inside
// the predict and K9 launches, see ProjRide).  fp64 throughout.
#pragma once
#include "pre3_internal.h"

namespace pre3 {

struct CamD { double f, Cx, Cy, k1, k2, nRows, nCols; };
struct U7 { double v[7]; };

// ------------------------------------------------------------------------------------------------
// device math (fp64)
// ------------------------------------------------------------------------------------------------

// q2r.m:29-36
__device__ inline void d_q2r(const double *q, double *R)
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = r * r + x * x - y * y - z * z; R[1] = 2 * (x * y - r * z);           R[2] = 2 * (z * x + r * y);
    R[3] = 2 * (x * y + r * z);           R[4] = r * r - x * x + y * y - z * z; R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (z * x - r * y);           R[7] = 2 * (y * z + r * x);           R[8] = r * r - x * x - y * y + z * z;
}

// slamToolbox .../Rotations/q2R.m:18-34
__device__ inline void d_q2R_sola(const double *q, double *R)
{
    double a = q[0], b = q[1], c = q[2], d = q[3];
    double aa = a * a, ab = 2 * a * b, ac = 2 * a * c, ad = 2 * a * d;
    double bb = b * b, bc = 2 * b * c, bd = 2 * b * d, cc = c * c, cd = 2 * c * d, dd = d * d;
    R[0] = aa + bb - cc - dd; R[1] = bc - ad;           R[2] = bd + ac;
    R[3] = bc + ad;           R[4] = aa - bb + cc - dd; R[5] = cd - ab;
    R[6] = bd - ac;           R[7] = cd + ab;           R[8] = aa - bb - cc + dd;
}

// hu_my_version.m:41-42 + distort_fm_my_version.m:52-61
__device__ inline void d_pinhole_distort(const double *hrl, const CamD &cam, double *uvd)
{
    double uu = cam.Cx + (hrl[0] / hrl[2]) * cam.f;
    double vu = cam.Cy + (hrl[1] / hrl[2]) * cam.f;
    double xu = (uu - cam.Cx) / cam.f, yu = (vu - cam.Cy) / cam.f;
    double ru = sqrt(xu * xu + yu * yu);
    double r2 = ru * ru;
    double D = 1 + cam.k1 * r2 + cam.k2 * (r2 * r2);
    uvd[0] = xu * D * cam.f + cam.Cx;
    uvd[1] = yu * D * cam.f + cam.Cy;
}

// direction vector of a landmark in the world frame before rotation: (y-r)*rho + m(theta,phi)  or  y-r
// (sc: sin / cos of theta and phi, given by a caller that needs them again -- project_core's Jacobian: a double-precision sincos is ~1000 shader
//  cycles, and the four of them were all but the whole cost of a projection, tools/probe_project.hip)
__device__ inline void d_ray(int type, const double *y, const double *t, double *v, const double *sc = nullptr)
{
    if (type == PRE3_INVDEPTH) {
        double sth, cth, sphi, cphi;                       // one shared range reduction per angle instead of four separate calls
        if (sc) { sth = sc[0]; cth = sc[1]; sphi = sc[2]; cphi = sc[3]; }
        else { sincos(y[3], &sth, &cth); sincos(y[4], &sphi, &cphi); }
        double mi0 = cphi * sth, mi1 = -sphi, mi2 = cphi * cth;   // m.m:38-40
        v[0] = (y[0] - t[0]) * y[5] + mi0;
        v[1] = (y[1] - t[1]) * y[5] + mi1;
        v[2] = (y[2] - t[2]) * y[5] + mi2;
    } else {
        v[0] = y[0] - t[0]; v[1] = y[1] - t[1]; v[2] = y[2] - t[2];
    }
}

// ------------------------------------------------------------------------------------------------
// K1 predict: x_k_km1 from x_k_k and u; P rows/cols 3..6 and the 7x7 pose block in place.
// F = blkdiag(I3, Qq1, I6) and Jnorm = blkdiag(I3, Jn, I..) only touch rows/cols 3..6, so the
// reference's full 13 x n products reduce to a 4 x n strip (multiplications by 1/0 are exact).
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// K2 project + Jacobian of one landmark (predict_camera_measurements.m:27-68, calculate_Hi_*_my_version.m)
// ------------------------------------------------------------------------------------------------
// project_core: the arithmetic, on a pose x[0..6] and the landmark's own entries y[0..5]; outputs in the caller's arrays (hc_out [14], hl_out [12]).
// had / h_old: a prediction kept from before (quirk Q7).  Returns "has a prediction now"; zi = the prediction the Jacobian was taken at; fresh = it is new.
__device__ inline bool project_core(const int type, const double *__restrict__ x, const double *__restrict__ y, const CamD &cam, const int had, const double *h_old,
                                    double *zi, bool &fresh, double *__restrict__ hc_out, double *__restrict__ hl_out)
{
    double Rwc[9];
    d_q2r(x + 3, Rwc);
    double v[3], hrl[3];
    double tsc[4] = { 0, 1, 0, 1 };                        // sin / cos of theta, phi: once for the ray and the Jacobian (the same values either way)
    if (type == PRE3_INVDEPTH) { sincos(y[3], &tsc[0], &tsc[1]); sincos(y[4], &tsc[2], &tsc[3]); }
    d_ray(type, y, x, v, tsc);
    // r_cw = r_wc' (hi_inverse_depth.m:33); hi_cartesian.m:33 uses inv(r_wc) = r_wc' to rounding
    for (int c = 0; c < 3; ++c) hrl[c] = Rwc[0 * 3 + c] * v[0] + Rwc[1 * 3 + c] * v[1] + Rwc[2 * 3 + c] * v[2];
    const double PI = 3.141592653589793238462643383279502884;
    double ax = atan2(hrl[0], hrl[2]) * 180 / PI, ay = atan2(hrl[1], hrl[2]) * 180 / PI;
    bool ok = !(ax < -60 || ax > 60 || ay < -60 || ay > 60);
    double uvd[2] = { 0, 0 };
    if (ok) {
        d_pinhole_distort(hrl, cam, uvd);
        ok = (uvd[0] > 0) && (uvd[0] < cam.nCols) && (uvd[1] > 0) && (uvd[1] < cam.nRows);
    }
    fresh = ok;
    if (ok) { zi[0] = uvd[0]; zi[1] = uvd[1]; }
    else if (had) { zi[0] = h_old[0]; zi[1] = h_old[1]; }          // stale h kept (quirk Q7)
    const bool now = ok || had;
    if (!now) return false;
    // ---- Jacobian (calculate_Hi_*_my_version.m); distortion Jacobian at the stored h (quirk Q8)
    double u_ = zi[0], v_ = zi[1];
    double xx = u_ - cam.Cx, yy = v_ - cam.Cy, f2 = cam.f * cam.f;
    double r2 = (xx * xx + yy * yy) / f2, r4 = r2 * r2;
    double g = cam.k1 + 2 * cam.k2 * r2, D0 = 1 + cam.k1 * r2 + cam.k2 * r4;
    double Jd[4] = { D0 + xx * g * (2 * xx / f2), xx * g * (2 * yy / f2), yy * g * (2 * xx / f2), D0 + yy * g * (2 * yy / f2) };
    // hc = Rrw * a  with Rrw = inv(q2r(q)) = q2r(q)' to rounding
    double hc[3] = { hrl[0], hrl[1], hrl[2] };
    double f = cam.f;
    double dhu[6] = { f / hc[2], 0, -hc[0] * f / (hc[2] * hc[2]),  0, f / hc[2], -hc[1] * f / (hc[2] * hc[2]) };
    double A[6];   // dh_dhrl = dhd_dhu * dhu_dhrl (2x3)
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) A[r * 3 + c] = Jd[r * 2] * dhu[c] + Jd[r * 2 + 1] * dhu[3 + c];
    double Rrw[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rrw[r * 3 + c] = Rwc[c * 3 + r];
    double sc = (type == PRE3_INVDEPTH) ? y[5] : 1.0;
    // dh_drw = A * (-Rrw*rho)
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            hc_out[r * 7 + c] = A[r * 3] * (-Rrw[c] * sc) + A[r * 3 + 1] * (-Rrw[3 + c] * sc) + A[r * 3 + 2] * (-Rrw[6 + c] * sc);
    // dhrl_dqwr = dRq_times_a_by_dq(qconj(q), a) * diag(1,-1,-1,-1)   (dRq_times_a_by_dq.m:29-101)
    double q0 = x[3], qx = -x[4], qy = -x[5], qz = -x[6];
    double a0 = v[0], a1 = v[1], a2 = v[2];
    double dq[12];
    dq[0] = 2 * q0 * a0 - 2 * qz * a1 + 2 * qy * a2;  dq[4] = 2 * qz * a0 + 2 * q0 * a1 - 2 * qx * a2;  dq[8]  = -2 * qy * a0 + 2 * qx * a1 + 2 * q0 * a2;
    dq[1] = 2 * qx * a0 + 2 * qy * a1 + 2 * qz * a2;  dq[5] = 2 * qy * a0 - 2 * qx * a1 - 2 * q0 * a2;  dq[9]  = 2 * qz * a0 + 2 * q0 * a1 - 2 * qx * a2;
    dq[2] = -2 * qy * a0 + 2 * qx * a1 + 2 * q0 * a2; dq[6] = 2 * qx * a0 + 2 * qy * a1 + 2 * qz * a2;  dq[10] = -2 * q0 * a0 + 2 * qz * a1 - 2 * qy * a2;
    dq[3] = -2 * qz * a0 - 2 * q0 * a1 + 2 * qx * a2; dq[7] = 2 * q0 * a0 - 2 * qz * a1 + 2 * qy * a2;  dq[11] = 2 * qx * a0 + 2 * qy * a1 + 2 * qz * a2;
    for (int r = 0; r < 3; ++r) { dq[r * 4 + 1] = -dq[r * 4 + 1]; dq[r * 4 + 2] = -dq[r * 4 + 2]; dq[r * 4 + 3] = -dq[r * 4 + 3]; }
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 4; ++c)
            hc_out[r * 7 + 3 + c] = A[r * 3] * dq[c] + A[r * 3 + 1] * dq[4 + c] + A[r * 3 + 2] * dq[8 + c];
    for (int t = 0; t < 12; ++t) hl_out[t] = 0;
    if (type == PRE3_INVDEPTH) {
        double lambda = y[5];
        const double sth = tsc[0], cth = tsc[1], sph = tsc[2], cph = tsc[3];
        double dth[3] = { cph * cth, 0, -cph * sth };
        double dph[3] = { -sph * sth, -cph, -sph * cth };
        double d3[3] = { y[0] - x[0], y[1] - x[1], y[2] - x[2] };
        double B[18];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) B[r * 6 + c] = lambda * Rrw[r * 3 + c];
            B[r * 6 + 3] = Rrw[r * 3] * dth[0] + Rrw[r * 3 + 1] * dth[1] + Rrw[r * 3 + 2] * dth[2];
            B[r * 6 + 4] = Rrw[r * 3] * dph[0] + Rrw[r * 3 + 1] * dph[1] + Rrw[r * 3 + 2] * dph[2];
            B[r * 6 + 5] = Rrw[r * 3] * d3[0] + Rrw[r * 3 + 1] * d3[1] + Rrw[r * 3 + 2] * d3[2];
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 6; ++c) hl_out[r * 6 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[6 + c] + A[r * 3 + 2] * B[12 + c];
    } else {
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) hl_out[r * 6 + c] = A[r * 3] * Rrw[c] + A[r * 3 + 1] * Rrw[3 + c] + A[r * 3 + 2] * Rrw[6 + c];
    }
    return true;
}

__device__ inline void project_one(const int i, const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                            const double *__restrict__ x, const CamD &cam, int clear_first,
                            double *__restrict__ h, int32_t *__restrict__ has_h, double *__restrict__ Hc, double *__restrict__ Hl,
                            const double *__restrict__ x_lm = nullptr /* the landmark part of the state, if it lives in another vector */)
{
    const double *y = (x_lm ? x_lm : x) + lm_off[i];
    const int had = clear_first ? 0 : has_h[i];
    double h_old[2] = { 0, 0 };
    if (had) { h_old[0] = h[2 * i]; h_old[1] = h[2 * i + 1]; }
    double zi[2] = { 0, 0 };
    bool fresh = false;
    const bool now = project_core(lm_type[i], x, y, cam, had, h_old, zi, fresh, Hc + 14 * i, Hl + 12 * i);
    if (fresh) { h[2 * i] = zi[0]; h[2 * i + 1] = zi[1]; }
    has_h[i] = now ? 1 : 0;
}


// ---- riders: work that depends on a few producer workgroups of the SAME launch -----------------------------------------
// A kernel boundary costs ~5 us here, more than the projection itself hides: the landmarks are projected by extra workgroups
// appended to the launch that produces the state they need (k_predict for the IC search, the K9 launch -- whose x-update
// workgroups write x_k_k -- for the rescue).  Producers signal a device counter (monotonic, the host passes the target);
// riders have higher block indices, so every producer has been dispatched before a rider can spin.
// The predicted pose (predict_state_and_covariance.m:60-77: r + R(q) u_r, q x u_q, normalised), ONE function for the prediction's block 0 and for
// the projection riders of the same launch, which compute it themselves instead of waiting for block 0 to publish it (round 5: fence + counter +
// poll were ~1.5 us in front of every projection).  The same expressions, so the same bits as x_out.
__device__ __forceinline__ void predict_pose(const double *__restrict__ x_in, const U7 &u, double (&xo)[7], double (&pose)[7])
{
    const double *q = x_in + 3;
    // qProd.m:16-33
    const double a = q[0], b = q[1], c = q[2], d = q[3];
    const double w = u.v[3], x = u.v[4], y = u.v[5], z = u.v[6];
    xo[3] = a * w - b * x - c * y - d * z;
    xo[4] = a * x + b * w + c * z - d * y;
    xo[5] = a * y - b * z + c * w + d * x;
    xo[6] = a * z + b * y - c * x + d * w;
    double R[9];
    d_q2R_sola(q, R);
    for (int i = 0; i < 3; ++i) xo[i] = x_in[i] + (R[i * 3] * u.v[0] + R[i * 3 + 1] * u.v[1] + R[i * 3 + 2] * u.v[2]);
    const double nq = sqrt(xo[3] * xo[3] + xo[4] * xo[4] + xo[5] * xo[5] + xo[6] * xo[6]);
    for (int i = 0; i < 3; ++i) pose[i] = xo[i];
    for (int i = 0; i < 4; ++i) pose[3 + i] = xo[3 + i] / nq;
}

struct ProjRide {
    int n_blocks;                       // 0: no rider in this launch
    int own_pose;                       // 1: the riders compute the predicted pose themselves from (x_prev, u) -- no producer to wait for
    const double *x_prev; U7 u;
    int N, clear_first;
    const int32_t *lm_type, *lm_off; const double *x; const double *x_lm /* landmark entries (null: x) */; CamD cam;
    double *h; int32_t *has_h; double *Hc, *Hl;
    unsigned int *ctr; unsigned int target;
    int32_t *guard;                     // device error word (stats[7]): set when a wait gives up
};

// every device-side wait on another workgroup is bounded: HIP promises no dispatch order, so a counter that never arrives must not hang
// the GPU.  ~2^21 polls with s_sleep in between are of the order of a second; the host then reports PRE3_E_HIP (pre3_api.hip, fetch_stats).
constexpr int SPIN_LIMIT = 1 << 21;
__device__ __forceinline__ bool bounded_wait(const unsigned int *ctr, unsigned int target, int32_t *guard)
{
    for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
        if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    if (guard) atomicExch(guard, 1);
    return false;
}

__device__ __forceinline__ void ride_signal(unsigned int *ctr)      // call from every thread of a producer workgroup
{
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

__device__ __forceinline__ void proj_ride_block(const ProjRide &pr, int blk)
{
    const int i = blk * 64 + threadIdx.x;
    if (pr.own_pose) {
        if (threadIdx.x < 64 && i < pr.N) {
            double xo[7], pose[7];
            predict_pose(pr.x_prev, pr.u, xo, pose);
            project_one(i, pr.lm_type, pr.lm_off, pose, pr.cam, pr.clear_first, pr.h, pr.has_h, pr.Hc, pr.Hl, pr.x_lm);
        }
        return;
    }
    if (threadIdx.x == 0) bounded_wait(pr.ctr, pr.target, pr.guard);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x < 64 && i < pr.N) project_one(i, pr.lm_type, pr.lm_off, pr.x, pr.cam, pr.clear_first, pr.h, pr.has_h, pr.Hc, pr.Hl, pr.x_lm);
}

// ---- S_i = H_i P H_i' (+ I) per landmark (search_IC_matches.m:36; mode 1: the chi2 gate of rescue_hi_inliers.m:35-46).  Shared by k_innovation
// (pre3_geom.hip) and by the rider blocks of k_ell_HP_build (pre3_update.hip: inside pre3_step the S_i pass has no launch of its own).
template <typename T>
__device__ __forceinline__ void innovation_body(int N, const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                                                const T *__restrict__ P, int ld, const double *Hc, const double *Hl,
                                                const int32_t *has_h, int mode, double chi2,
                                                const double *h, const double *__restrict__ z,
                                                const int32_t *__restrict__ ic, const int32_t *__restrict__ li, int32_t *__restrict__ hi,
                                                double *__restrict__ S, int32_t *__restrict__ has_S, const int gt /* global lane: 16 per landmark */,
                                                const float *__restrict__ pend_W = nullptr, const int pend_ldw = 0, const int pend_rows = 0)
{
    // 16 lanes per landmark: lane b < 13 owns column b of the gathered 13x13 block of P (7 pose + 6 landmark
    // entries; P is symmetric, so the column is read as a row: two contiguous runs), then a 16-lane shuffle sum.
    const int i = gt >> 4, b = gt & 15;
    const bool valid = i < N;
    const int ii = valid ? i : 0;
    bool active = valid;
    if (mode == 0) active = active && has_h[ii];
    else active = active && (ic[ii] == 1 && li[ii] == 0);
    const int d = lm_type[ii] == PRE3_INVDEPTH ? 6 : 3;
    const int off = lm_off[ii];
    const int nn = 7 + d;
    double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    if (active && b < nn) {
        const int ib = b < 7 ? b : off + b - 7;
        const T *prow = P + (size_t)ib * ld;
        double hp0 = 0, hp1 = 0;
#pragma unroll
        for (int a = 0; a < 7; ++a) { double p = (double)prow[a]; hp0 += Hc[14 * ii + a] * p; hp1 += Hc[14 * ii + 7 + a] * p; }
#pragma unroll
        for (int a = 0; a < 6; ++a)
            if (a < d) { double p = (double)prow[off + a]; hp0 += Hl[12 * ii + a] * p; hp1 += Hl[12 * ii + 6 + a] * p; }
        const double h0b = b < 7 ? Hc[14 * ii + b] : Hl[12 * ii + b - 7];
        const double h1b = b < 7 ? Hc[14 * ii + 7 + b] : Hl[12 * ii + 6 + b - 7];
        s00 = hp0 * h0b; s01 = hp0 * h1b; s10 = hp1 * h0b; s11 = hp1 * h1b;
    }
    if (pend_rows > 0 && active) {
        // P is P - W~'W~ (PendW): S_i -= g g', g(c, k) = H_i(c, :) W~(k, :)'; lane b takes the rows k = b (mod 16), its share leaves with the sums below
        double hc0[7], hc1[7], hl0[6], hl1[6];                        // (the landmark's rows once, not per pending row)
#pragma unroll
        for (int a = 0; a < 7; ++a) { hc0[a] = Hc[14 * ii + a]; hc1[a] = Hc[14 * ii + 7 + a]; }
#pragma unroll
        for (int a = 0; a < 6; ++a) { hl0[a] = a < d ? Hl[12 * ii + a] : 0.0; hl1[a] = a < d ? Hl[12 * ii + 6 + a] : 0.0; }
        for (int k = b; k < pend_rows; k += 16) {
            const float *wr = pend_W + (size_t)k * pend_ldw;
            float wp[7], wl[6];
#pragma unroll
            for (int a = 0; a < 7; ++a) wp[a] = wr[a];
#pragma unroll
            for (int a = 0; a < 6; ++a) wl[a] = a < d ? wr[off + a] : 0.f;
            double g0 = 0, g1 = 0;
#pragma unroll
            for (int a = 0; a < 7; ++a) { const double w = (double)wp[a]; g0 += hc0[a] * w; g1 += hc1[a] * w; }
#pragma unroll
            for (int a = 0; a < 6; ++a)
                if (a < d) { const double w = (double)wl[a]; g0 += hl0[a] * w; g1 += hl1[a] * w; }
            s00 -= g0 * g0; s01 -= g0 * g1; s10 -= g1 * g0; s11 -= g1 * g1;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s00 += __shfl_xor(s00, o, 16); s01 += __shfl_xor(s01, o, 16);
        s10 += __shfl_xor(s10, o, 16); s11 += __shfl_xor(s11, o, 16);
    }
    if (!valid || b != 0) return;
    if (mode == 0) {
        if (!active) { has_S[i] = 0; return; }
        S[4 * i + 0] = s00 + 1; S[4 * i + 1] = s01; S[4 * i + 2] = s10; S[4 * i + 3] = s11 + 1;
        has_S[i] = 1;
    } else {
        if (!active) return;
        double det = s00 * s11 - s01 * s10;
        double i00 = s11 / det, i01 = -s01 / det, i10 = -s10 / det, i11 = s00 / det;
        double n0 = z[2 * i] - h[2 * i], n1 = z[2 * i + 1] - h[2 * i + 1];
        double t0 = n0 * i00 + n1 * i10, t1 = n0 * i01 + n1 * i11;
        double d2 = t0 * n0 + t1 * n1;
        hi[i] = d2 < chi2 ? 1 : 0;
    }
}

// The per-step inbox [meas | ic | hyp | z]: pinned host memory, read over PCIe by the device itself (16 bytes per lane).  ONE workgroup, so
// that its last act can be to publish `seq` in the pinned mailbox: the host may overwrite the inbox once it reads that number back.
struct InboxRide { const int4 *src; int4 *dst; int n16; int32_t *mail; int32_t seq; int slot = 10; int32_t *clear = nullptr; int n_clear = 0; };      // clear: n_clear ints zeroed on the way (the inlier flags behind the inbox: was a hipMemsetAsync, i.e. a fill kernel between barrier packets)        // n16 == 0: no pull in this launch; slot: the mailbox word that takes seq
__device__ __forceinline__ void inbox_pull_block(const InboxRide &ib)
{
    // four PCIe reads in flight per lane (a read is ~1.5 us; one after the other they made this block the long pole of k_predict)
    for (int i0 = threadIdx.x; i0 < ib.n16; i0 += 4 * blockDim.x) {
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * blockDim.x; v[u] = i < ib.n16 ? ib.src[i] : int4{ 0, 0, 0, 0 }; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * blockDim.x; if (i < ib.n16) ib.dst[i] = v[u]; }
    }
    for (int i = threadIdx.x; i < ib.n_clear; i += blockDim.x) ib.clear[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(ib.mail + ib.slot, ib.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// PRE3_OPT_PEND_HI: the HI update's down-date P - W~'W~ (update.m:37-38 of ekf_update_hi_inliers.m) left pending across the step boundary -- W~ (rows x ldw,
// f32; column ld = L^-1 nu) and its bf16 planes in k_downdate_b3's layout.  rows == 0: nothing pending.  Whoever reads P while it is pending reads
// P - W~'W~ (k_predict transforms W~ with P; k_ell_HP_build_mb and the S_i pass subtract the rank-`rows` term; k_cholp's consumers take W~ as the
// panels in front of panel 0) or runs behind pend_flush().
struct PendW { float *W; int ldw, rows; void *Wp; int nst_total; };
PendW pend_args(const pre3_ctx *c);           // what is pending on this context (rows == 0: nothing)              (pre3_update.hip)
int pend_flush(pre3_ctx *c);                  // k_downdate_b3 on the pending rows, if any: P is P again           (pre3_update.hip)
struct InnovRide {                       // mode-0 innovation riding in another launch; n_blocks == 0: none
    int n_blocks, N, ld, n_clear;
    const int32_t *lm_type, *lm_off, *has_h;
    const void *P; const double *Hc, *Hl;
    double *S; int32_t *has_S, *clear;
    const float *pend_W = nullptr; int pend_ldw = 0, pend_rows = 0;        // PendW (fp32 contexts)
};
template <typename T>
__device__ __forceinline__ void innov_ride_block(const InnovRide &ir, int blk)
{
    const int gt = blk * blockDim.x + threadIdx.x;
    for (int t = gt; t < ir.n_clear; t += ir.n_blocks * blockDim.x) ir.clear[t] = 0;
    innovation_body<T>(ir.N, ir.lm_type, ir.lm_off, static_cast<const T *>(ir.P), ir.ld, ir.Hc, ir.Hl, ir.has_h, 0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr,
                       ir.S, ir.has_S, gt, ir.pend_W, ir.pend_ldw, ir.pend_rows);
}

ProjRide make_proj_ride(pre3_ctx *c, int which, int clear_first, int slot, int n_producers);   // pre3_geom.hip


// ---- (best, second, first arg) of siftmatch.c:110-116's scan, and the IC search's small-problem matcher tile (pre3_match.hip, k_ic_match_small; it
// also rides in k_project_innovation's launch) --------------------------------------------------------------------------------------------
template <typename ACC> struct Best3 { ACC best, second; int k; };

template <typename ACC> __device__ inline ACC acc_max();
template <> __device__ inline double acc_max<double>() { return INFINITY; }
template <> __device__ inline float acc_max<float>() { return INFINITY; }
template <> __device__ inline int acc_max<int>() { return 0x7fffffff; }

// merge two scan states (order independent; ties -> lowest index)
template <typename ACC>
__device__ inline void merge3(ACC &best, ACC &second, int &k, ACC ob, ACC os, int ok)
{
    if (ok < 0) return;
    if (k < 0) { best = ob; second = os; k = ok; return; }
    if (ob < best || (ob == best && ok < k)) {
        ACC ns = os < best ? os : best;
        best = ob; k = ok; second = ns;
    } else {
        ACC ns = ob < second ? ob : second;
        second = ns;
    }
}

template <typename ACC>
__device__ inline void push3(ACC &best, ACC &second, int &k, ACC v, int idx)
{
    // siftmatch.c:110-116 for increasing idx
    if (v < best) { second = best; best = v; k = idx; }
    else if (v < second) { second = v; }
}


constexpr int ICS_T = 32, ICS_LD = DESC_DIM + 1;          // tile edge; LDS row stride in doubles (odd: the 16 column pairs of a wave hit 16 different banks)
struct IcMatchRide {
    int n_blocks, ntn, N, K2;                     // n_blocks = ntn * ceil(N / 32) tiles, column tile fastest
    const double *bank, *scan;
    const int32_t *has_h;                         // not null: tiles without a predicted landmark leave at once (a launch of its own, behind the projection)
    double *pb, *ps; int32_t *pa;                 // partials [column tile][landmark] (a workgroup's 32 results are whole lines; [landmark][tile] rows measured slower: 27 vs 19 us for the gate)
};
constexpr int ICS_MAXT = 64;
// siftmatch.c:97-116 exactly (every pair's bins in order, no contraction) for 32 landmarks x 32 keypoints; 256 threads, 2 x 2 pairs per thread
__device__ __forceinline__ void ic_match_tile(const IcMatchRide &r, const int tile, double (*Qs)[ICS_LD], double (*Bs)[ICS_LD])
{
#pragma clang fp contract(off)
    const int tid = threadIdx.x, q0 = (tile / r.ntn) * ICS_T, k0 = (tile % r.ntn) * ICS_T, N = r.N, K2 = r.K2;
    if (r.has_h != nullptr) {
        const int any = tid < ICS_T && q0 + tid < N ? r.has_h[q0 + tid] != 0 : 0;
        if (!__syncthreads_or(any)) return;                // nothing predicted among these 32 landmarks: their partials are never read
    }
    {   // the tile's 32 + 32 descriptors, whole: 8 lanes per descriptor, 16 bytes per lane and load, every load in flight before the first LDS store
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const int lq = tid >> 3, l8 = tid & 7;
        const bool qok = q0 + lq < N, bok = k0 + lq < K2;
        const d2_t *qs = reinterpret_cast<const d2_t *>(r.bank + (size_t)(qok ? q0 + lq : 0) * DESC_DIM), *bs = reinterpret_cast<const d2_t *>(r.scan + (size_t)(bok ? k0 + lq : 0) * DESC_DIM);
        d2_t qv[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { qv[e] = qs[l8 + 8 * e]; bv[e] = bs[l8 + 8 * e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int bin = 2 * (l8 + 8 * e);
            Qs[lq][bin] = qok ? qv[e][0] : 0.0; Qs[lq][bin + 1] = qok ? qv[e][1] : 0.0;
            Bs[lq][bin] = bok ? bv[e][0] : 0.0; Bs[lq][bin + 1] = bok ? bv[e][1] : 0.0;
        }
    }
    __syncthreads();
    const int tq = tid >> 4, tk = tid & 15;                // queries 2 tq, 2 tq + 1 against columns 2 tk, 2 tk + 1
    double acc[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
#pragma unroll 8
    for (int bin = 0; bin < DESC_DIM; ++bin) {
        const double q[2] = { Qs[2 * tq][bin], Qs[2 * tq + 1][bin] }, b[2] = { Bs[2 * tk][bin], Bs[2 * tk + 1][bin] };
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double delta = q[a] - b[j];
                const double sq = delta * delta;
                acc[a][j] = acc[a][j] + sq;
            }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        double best = acc_max<double>(), second = acc_max<double>();
        int bk = -1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k2 = k0 + 2 * tk + j;
            if (k2 < K2) push3(best, second, bk, acc[a][j], k2);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 16), os = __shfl_xor(second, o, 16);
            const int ok = __shfl_xor(bk, o, 16);
            merge3(best, second, bk, ob, os, ok);
        }
        const int q = q0 + 2 * tq + a;
        if (tk == 0 && q < N) {
            const size_t o = (size_t)(tile % r.ntn) * N + q;           // [column tile][landmark]
            r.pb[o] = best; r.ps[o] = second; r.pa[o] = bk;
        }
    }
}

}  // namespace pre3
