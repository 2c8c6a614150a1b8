// pre3_geom.hip -- per-landmark / per-hypothesis kernels of the 1-point-RANSAC EKF step (gfx950).
//
// These stages are gather- and latency-bound (a few hundred KB per launch), not GEMMs: one lane per
// landmark (K2/K3), one workgroup per hypothesis with a wavefront-shuffle min/count reduction (K4/K5),
// one wavefront for the sequential termination replay (K6).  All camera geometry is fp64 regardless of
// the covariance dtype; only reads of P / H*P are typed.
//
// Reference statements reproduced (paths under matlab_code/):
//   predict        predict_state_and_covariance.m:59-143, aux_code/odometry_model.m:44-68
//   project        predict_camera_measurements.m:27-68, hi_inverse_depth.m:33-85, hi_cartesian.m:33-81
//   jacobian       calculate_Hi_inverse_depth_my_version.m:46-183, calculate_Hi_cartesian_my_version.m
//   innovation     search_IC_matches.m:33-44; rescue gate @ekf_filter/rescue_hi_inliers.m:35-46
//   window gate    matching_sift_based.m:119-134
//   ransac         ransac_hypotheses.m:40-80, compute_hypothesis_support_fast.m:33-110
#include "pre3_internal.h"
#include "pre3_geomdev.h"
#include "pre3_chain.h"

namespace pre3 {

// process noise Pn (7x7), predict_state_and_covariance.m:98-102 (constant)
__device__ inline void d_process_noise(double *Pn)
{
    for (int i = 0; i < 49; ++i) Pn[i] = 0;
    double sx = 0.01 / 3; sx = sx * sx;
    Pn[0] = Pn[8] = Pn[16] = sx;
    const double PI = 3.141592653589793238462643383279502884;
    double a = 0.24 / 2 * PI / 180;
    double e[3] = { a * 1, a * 0.1, a * 1 };
    double sr = sin(e[0] / 2), sp = sin(e[1] / 2), sy = sin(e[2] / 2);
    double cr = cos(e[0] / 2), cp = cos(e[1] / 2), cy = cos(e[2] / 2);
    double Qe[12] = {   // e2q.m:25-30
        0.5 * (-cy * cp * sr + sy * sp * cr), 0.5 * (-cy * sp * cr + sy * cp * sr), 0.5 * (-sy * cp * cr + cy * sp * sr),
        0.5 * ( cy * cp * cr + sy * sp * sr), 0.5 * (-cy * sp * sr - sy * cp * cr), 0.5 * (-sy * cp * sr - cy * sp * cr),
        0.5 * (-cy * sp * sr + sy * cp * cr), 0.5 * ( cy * cp * cr - sy * sp * sr), 0.5 * (-sy * sp * cr + cy * cp * sr),
        0.5 * (-sy * cp * sr - cy * sp * cr), 0.5 * (-cy * cp * sr - sy * sp * cr), 0.5 * ( cy * cp * cr + sy * sp * sr) };
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += (Qe[i * 3 + t] * (e[t] * e[t])) * Qe[j * 3 + t];
            Pn[(3 + i) * 7 + 3 + j] = s;
        }
}

// pred_params layout: [0..15] A4 = Qq1, [16..31] Jn, [32..80] Q7 = G Pn G' (7x7), [96..111] Jn of the last update (a copy the prediction's
// launch does not overwrite: k_predict reads it when it carries that update's update.m:42-46 pass, fuse_jn)
// One row of the 4x4 normalisation Jacobian applied to a 4-vector, with the contraction spelled out: k_jnorm_P and the prediction launch
// that carries the same pass (fuse_jn) must round identically (tests/test_gpu_synth.py: deferred == immediate HI update, bit for bit).
__device__ __forceinline__ double jn_row(const double *J, int i, const double v[4])
{
    return fma(J[i * 4 + 3], v[3], fma(J[i * 4 + 2], v[2], fma(J[i * 4 + 1], v[1], J[i * 4] * v[0])));
}
// k_predict_x and k_predict_P in ONE launch (a kernel boundary costs ~5 us on this platform, more than either kernel):
// lane 0 of every block recomputes the quaternion product and its normalisation Jacobian (a few dozen flops) instead of
// reading them from a previous kernel; block 0 additionally owns x_out[0:13], the process noise and the 7x7 pose block.
template <typename T>
__global__ __launch_bounds__(256) void k_predict(const double *__restrict__ x_in, double *x_out, T *__restrict__ P, int n, int ld, U7 u,
                                                 double *__restrict__ params, int n_pred_blocks, ProjRide pr, InboxRide ib, int fuse_jn, PendW pw = PendW{})
{
    if ((int)blockIdx.x >= n_pred_blocks + pr.n_blocks) { inbox_pull_block(ib); return; }               // the step's inbox crosses PCIe beside the prediction
    if ((int)blockIdx.x >= n_pred_blocks) { proj_ride_block(pr, blockIdx.x - n_pred_blocks); return; }   // IC-search projection rides along
    __shared__ double sQq1[16], sJn[16], sQ[49], sG[49], sPn[49], sGP[49], sJu[16];
    __shared__ double corner[49];      // old P[0:7,0:7]
    // fuse_jn: the previous update's rows/cols 3..6 <- Jn pass (update.m:42-46, k_jnorm_P) is applied here first, value for value as
    // that kernel would have stored it (rounded to T), so the launch in front of this one is saved (pre3_step with PRE3_OPT_DEFER_HI)
    if (fuse_jn && threadIdx.x >= 64 && threadIdx.x < 80) sJu[threadIdx.x - 64] = params[96 + threadIdx.x - 64];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    // landmarks copied (predict_state_and_covariance.m:79)
    for (int i = 13 + j; i < n; i += n_pred_blocks * blockDim.x) x_out[i] = x_in[i];
    double v[4] = { 0, 0, 0, 0 };
    if (j >= 7 && j < n) for (int t = 0; t < 4; ++t) v[t] = (double)P[(3 + t) * ld + j];
    if (blockIdx.x == 0 && threadIdx.x < 49) corner[threadIdx.x] = (double)P[(threadIdx.x / 7) * ld + (threadIdx.x % 7)];
    if (threadIdx.x == 0) {
        const double *q = x_in + 3;
        const double a = q[0], b = q[1], c = q[2], d = q[3];
        const double w = u.v[3], x = u.v[4], y = u.v[5], z = u.v[6];
        // the predicted pose (predict_pose, pre3_geomdev.h: the projection riders of this launch compute the same thing for themselves)
        double xo[7], pose[7];
        predict_pose(x_in, u, xo, pose);
        double R[9];
        if (blockIdx.x == 0) {
            d_q2R_sola(q, R);
            for (int i = 0; i < 7; ++i) x_out[i] = pose[i];
            for (int i = 7; i < 13; ++i) x_out[i] = 0;
            // (riders that wait for the pose -- PRE3_RIDE_POSE=0 -- get their signal right behind it: the rest of this lane's serial work, a few
            //  hundred dependent fp64 operations, need not stand in front of the 500 projections)
            if (pr.n_blocks && !pr.own_pose) { __threadfence(); __hip_atomic_fetch_add(pr.ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        }
        const double Qq1[16] = { w, -x, -y, -z,  x, w, z, -y,  y, -z, w, x,  z, y, -x, w };
        double Jn[16];
        d_normjac(xo + 3, Jn);     // at the un-normalised q (predict_state_and_covariance.m:137)
        for (int i = 0; i < 16; ++i) { sQq1[i] = Qq1[i]; sJn[i] = Jn[i]; }
        if (blockIdx.x == 0) {
            const double Qq2[16] = { a, -b, -c, -d,  b, a, -d, c,  c, d, a, -b,  d, -c, b, a };
            // G = [R 0; 0 Qq2] (7x7 non-zero part) and the process noise go to LDS; Q7 = G Pn G' is formed by 49 lanes below
            double Pn[49];
            for (int i = 0; i < 49; ++i) sG[i] = 0;
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) sG[i * 7 + k] = R[i * 3 + k];
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) sG[(3 + i) * 7 + 3 + k] = Qq2[i * 4 + k];
            d_process_noise(Pn);
            for (int i = 0; i < 49; ++i) sPn[i] = Pn[i];
            for (int i = 0; i < 16; ++i) { params[i] = Qq1[i]; params[16 + i] = Jn[i]; }
        }
    }
    // the projection riders read the landmarks from x_in (the prediction copies them unchanged) and only the POSE from x_out: block 0's
    // signal is the only one they wait for (every block signalling cost each of them a device-scope release)
    __syncthreads();                                               // (block 0's lane 0 has signalled the riders already, behind the pose)
    double cfix = 0; int cpos = -1;
    if (fuse_jn) {
        if (j >= 7 && j < n) { double w[4]; for (int i = 0; i < 4; ++i) w[i] = (double)(T)jn_row(sJu, i, v); for (int i = 0; i < 4; ++i) v[i] = w[i]; }
        if (blockIdx.x == 0 && threadIdx.x < 28) {
            // the 7x7 corner: columns 0..2 of rows 3..6 (and their mirror), then the 4x4 block J C J'
            const int t = threadIdx.x;
            if (t < 12) {
                const int jc = t >> 2, i = t & 3;
                const double w[4] = { corner[3 * 7 + jc], corner[4 * 7 + jc], corner[5 * 7 + jc], corner[6 * 7 + jc] };
                cfix = (double)(T)jn_row(sJu, i, w); cpos = (3 + i) * 7 + jc;
            } else {
                const int i = (t - 12) >> 2, k = (t - 12) & 3;
                double T1[4];
                for (int c2 = 0; c2 < 4; ++c2) { const double w[4] = { corner[3 * 7 + 3 + c2], corner[4 * 7 + 3 + c2], corner[5 * 7 + 3 + c2], corner[6 * 7 + 3 + c2] }; T1[c2] = jn_row(sJu, i, w); }
                cfix = (double)(T)jn_row(sJu, k, T1); cpos = (3 + i) * 7 + 3 + k;
            }
        }
        __syncthreads();
        if (cpos >= 0) { corner[cpos] = cfix; if (threadIdx.x < 12) corner[(cpos % 7) * 7 + cpos / 7] = cfix; }
        // (block 0 reads the corner behind the barriers of its pose-block section below)
    }
    if (j >= 7 && j < n) {
        double a[4], b[4];
        for (int i = 0; i < 4; ++i) a[i] = sQq1[i * 4] * v[0] + sQq1[i * 4 + 1] * v[1] + sQq1[i * 4 + 2] * v[2] + sQq1[i * 4 + 3] * v[3];
        for (int i = 0; i < 4; ++i) b[i] = sJn[i * 4] * a[0] + sJn[i * 4 + 1] * a[1] + sJn[i * 4 + 2] * a[2] + sJn[i * 4 + 3] * a[3];
        for (int i = 0; i < 4; ++i) { P[(3 + i) * ld + j] = (T)b[i]; P[j * ld + 3 + i] = (T)b[i]; }
    }
    if (blockIdx.x == 0) {
        // pose block: C = F7 * P7 * F7' + Q7 with F7 = blkdiag(I3, Qq1); then J7 C J7', J7 = blkdiag(I3, Jn).
        __shared__ double F7[49], J7[49], T1[49], C7[49];
        const int t = threadIdx.x, i = t / 7, k = t % 7;
        // Q7 = G Pn G' (same summation order as the serial form: t = 0..6)
        if (t < 49) { double s = 0; for (int q2 = 0; q2 < 7; ++q2) s += sG[i * 7 + q2] * sPn[q2 * 7 + k]; sGP[t] = s; }
        __syncthreads();
        if (t < 49) { double s = 0; for (int q2 = 0; q2 < 7; ++q2) s += sGP[i * 7 + q2] * sG[k * 7 + q2]; sQ[t] = s; params[32 + t] = s; }
        __syncthreads();
        if (t < 49) {
            double f = (i == k && i < 3) ? 1.0 : 0.0, jn = f;
            if (i >= 3 && k >= 3) { f = sQq1[(i - 3) * 4 + (k - 3)]; jn = sJn[(i - 3) * 4 + (k - 3)]; }
            F7[t] = f; J7[t] = jn;
        }
        __syncthreads();
        if (t < 49) { double s = 0; for (int q2 = 0; q2 < 7; ++q2) s += F7[i * 7 + q2] * corner[q2 * 7 + k]; T1[t] = s; }
        __syncthreads();
        if (t < 49) { double s = 0; for (int q2 = 0; q2 < 7; ++q2) s += T1[i * 7 + q2] * F7[k * 7 + q2]; C7[t] = s + sQ[t]; }
        __syncthreads();
        if (t < 49) { double s = 0; for (int q2 = 0; q2 < 7; ++q2) s += J7[i * 7 + q2] * C7[q2 * 7 + k]; T1[t] = s; }
        __syncthreads();
        if (t < 49) { double s = 0; for (int q2 = 0; q2 < 7; ++q2) s += T1[i * 7 + q2] * J7[k * 7 + q2]; P[i * ld + k] = (T)s; }
    }
    if constexpr (sizeof(T) == 4) {
        // PRE3_OPT_PEND_HI: P stands for P - W~'W~, so W~ takes the same congruence -- columns 3..6 of every row, through the same chain (the pending
        // update.m:42-46 pass first, rounded as it is for P, then Qq1 and Jn) --, and the planes of the one column block that holds them are split
        // again.  The last prediction block does it (it has the fewest columns of P).
        if (pw.rows > 0 && (int)blockIdx.x == n_pred_blocks - 1) {
            for (int k = threadIdx.x; k < pw.rows; k += blockDim.x) {
                float *wr = pw.W + (size_t)k * pw.ldw + 3;
                double wv[4] = { (double)wr[0], (double)wr[1], (double)wr[2], (double)wr[3] };
                if (fuse_jn) { double w[4]; for (int i = 0; i < 4; ++i) w[i] = (double)(T)jn_row(sJu, i, wv); for (int i = 0; i < 4; ++i) wv[i] = w[i]; }
                double a[4], b[4];
                for (int i = 0; i < 4; ++i) a[i] = sQq1[i * 4] * wv[0] + sQq1[i * 4 + 1] * wv[1] + sQq1[i * 4 + 2] * wv[2] + sQq1[i * 4 + 3] * wv[3];
                for (int i = 0; i < 4; ++i) b[i] = sJn[i * 4] * a[0] + sJn[i * 4 + 1] * a[1] + sJn[i * 4 + 2] * a[2] + sJn[i * 4 + 3] * a[3];
                for (int i = 0; i < 4; ++i) wr[i] = (float)b[i];
            }
            __syncthreads();
            const int nst = (pw.rows + B3_BK - 1) / B3_BK;
            for (int st = 0; st < nst; ++st) b3_split_block(pw.W, pw.ldw, static_cast<bf16x8_t *>(pw.Wp), pw.nst_total, 0, st, threadIdx.x);
        }
    }
}

// The rescue stage's chi2 gate (rescue_hi_inliers.m:35-46) in the SAME launch as the rows / columns 3..6 pass it depends on (round 5): the gate's
// blocks never read what that pass writes.  The persistent launch's consumers leave the un-normalised rows 3..6 of P in a side buffer Q (4 x ld),
// and an entry of the normalised P in those rows / columns is recomputed here from Q and Jn exactly as the pass computes and rounds it
// (jn_row; the 4 x 4 corner through its double-precision intermediate) -- every other entry of the gathered 13 x 13 block is read from P, where
// the pass leaves it alone.  Sixteen landmarks per block: the first sixteen lanes project them at x_k_k (as k_project_innovation), then sixteen
// lanes per landmark gather H P H' as innovation_body does, term for term.
template <typename T>
struct GateRide {
    int n_blocks, N; const T *Q; double chi2; int project;        // project == 0: h / H at x_k_k are there already (the persistent launch's strips)
    const int32_t *lm_type, *lm_off; const double *x; CamD cam;
    double *h; int32_t *has_h; double *Hc, *Hl; const double *z; const int32_t *ic, *li; int32_t *hi;
};
template <typename T>
__device__ __forceinline__ void gate_ride_block(const GateRide<T> &g, const T *__restrict__ P, int ld, const double *__restrict__ params, int blk)
{
    __shared__ double gJn[16];
    if (threadIdx.x < 16) gJn[threadIdx.x] = params[16 + threadIdx.x];
    if (g.project && threadIdx.x < 16 && blk * 16 + (int)threadIdx.x < g.N) project_one(blk * 16 + threadIdx.x, g.lm_type, g.lm_off, g.x, g.cam, 0, g.h, g.has_h, g.Hc, g.Hl);
    __threadfence_block();
    __syncthreads();
    const int gt = blk * 256 + threadIdx.x, i = gt >> 4, b = gt & 15;
    const bool valid = i < g.N;
    const int ii = valid ? i : 0;
    const bool active = valid && g.ic[ii] == 1 && g.li[ii] == 0;
    const int d = g.lm_type[ii] == PRE3_INVDEPTH ? 6 : 3, off = g.lm_off[ii], nn = 7 + d;
    const double *Hc = g.Hc, *Hl = g.Hl;
    double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    if (active && b < nn) {
        const int ib = b < 7 ? b : off + b - 7;
        // the un-normalised rows 3..6 of P at column c
        auto qcol = [&](int c, double (&v)[4]) { for (int t = 0; t < 4; ++t) v[t] = (double)g.Q[(size_t)t * ld + c]; };
        double p7[7], pl[6];
        if (ib < 3 || ib >= 7) {
            // a row the pass leaves alone, except at the columns 3..6: P(ib, 3+k) = P(3+k, ib) = (T) Jn(k, :) . Pold(3..6, ib)
            const T *prow = P + (size_t)ib * ld;
            double v[4]; qcol(ib, v);
            for (int a = 0; a < 3; ++a) p7[a] = (double)prow[a];
            for (int k = 0; k < 4; ++k) p7[3 + k] = (double)(T)jn_row(gJn, k, v);
            for (int a = 0; a < 6; ++a) pl[a] = a < d ? (double)prow[off + a] : 0.0;
        } else {
            // row 3 + r of the normalised P: (T) Jn(r, :) . Pold(3..6, c) off the corner; the corner as k_jnorm_P's thread 0 computes it
            const int r = ib - 3;
            for (int a = 0; a < 3; ++a) { double v[4]; qcol(a, v); p7[a] = (double)(T)jn_row(gJn, r, v); }
            double T1[4];
            for (int k = 0; k < 4; ++k) { double w[4]; qcol(3 + k, w); T1[k] = jn_row(gJn, r, w); }
            for (int k = 0; k < 4; ++k) p7[3 + k] = (double)(T)jn_row(gJn, k, T1);
            for (int a = 0; a < 6; ++a) { pl[a] = 0.0; if (a < d) { double v[4]; qcol(off + a, v); pl[a] = (double)(T)jn_row(gJn, r, v); } }
        }
        double hp0 = 0, hp1 = 0;
#pragma unroll
        for (int a = 0; a < 7; ++a) { hp0 += Hc[14 * ii + a] * p7[a]; hp1 += Hc[14 * ii + 7 + a] * p7[a]; }
#pragma unroll
        for (int a = 0; a < 6; ++a)
            if (a < d) { hp0 += Hl[12 * ii + a] * pl[a]; hp1 += Hl[12 * ii + 6 + a] * pl[a]; }
        const double h0b = b < 7 ? Hc[14 * ii + b] : Hl[12 * ii + b - 7];
        const double h1b = b < 7 ? Hc[14 * ii + 7 + b] : Hl[12 * ii + 6 + b - 7];
        s00 = hp0 * h0b; s01 = hp0 * h1b; s10 = hp1 * h0b; s11 = hp1 * h1b;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s00 += __shfl_xor(s00, o, 16); s01 += __shfl_xor(s01, o, 16);
        s10 += __shfl_xor(s10, o, 16); s11 += __shfl_xor(s11, o, 16);
    }
    if (!valid || b != 0 || !active) return;
    const double det = s00 * s11 - s01 * s10;
    const double i00 = s11 / det, i01 = -s01 / det, i10 = -s10 / det, i11 = s00 / det;
    const double n0 = g.z[2 * i] - g.h[2 * i], n1 = g.z[2 * i + 1] - g.h[2 * i + 1];
    const double t0 = n0 * i00 + n1 * i10, t1 = n0 * i01 + n1 * i11;
    const double d2 = t0 * n0 + t1 * n1;
    g.hi[i] = d2 < g.chi2 ? 1 : 0;
}

// rows/cols 3..6 <- Jn (update.m:42-46).  params[16..31] = Jn.
template <typename T>
__global__ void k_jnorm_P(T *__restrict__ P, int n, int ld, const double *__restrict__ params, int n_jn_blocks, ProjRide pr, GateRide<T> gr)
{
    // the rescue's projection at x_k_k (rescue_hi_inliers.m:31-32) rides along when the update's x came out of the launch in front
    // (persistent factorisation with its own x-update): it needs nothing of this launch -- and, when the launch in front also left the
    // un-normalised rows 3..6 behind (GateRide), the whole gate does
    if ((int)blockIdx.x >= n_jn_blocks) {
        if (gr.n_blocks > 0) gate_ride_block<T>(gr, P, ld, params, blockIdx.x - n_jn_blocks);
        else proj_ride_block(pr, blockIdx.x - n_jn_blocks);
        return;
    }
    __shared__ double sJn[16];
    __shared__ double corner[16];
    if (threadIdx.x < 16) sJn[threadIdx.x] = params[16 + threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x < 16) corner[threadIdx.x] = (double)P[(3 + threadIdx.x / 4) * ld + 3 + (threadIdx.x % 4)];
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    bool strip = j < n && (j < 3 || j >= 7);
    double v[4] = { 0, 0, 0, 0 };
    if (strip) for (int t = 0; t < 4; ++t) v[t] = (double)P[(3 + t) * ld + j];
    __syncthreads();
    if (strip) {
        for (int i = 0; i < 4; ++i) {
            const double b = jn_row(sJn, i, v);
            P[(3 + i) * ld + j] = (T)b; P[j * ld + 3 + i] = (T)b;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double T1[16];
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) { const double w[4] = { corner[k], corner[4 + k], corner[8 + k], corner[12 + k] }; T1[i * 4 + k] = jn_row(sJn, i, w); }
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) P[(3 + i) * ld + 3 + k] = (T)jn_row(sJn, k, &T1[i * 4]);
    }
}

// ------------------------------------------------------------------------------------------------
// K2 project + Jacobian: one lane per landmark (project_one: pre3_geomdev.h)
// ------------------------------------------------------------------------------------------------
__global__ void k_project(int N, const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                          const double *__restrict__ x, CamD cam, int clear_first,
                          double *__restrict__ h, int32_t *__restrict__ has_h, double *__restrict__ Hc, double *__restrict__ Hl)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    project_one(i, lm_type, lm_off, x, cam, clear_first, h, has_h, Hc, Hl);
}

// ------------------------------------------------------------------------------------------------
// K3 innovation covariance / rescue gate: one lane per landmark gathers the 13x13 (10x10) block of P
// that H_i's non-zeros select.  mode 0: S_i = H P H' + I for predicted landmarks.
// mode 1 (rescue_hi_inliers.m:35-46): for ic && !li: d2 = nu' inv(H P H') nu < chi2 -> hi flag.
// ------------------------------------------------------------------------------------------------

// the HI collection (k_collect_hi's work) as the tail of the rescue launch: run by the first wave of the LAST workgroup
struct HiArgs { int fuse, m, seq; const int32_t *meas; int32_t *hi_meas, *sel_rows, *stats, *mail; unsigned int *done; };
__device__ __forceinline__ void collect_hi_body(int m, const int32_t *__restrict__ meas, const int32_t *__restrict__ lm_ic,
                                                const int32_t *__restrict__ lm_li, const int32_t *lm_hi,
                                                int32_t *__restrict__ hi_meas, int32_t *__restrict__ sel_rows, int32_t *__restrict__ stats,
                                                int32_t *mail, int seq);

__device__ __forceinline__ void hi_tail(const HiArgs &ha, const int32_t *__restrict__ ic, const int32_t *__restrict__ li, int32_t *hi)
{
    if (!ha.fuse) return;
    __shared__ int s_last;
    __syncthreads();                                   // this workgroup's hi flags are written
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(ha.done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) *ha.done = 0;
        collect_hi_body(ha.m, ha.meas, ic, li, hi, ha.hi_meas, ha.sel_rows, ha.stats, ha.mail, ha.seq);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_innovation(int N, const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                                                    const T *__restrict__ P, int ld, const double *__restrict__ Hc, const double *__restrict__ Hl,
                                                    const int32_t *__restrict__ has_h, int mode, double chi2,
                                                    const double *__restrict__ h, const double *__restrict__ z,
                                                    const int32_t *__restrict__ ic, const int32_t *__restrict__ li, int32_t *__restrict__ hi,
                                                    double *__restrict__ S, int32_t *__restrict__ has_S, int32_t *__restrict__ clear, int n_clear,
                                                    HiArgs ha)
{
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_clear; t += gridDim.x * blockDim.x) clear[t] = 0;
    innovation_body<T>(N, lm_type, lm_off, P, ld, Hc, Hl, has_h, mode, chi2, h, z, ic, li, hi, S, has_S, blockIdx.x * blockDim.x + threadIdx.x);
    hi_tail(ha, ic, li, hi);
}

// k_project and k_innovation in ONE launch: the first of a landmark's 16 lanes projects it and writes h / H, the block
// barrier publishes them, then the 16 lanes gather H P H' as before.  Saves a kernel boundary (~5 us) twice per step.
template <typename T>
__global__ __launch_bounds__(256) void k_project_innovation(int N, const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                                                            const double *__restrict__ x, CamD cam, int clear_first, const T *__restrict__ P, int ld,
                                                            double *Hc, double *Hl, int32_t *has_h, int mode, double chi2, double *h,
                                                            const double *__restrict__ z, const int32_t *__restrict__ ic,
                                                            const int32_t *__restrict__ li, int32_t *__restrict__ hi, double *__restrict__ S,
                                                            int32_t *__restrict__ has_S, int32_t *__restrict__ clear, int n_clear, HiArgs ha,
                                                            int32_t *__restrict__ clear2, int n_clear2, IcMatchRide mr)
{
    // pre3_ic_search (round 5): the small-problem matcher's tiles are extra workgroups of this launch -- they read the descriptor bank and the scan,
    // which nothing here touches
    const int n_main = gridDim.x - mr.n_blocks;
    if (mr.n_blocks > 0) {
        __shared__ double Qs[ICS_T][ICS_LD], Bs[ICS_T][ICS_LD];
        if ((int)blockIdx.x >= n_main) { ic_match_tile(mr, blockIdx.x - n_main, Qs, Bs); return; }
    }
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    // a step's IC search precedes its measurements: clear the inlier flags of the previous frame here (n_clear = 0 otherwise)
    for (int t = gt; t < n_clear; t += n_main * blockDim.x) clear[t] = 0;
    // (pre3_ic_search: individually_compatible of every landmark -- a hipMemsetAsync of 4 N bytes went out as three fill kernels, 15 us)
    for (int t = gt; t < n_clear2; t += n_main * blockDim.x) clear2[t] = 0;
    // the block's 16 landmarks are projected by the first 16 lanes of its first wave (one wave runs the long fp64 code, not four)
    if (threadIdx.x < 16 && (int)(blockIdx.x * 16 + threadIdx.x) < N) project_one(blockIdx.x * 16 + threadIdx.x, lm_type, lm_off, x, cam, clear_first, h, has_h, Hc, Hl);
    __threadfence_block();
    __syncthreads();
    innovation_body<T>(N, lm_type, lm_off, P, ld, Hc, Hl, has_h, mode, chi2, h, z, ic, li, hi, S, has_S, blockIdx.x * blockDim.x + threadIdx.x);
    hi_tail(ha, ic, li, hi);
}

// matching_sift_based.m:119-134
__global__ void k_window_gate(int M, const int32_t *__restrict__ pred_idx, const int32_t *__restrict__ k1,
                              const double *__restrict__ zc, const double *__restrict__ h, const double *__restrict__ S,
                              const int32_t *__restrict__ has_S, int strict, double *__restrict__ z, int32_t *__restrict__ ic,
                              int32_t *__restrict__ accept)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    int lm = pred_idx[k1[c]];
    int slm = strict ? pred_idx[c] : lm;
    double half = has_S[slm] ? ceil(3 * sqrt(S[4 * slm])) : 40.0;
    double dx = zc[2 * c] - h[2 * lm], dy = zc[2 * c + 1] - h[2 * lm + 1];
    int ok = sqrt(dx * dx + dy * dy) <= half;
    if (ok) { ic[lm] = 1; z[2 * lm] = zc[2 * c]; z[2 * lm + 1] = zc[2 * c + 1]; }
    if (accept) accept[c] = ok;
}

// ------------------------------------------------------------------------------------------------
// measurement rows in ELL form: row 2s+c of the selected measurement sel[s] (index into meas[])
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_build_rows(int nsel, const int32_t *__restrict__ sel, const int32_t *__restrict__ meas,
                             const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                             const double *__restrict__ Hc, const double *__restrict__ Hl,
                             const double *__restrict__ z, const double *__restrict__ h,
                             int32_t *__restrict__ row_col, T *__restrict__ row_val, double *__restrict__ row_nu, int r_pad)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;     // row
    if (a >= r_pad) return;
    int32_t *cc = row_col + a * ELLW;
    T *vv = row_val + a * ELLW;
    if (a >= 2 * nsel) {
        for (int t = 0; t < ELLW; ++t) { cc[t] = 0; vv[t] = (T)0; }
        row_nu[a] = 0;
        return;
    }
    int s = a >> 1, c = a & 1;
    int i = meas[sel ? sel[s] : s];
    int d = lm_type[i] == PRE3_INVDEPTH ? 6 : 3;
    int off = lm_off[i];
    for (int t = 0; t < 7; ++t) { cc[t] = t; vv[t] = (T)Hc[14 * i + c * 7 + t]; }
    for (int t = 0; t < 6; ++t) { cc[7 + t] = t < d ? off + t : 0; vv[7 + t] = t < d ? (T)Hl[12 * i + c * 6 + t] : (T)0; }
    for (int t = 13; t < ELLW; ++t) { cc[t] = 0; vv[t] = (T)0; }
    row_nu[a] = z[2 * i + c] - h[2 * i + c];
}

// ------------------------------------------------------------------------------------------------
// K4+K5 RANSAC hypothesis state + support: one workgroup per hypothesis.
//   S_h = G[sel,sel] + I (2k x 2k), w = S_h^-1 nu_h, x_i = x + (H P)[sel,:]' w   (ransac_hypotheses.m:61-63)
//   support per compute_hypothesis_support_fast.m:33-110 on the rows of x_i the scorer reads.
// ------------------------------------------------------------------------------------------------
__device__ inline double wave_min(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_sum(int v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Arguments of the selection stage when it rides at the end of the scoring launch (fuse != 0): the LAST workgroup to
// finish (device-wide ticket) replays the reference's loop over all supports -- one kernel boundary less per step.
struct SelArgs {
    int fuse, n_draw, k, early_exit, seq;
    int32_t *li_meas, *lm_li, *sel_rows, *stats, *mail;
    unsigned int *done;
};
__device__ __forceinline__ void select_body(int n_draw, int k, int early_exit, int m, const int32_t *__restrict__ meas,
                                            int32_t *support, const uint32_t *masks, int mask_words,
                                            int32_t *__restrict__ li_meas, int32_t *__restrict__ lm_li,
                                            int32_t *__restrict__ sel_rows, int32_t *__restrict__ stats,
                                            int32_t *mail, int seq, int err_idx = -1);

template <typename T, int K>
__global__ __launch_bounds__(512) void k_ransac_score(int hyp_begin, const int32_t *__restrict__ hyp, int m,
                                                      const int32_t *__restrict__ meas, const int32_t *__restrict__ lm_type,
                                                      const int32_t *__restrict__ lm_off, const double *__restrict__ x,
                                                      const T *__restrict__ HP, int ldw, const T *__restrict__ G, int ldg,
                                                      const double *__restrict__ row_nu, const double *__restrict__ z,
                                                      CamD cam, double threshold, int32_t *support,
                                                      uint32_t *masks, int mask_words, SelArgs sel,
                                                      const int32_t *__restrict__ row_col, const T *__restrict__ row_val)
{
    constexpr int R = 2 * K;                // compile-time so that every small array stays in registers
    extern __shared__ double s_res[];       // [m] residuals, then mask words
    __shared__ double s_w[R];
    __shared__ double s_red[8];
    __shared__ int s_cnt[8];
    const int hidx = hyp_begin + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Everything that does not depend on the solution of the hypothesis' 2k x 2k system is requested BEFORE the barrier behind that
    // solve: the rows of H*P the hypothesis draws (pose columns: uniform; this lane's landmark columns), the landmark's state and pixel.
    // Three dependent load levels (meas -> type / off -> H*P) used to start only after the solve; now they fly beside it.
    int rows[R];
#pragma unroll
    for (int s = 0; s < K; ++s) { const int j = hyp[hidx * K + s]; rows[2 * s] = 2 * j; rows[2 * s + 1] = 2 * j + 1; }
    const T *hp[R];
#pragma unroll
    for (int b = 0; b < R; ++b) hp[b] = HP + (size_t)rows[b] * ldw;
    constexpr bool PRE_POSE = R * sizeof(T) <= 48;                // (fp64 with k = 4 would spill: its pose columns are read after the barrier)
    T hpc[R][7];                                                   // pose columns of the drawn rows (the same for every lane)
    double xp[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        xp[c] = x[c];
        if (PRE_POSE) {
#pragma unroll
            for (int b = 0; b < R; ++b) hpc[b][c] = hp[b][c];
        }
    }
    const bool have0 = PRE_POSE && tid < m;                        // this lane's first measurement (all of them when m <= 512)
    int i0 = 0, type0 = PRE3_INVDEPTH;
    T hpl[R][6];
    double xl[6], z0 = 0, z1 = 0;
    if (have0) {
        i0 = meas[tid];
        type0 = lm_type[i0];
        const int off = lm_off[i0];
        // six consecutive columns per row as ONE access each (element-aligned only: off = 13 + 6 i), instead of six strided dword loads:
        // a wave's 36 + 6 load instructions become 6 + 1 wide ones and the texture path sees a third of the requests.  A Cartesian
        // landmark's last three columns belong to its neighbour (or to the padding of the row: ldw >= n + 64) and are zeroed after the load.
        struct __attribute__((packed, aligned(sizeof(T)))) Row6 { T v[6]; };
        struct __attribute__((packed, aligned(8))) X6 { double v[6]; };
        const X6 xv = *reinterpret_cast<const X6 *>(x + off);
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const Row6 rv = *reinterpret_cast<const Row6 *>(hp[b] + off);
#pragma unroll
            for (int c = 0; c < 6; ++c) hpl[b][c] = (c < 3 || type0 == PRE3_INVDEPTH) ? rv.v[c] : (T)0;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) xl[c] = (c < 3 || type0 == PRE3_INVDEPTH) ? xv.v[c] : 0.0;
        z0 = z[2 * i0]; z1 = z[2 * i0 + 1];
    }
    const int solver = (int)(blockDim.x >> 6) - 1;                 // the last wave: it has the fewest (at m <= 448: no) measurements of its own
    if (wv == solver) {
        // lane (a, b) gathers one entry of the augmented system [S_h | nu_h]; lane 0 then eliminates
        double A[R][R + 1];
        T gv = (T)0;
        if (G == nullptr) {
            // H*P*H' among the hypothesis' own rows, computed here: lane p < R(R+1)/2 takes pair p = (i >= j) and runs k_ell_G's sum for the
            // entry (max row, min row) term for term (the values a full build of G would hold) -- no launch for G in front of the scoring
            int pi = 0;
            while ((pi + 1) * (pi + 2) / 2 <= lane) ++pi;
            const int pj = lane - pi * (pi + 1) / 2;
            if (pi < R) {
                int ri = 0, rj = 0;
#pragma unroll
                for (int a = 0; a < R; ++a) { if (a == pi) ri = rows[a]; if (a == pj) rj = rows[a]; }
                const int ra = ri > rj ? ri : rj, rb = ri > rj ? rj : ri;
                T vv[ELLW]; int cc[ELLW];
#pragma unroll
                for (int t = 0; t < ELLW; ++t) { vv[t] = row_val[rb * ELLW + t]; cc[t] = row_col[rb * ELLW + t]; }
                T hv[ELLW];
#pragma unroll
                for (int t = 0; t < ELLW; ++t) hv[t] = HP[(size_t)ra * ldw + cc[t]];
#pragma unroll
                for (int t = 0; t < ELLW; ++t) gv = ell_fma(vv[t], hv[t], gv);
            }
        }
#pragma unroll
        for (int a = 0; a < R; ++a) {
#pragma unroll
            for (int b = 0; b < R; ++b) {                      // G holds its lower triangle: entry (max, min)
                if (G != nullptr) {
                    const int ra = rows[a] > rows[b] ? rows[a] : rows[b], rb = rows[a] > rows[b] ? rows[b] : rows[a];
                    A[a][b] = (double)G[(size_t)ra * ldg + rb] + (a == b ? 1.0 : 0.0);
                } else {
                    const int hi_ = a > b ? a : b, lo_ = a > b ? b : a;
                    A[a][b] = (double)__shfl(gv, hi_ * (hi_ + 1) / 2 + lo_, 64) + (a == b ? 1.0 : 0.0);
                }
            }
            A[a][R] = row_nu[rows[a]];
        }
        if (lane == 0) {
            // Gaussian elimination with partial pivoting (MATLAB's inv(S)*nu up to rounding), fully unrolled
#pragma unroll
            for (int c = 0; c < R; ++c) {
                int p = c; double best = fabs(A[c][c]);
#pragma unroll
                for (int a = c + 1; a < R; ++a) { double v = fabs(A[a][c]); if (v > best) { best = v; p = a; } }
#pragma unroll
                for (int a = c + 1; a < R; ++a)
                    if (a == p) {
#pragma unroll
                        for (int b = 0; b <= R; ++b) { double t = A[c][b]; A[c][b] = A[a][b]; A[a][b] = t; }
                    }
                const double dinv = 1.0 / A[c][c];
#pragma unroll
                for (int a = c + 1; a < R; ++a) {
                    const double l = A[a][c] * dinv;
#pragma unroll
                    for (int b = c + 1; b <= R; ++b) A[a][b] -= l * A[c][b];
                }
            }
            double w[R];
#pragma unroll
            for (int a = R - 1; a >= 0; --a) {
                double s = A[a][R];
#pragma unroll
                for (int b = a + 1; b < R; ++b) s -= A[a][b] * w[b];
                w[a] = s / A[a][a];
            }
#pragma unroll
            for (int a = 0; a < R; ++a) s_w[a] = w[a];
        }
    }
    __syncthreads();
    double w[R];
#pragma unroll
    for (int b = 0; b < R; ++b) w[b] = s_w[b];
    // pose part of x_i (every lane redundantly)
    double xc[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        double s = 0;
#pragma unroll
        for (int b = 0; b < R; ++b) s += w[b] * (double)(PRE_POSE ? hpc[b][c] : hp[b][c]);
        xc[c] = xp[c] + s;
    }
    double rot[9];
    d_q2r(xc + 3, rot);                   // un-normalised quaternion (quirk Q4)
    double lmin = INFINITY;
    for (int j = tid; j < m; j += blockDim.x) {
        const bool pre = PRE_POSE && j == tid;                     // the first measurement of the lane came with the prefetch
        const int i = pre ? i0 : meas[j];
        const int type = pre ? type0 : lm_type[i], off = pre ? 0 : lm_off[i];
        double y[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double s = 0;
            if (c < 3 || type == PRE3_INVDEPTH) {
#pragma unroll
                for (int b = 0; b < R; ++b) s += w[b] * (double)(pre ? hpl[b][c] : hp[b][off + c]);
                s += pre ? xl[c] : x[off + c];
            }
            y[c] = s;
        }
        double v[3], hc[3];
        d_ray(type, y, xc, v);
#pragma unroll
        for (int c = 0; c < 3; ++c) hc[c] = rot[0 * 3 + c] * v[0] + rot[1 * 3 + c] * v[1] + rot[2 * 3 + c] * v[2];
        double uvd[2];
        d_pinhole_distort(hc, cam, uvd);
        const double n0 = (pre ? z0 : z[2 * i]) - uvd[0], n1 = (pre ? z1 : z[2 * i + 1]) - uvd[1];
        const double res = sqrt(n0 * n0 + n1 * n1);
        s_res[j] = res;
        if (type == PRE3_INVDEPTH) lmin = fmin(lmin, res);
    }
    lmin = wave_min(lmin);
    if (lane == 0) s_red[wv] = lmin;
    uint32_t *s_mask = (uint32_t *)(s_res + m);
    for (int wd = tid; wd < mask_words; wd += blockDim.x) s_mask[wd] = 0;
    __syncthreads();
    double minres = s_red[0];
    for (int w2 = 1; w2 < (int)(blockDim.x >> 6); ++w2) minres = fmin(minres, s_red[w2]);
    int cnt = 0;
    for (int j = tid; j < m; j += blockDim.x) {
        const int type = lm_type[meas[j]];
        const double res = s_res[j];
        // NaN residuals compare false, as in MATLAB
        const int in = (type == PRE3_INVDEPTH) ? (res < (minres + threshold)) : (res < threshold);
        if (in) { atomicOr(&s_mask[j >> 5], 1u << (j & 31)); ++cnt; }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) s_cnt[wv] = cnt;
    __syncthreads();
    if (tid == 0) { int sc = 0; for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) sc += s_cnt[w2]; support[hidx] = sc; }
    for (int wd = tid; wd < mask_words; wd += blockDim.x) masks[(size_t)hidx * mask_words + wd] = s_mask[wd];
    if (sel.fuse) {
        __shared__ int s_last;
        __syncthreads();                                   // this workgroup's support and mask are written
        if (tid == 0) {
            __threadfence();
            s_last = atomicAdd(sel.done, 1u) == gridDim.x - 1;
        }
        __syncthreads();
        if (s_last) {                                               // every thread enters (the stage's barriers are reached by all 512); 256 lanes work
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // see every other workgroup's results
            if (tid == 0) *sel.done = 0;
            select_body(sel.n_draw, sel.k, sel.early_exit, m, meas, support, masks, mask_words, sel.li_meas, sel.lm_li, sel.sel_rows,
                        sel.stats, sel.mail, sel.seq);
        }
    }
}

// K6: sequential replay of ransac_hypotheses.m:40-80 over the supports (quirk Q1), winner's mask ->
// low_innovation_inlier flags (set_as_most_supported_hypothesis.m:32-52) + compacted row list.
// stats: [0] best [1] iterations [2] n_hyp [3] max_support [4] n_li
__device__ inline int wave_compact(int flag, int lane, int base_cnt, int32_t *__restrict__ dst, int value)
{
    unsigned long long mask = __ballot(flag);
    int pos = __popcll(mask & ((1ull << lane) - 1ull));
    if (flag) dst[base_cnt + pos] = value;
    return base_cnt + __popcll(mask);
}

__device__ inline int ransac_n_hyp(int sup, int m)
{
    // ransac_hypotheses.m:77-78: epsilon = 1 - support/num_IC; n_hyp = ceil(log(1-p)/log(1-(1-epsilon)))
    double epsilon = 1 - ((double)sup / (double)m);
    return (int)ceil(log(1 - 0.99) / log(1 - (1 - epsilon)));
}

// The reference's loop (ransac_hypotheses.m:40-80) only changes state at "improvements" (support > running max), and its exit test
// n_hyp <= k can only become true at an improvement.  n_hyp falls as the support grows, so the FIRST index whose support passes the test is
// an improvement (every earlier support failed it, hence is smaller) and is where the loop stops; the winner is the first maximum up to
// there.  Both are reductions: every thread tests its own supports (the logarithms run side by side instead of one improvement after the
// other), a min over the indices that pass, then a max over (support, -index).  Every thread of the workgroup calls this (barriers inside);
// threads >= 256 only keep the barriers.
struct SelBest { int best, iters, n_hyp, max_support; };
__device__ __forceinline__ SelBest select_find_best(int n_draw, int k, int early_exit, int m, const int32_t *support)
{
    __shared__ int s_stop[4];
    __shared__ unsigned long long s_key[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool act = tid < 256;
    const int limit = early_exit ? (n_draw < 1000 ? n_draw : 1000) : n_draw;
    int v4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int it = tid + 256 * q; v4[q] = (act && it < limit) ? support[it] : -1; }
    int e = limit;                                              // the index the loop stops at (limit: it runs to the end)
    if (early_exit) {
        int mine = 0x7fffffff;
#pragma unroll
        for (int q = 3; q >= 0; --q) if (v4[q] >= 1 && ransac_n_hyp(v4[q], m) <= k) mine = tid + 256 * q;
        for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
        if (act && lane == 0) s_stop[wv] = mine;
        __syncthreads();
        const int first = min(min(s_stop[0], s_stop[1]), min(s_stop[2], s_stop[3]));
        if (first < limit) e = first;
    }
    unsigned long long key = 0;                                 // (support, ~index): the maximum is the first largest support
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int it = tid + 256 * q;
        if (it <= e && v4[q] >= 1) key = max(key, ((unsigned long long)v4[q] << 32) | (unsigned long long)(0xffffffffu - (unsigned)it));
    }
    for (int it = tid + 1024; act && it < limit; it += 256) {   // (only without the early exit: more than 1000 supports)
        const int v = support[it];
        if (v >= 1) key = max(key, ((unsigned long long)v << 32) | (unsigned long long)(0xffffffffu - (unsigned)it));
    }
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned long long)__shfl_xor((long long)key, o, 64));
    if (act && lane == 0) s_key[wv] = key;
    __syncthreads();
    key = max(max(s_key[0], s_key[1]), max(s_key[2], s_key[3]));
    SelBest r;
    r.best = key ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : -1;
    r.max_support = (int)(key >> 32);
    r.iters = e < limit ? e + 1 : limit;
    r.n_hyp = r.best >= 0 ? ransac_n_hyp(r.max_support, m) : 1000;
    return r;
}

// out[0..5]: written twice -- device stats (for later kernels) and the pinned host mirror (polled by the host).
__device__ __forceinline__ void select_body(int n_draw, int k, int early_exit, int m, const int32_t *__restrict__ meas,
                                            int32_t *support, const uint32_t *masks, int mask_words,
                                            int32_t *__restrict__ li_meas, int32_t *__restrict__ lm_li,
                                            int32_t *__restrict__ sel_rows, int32_t *__restrict__ stats,
                                            int32_t *mail, int seq, int err_idx)
{
    __shared__ int s_wcnt[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    const SelBest sb = select_find_best(n_draw, k, early_exit, m, support);
    if (tid == 0) { stats[0] = sb.best; stats[1] = sb.iters; stats[2] = sb.n_hyp; stats[3] = sb.max_support; }
    const int best = sb.best, iters = sb.iters;
    const bool act = tid < 256;                                            // the launch may have more waves (k_ransac_score: 8): they only keep the barriers
    // (k_select_gather: the gather's workgroups replay select_find_best on this array while this store runs.  That is a benign race by an invariant
    //  both sides keep: only entries BEHIND the exit index `iters` are overwritten, the replay's result depends on the entries up to it alone --
    //  its exit test passes at the first index whose support suffices, and a -1 never passes it.  Trimming at or below `iters` would break it.)
    for (int it = iters + tid; act && it < n_draw; it += 256) support[it] = -1;   // never evaluated by the reference
    // winner's mask -> flags (set_as_most_supported_hypothesis.m:32-52) + ordered compaction of the LI rows
    for (int base = 0; base < m; base += 256) {
        const int j = base + tid;
        int in = 0;
        if (act && j < m) {
            in = best >= 0 ? (masks[(size_t)best * mask_words + (j >> 5)] >> (j & 31)) & 1 : 0;
            li_meas[j] = in; lm_li[meas[j]] = in;
        }
        const unsigned long long bal = __ballot(in);
        if (act && lane == 0) s_wcnt[wv] = __popcll(bal);
        __syncthreads();
        int pre = s_base;
        for (int w2 = 0; w2 < wv && w2 < 4; ++w2) pre += s_wcnt[w2];
        if (in) sel_rows[pre + __popcll(bal & ((1ull << lane) - 1ull))] = j;
        // the measurements that are NOT low-innovation inliers, as landmarks, in measurement order behind the list's first m entries: the rescue
        // stage's candidates (rescue_hi_inliers.m:36) -- the persistent launch's strips deal them out by rank (pre3_cholp.hip, tail_gate_body)
        else if (act && j < m) sel_rows[m + j - (pre + __popcll(bal & ((1ull << lane) - 1ull)))] = (j << 16) | (meas[j] & 0xffff);      // (measurement | landmark: the tail is off beyond 65535 landmarks)
        __syncthreads();
        if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    if (tid == 0) {
        stats[4] = s_base;
        // host mailbox: payload, system-scope fence, then the sequence word the host polls
        mail[0] = sb.best; mail[1] = sb.iters; mail[2] = sb.n_hyp; mail[3] = sb.max_support; mail[4] = s_base;
        // a sharded round (pre3_ransac_sharded): the word behind the all-reduced supports and masks counts the ranks whose slice is missing
        mail[11] = err_idx >= 0 ? support[err_idx] : 0;
        __threadfence_system();
        __hip_atomic_store(&mail[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_ransac_select(int n_draw, int k, int early_exit, int m, const int32_t *__restrict__ meas,
                                                       int32_t *__restrict__ support, const uint32_t *__restrict__ masks, int mask_words,
                                                       int32_t *__restrict__ li_meas, int32_t *__restrict__ lm_li,
                                                       int32_t *__restrict__ sel_rows, int32_t *__restrict__ stats,
                                                       int32_t *mail, int seq, int err_idx)
{
    select_body(n_draw, k, early_exit, m, meas, support, masks, mask_words, li_meas, lm_li, sel_rows, stats, mail, seq, err_idx);
}

// ---- selection + LI gather in ONE launch (pre3_step) --------------------------------------------------------------------------------
// k_ransac_select -> k_gather_li was two dependent launches (7.5 + 7.1 us) for a few hundred integer operations and a copy.  Here every
// workgroup of the gather replays the selection for itself -- the improvement walk of select_body with its early-exit test evaluated by all
// improvement lanes of a chunk at once (the test only depends on the lane's own support; the first lane that passes it is where the
// reference's loop stops), then the winner's mask -> ordered list of LI measurements in LDS -- and gathers its rows of H*P (-> W) or
// computes its entries of S = H*P*H' + I from the ELL rows (k_ell_G's sum, term for term).  One more workgroup (blockIdx.y == ny) runs
// select_body itself: the flags, the list, the counts and the mailbox for everything downstream (k_cholp reads the row count it leaves).
constexpr int SG_RB = 8;                    // rows per workgroup
constexpr int SG_MAXW = 64;                 // mask words a wave takes in one go (m <= 2048)
constexpr int SGL_RB = 4;                   // rows per workgroup of the staged form

// The selection replayed by this workgroup: the ordered list of LI measurement positions into s_sel, their number returned.
__device__ __forceinline__ int select_local_list(int n_draw, int k, int early_exit, int m, const int32_t *support, const uint32_t *masks, int mask_words,
                                                 int *s_sel /* [SG_MAXW * 32] */)
{
    __shared__ int s_pre[SG_MAXW + 1];
    __shared__ uint32_t s_word[SG_MAXW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int best = select_find_best(n_draw, k, early_exit, m, support).best;
    if (wv == 0) {
        // the winner's mask: one word per lane, exclusive prefix of the popcounts
        uint32_t wd = (best >= 0 && lane < mask_words) ? masks[(size_t)best * mask_words + lane] : 0u;
        if (lane == mask_words - 1 && (m & 31)) wd &= (1u << (m & 31)) - 1u;
        int inc = __popc(wd);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        s_word[lane] = wd; s_pre[lane + 1] = inc;
        if (lane == 0) s_pre[0] = 0;
    }
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        const uint32_t wd = s_word[j >> 5];
        if ((wd >> (j & 31)) & 1u) s_sel[s_pre[j >> 5] + __popc(wd & ((1u << (j & 31)) - 1u))] = j;
    }
    __syncthreads();
    return s_pre[mask_words];
}

template <typename T>
__global__ __launch_bounds__(256) void k_select_gather(int n_draw, int k, int early_exit, int m, const int32_t *__restrict__ meas,
                                                       int32_t *support, const uint32_t *masks, int mask_words,
                                                       int32_t *__restrict__ li_meas, int32_t *__restrict__ lm_li, int32_t *__restrict__ sel_rows,
                                                       int32_t *__restrict__ stats, int32_t *mail, int seq,
                                                       int ny, int gxW, const T *__restrict__ HP, T *__restrict__ W, int ldw,
                                                       const int32_t *__restrict__ row_col, const T *__restrict__ row_val, T *__restrict__ S)
{
    if ((int)blockIdx.y == ny) {
        if (blockIdx.x == 0) select_body(n_draw, k, early_exit, m, meas, support, masks, mask_words, li_meas, lm_li, sel_rows, stats, mail, seq);
        return;
    }
    __shared__ int s_sel[SG_MAXW * 32];
    const int tid = threadIdx.x;
    const int r = 2 * select_local_list(n_draw, k, early_exit, m, support, masks, mask_words, s_sel), r_pad = (r + NB - 1) / NB * NB;
    const int a0 = blockIdx.y * SG_RB;
    if (a0 >= r_pad) return;
    if ((int)blockIdx.x < gxW) {
        const int j = (blockIdx.x * 256 + tid) * 4;
        if (j >= ldw) return;
        typedef T v4_t __attribute__((ext_vector_type(4)));
        v4_t v[SG_RB];
#pragma unroll
        for (int q = 0; q < SG_RB; ++q) {
            const int a = a0 + q;
            v[q] = v4_t{ (T)0, (T)0, (T)0, (T)0 };
            if (a < r) v[q] = *reinterpret_cast<const v4_t *>(HP + (size_t)(2 * s_sel[a >> 1] + (a & 1)) * ldw + j);
        }
#pragma unroll
        for (int q = 0; q < SG_RB; ++q) *reinterpret_cast<v4_t *>(W + (size_t)(a0 + q) * ldw + j) = v[q];      // (r_pad is a multiple of 64: all SG_RB rows are inside)
    } else {
        const int b = (blockIdx.x - gxW) * 256 + tid;
        if (b >= r_pad) return;
        // S holds its lower triangle (the factorisation never reads above the diagonal: 0 / 1 there); sel is ascending
        const bool any = b < r && b <= a0 + SG_RB - 1;
        T vv[ELLW]; int cc[ELLW];
        if (any) {
            const int rb = 2 * s_sel[b >> 1] + (b & 1);
#pragma unroll
            for (int t = 0; t < ELLW; ++t) { vv[t] = row_val[rb * ELLW + t]; cc[t] = row_col[rb * ELLW + t]; }
        }
#pragma unroll
        for (int q = 0; q < SG_RB; ++q) {
            const int a = a0 + q;
            T out = (a == b) ? (T)1 : (T)0;
            if (any && a < r && b <= a) {
                const T *hp = HP + (size_t)(2 * s_sel[a >> 1] + (a & 1)) * ldw;
                T g = (T)0;
#pragma unroll
                for (int t = 0; t < ELLW; ++t) g = ell_fma(vv[t], hp[cc[t]], g);
                out += g;
            }
            S[(size_t)a * r_pad + b] = out;
        }
    }
}

// The staged form (the SGL_RB rows of H*P fit in LDS: N = 500 in fp32): a workgroup reads its rows ONCE -- 16-byte loads, on their way to W
// and into LDS -- and takes the entries of S out of LDS.  In the form above every row is read by the W copy and, entry by entry, by the S
// workgroups, from an L2 that another XCD's k_ell_HP_build wrote (13.8 us against ~10).
// RB rows per workgroup, NIT 1024-column pieces of a row held in registers at once: <4, 4> at N = 500 (the four rows fit the 64 KB a launch gets
// without opting in); <1, 12> for long rows (N = 2000: one 48-KB row per workgroup -- round 5: that size used to take the unstaged form above, whose S
// entries are 85 M scattered reads of H*P from memory: 147 us of the step)
template <typename T, int RB = 4, int NIT_ = 4>
__global__ __launch_bounds__(256) void k_select_gather_lds(int n_draw, int k, int early_exit, int m, const int32_t *__restrict__ meas,
                                                           int32_t *support, const uint32_t *masks, int mask_words,
                                                           int32_t *__restrict__ li_meas, int32_t *__restrict__ lm_li, int32_t *__restrict__ sel_rows,
                                                           int32_t *__restrict__ stats, int32_t *mail, int seq,
                                                           int ny, const T *__restrict__ HP, T *__restrict__ W, int ldw,
                                                           const int32_t *__restrict__ row_col, const T *__restrict__ row_val, T *__restrict__ S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
    if ((int)blockIdx.x == ny) {
        select_body(n_draw, k, early_exit, m, meas, support, masks, mask_words, li_meas, lm_li, sel_rows, stats, mail, seq);
        return;
    }
    __shared__ int s_sel[SG_MAXW * 32];
    const int tid = threadIdx.x;
    const int r = 2 * select_local_list(n_draw, k, early_exit, m, support, masks, mask_words, s_sel), r_pad = (r + NB - 1) / NB * NB;
    const int a0 = blockIdx.x * RB;
    if (a0 >= r_pad) return;
    T *rows = reinterpret_cast<T *>(sg_smem);
    typedef T v4_t __attribute__((ext_vector_type(4)));
    const T *src[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) { const int a = a0 + q; src[q] = a < r ? HP + (size_t)(2 * s_sel[a >> 1] + (a & 1)) * ldw : nullptr; }
    // every load of the workgroup is in flight before its first store (round 5): the ELL row of this lane's first column of S, and the four rows of
    // H*P in up to four 16-byte pieces per lane -- they used to be three to four dependent round trips to an L2 that another XCD wrote
    const bool any0 = tid < r && tid <= a0 + RB - 1;
    T vv0[ELLW]; int cc0[ELLW];
#pragma unroll
    for (int t = 0; t < ELLW; ++t) { vv0[t] = (T)0; cc0[t] = 0; }
    if (any0) {
        const int rb = 2 * s_sel[tid >> 1] + (tid & 1);
#pragma unroll
        for (int t = 0; t < ELLW; ++t) { vv0[t] = row_val[rb * ELLW + t]; cc0[t] = row_col[rb * ELLW + t]; }
    }
    {
        constexpr int NIT = NIT_;                           // (<4, 4>: four rows of at most ~3400 columns, four pieces of 1024; <1, 12>: one row of up to 12288)
        v4_t v[NIT][RB];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int j = tid * 4 + it * 1024;
#pragma unroll
            for (int q = 0; q < RB; ++q) v[it][q] = (j < ldw && src[q]) ? *reinterpret_cast<const v4_t *>(src[q] + j) : v4_t{ (T)0, (T)0, (T)0, (T)0 };
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int j = tid * 4 + it * 1024;
            if (j < ldw) {
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    *reinterpret_cast<v4_t *>(W + (size_t)(a0 + q) * ldw + j) = v[it][q];
                    *reinterpret_cast<v4_t *>(rows + (size_t)q * ldw + j) = v[it][q];
                }
            }
        }
        for (int j = tid * 4 + NIT * 1024; j < ldw; j += 1024) {
            v4_t w[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) w[q] = src[q] ? *reinterpret_cast<const v4_t *>(src[q] + j) : v4_t{ (T)0, (T)0, (T)0, (T)0 };
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                *reinterpret_cast<v4_t *>(W + (size_t)(a0 + q) * ldw + j) = w[q];
                *reinterpret_cast<v4_t *>(rows + (size_t)q * ldw + j) = w[q];
            }
        }
    }
    __syncthreads();
    for (int b = tid; b < r_pad; b += 256) {
        const bool any = b < r && b <= a0 + RB - 1;
        T vv[ELLW]; int cc[ELLW];
        if (b == tid) {
#pragma unroll
            for (int t = 0; t < ELLW; ++t) { vv[t] = vv0[t]; cc[t] = cc0[t]; }
        } else if (any) {
            const int rb = 2 * s_sel[b >> 1] + (b & 1);
#pragma unroll
            for (int t = 0; t < ELLW; ++t) { vv[t] = row_val[rb * ELLW + t]; cc[t] = row_col[rb * ELLW + t]; }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int a = a0 + q;
            T out = (a == b) ? (T)1 : (T)0;
            if (any && a < r && b <= a) {
                const T *hp = rows + (size_t)q * ldw;
                T g = (T)0;
#pragma unroll
                for (int t = 0; t < ELLW; ++t) g = ell_fma(vv[t], hp[cc[t]], g);
                out += g;
            }
            S[(size_t)a * r_pad + b] = out;
        }
    }
}

// hi flags (landmark order, written by k_innovation mode 1) -> measurement order + compacted list
__device__ __forceinline__ void collect_hi_body(int m, const int32_t *__restrict__ meas, const int32_t *__restrict__ lm_ic,
                                                const int32_t *__restrict__ lm_li, const int32_t *lm_hi,
                                                int32_t *__restrict__ hi_meas, int32_t *__restrict__ sel_rows, int32_t *__restrict__ stats,
                                                int32_t *mail, int seq)
{
    const int tid = threadIdx.x;
    int cnt = 0;
    // the flag loads (meas -> ic/li/hi: two dependent levels) are issued for 8 chunks at a time so that their latencies overlap;
    // the ordered compaction then only needs ballots
    for (int base0 = 0; base0 < m; base0 += 8 * 64) {
        int in[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = base0 + q * 64 + tid;
            in[q] = 0;
            if (j < m) {
                const int i = meas[j];
                in[q] = (lm_ic[i] == 1 && lm_li[i] == 0) ? lm_hi[i] : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = base0 + q * 64 + tid;
            if (base0 + q * 64 < m) {
                if (j < m) hi_meas[j] = in[q];
                cnt = wave_compact(in[q], tid, cnt, sel_rows, j);
            }
        }
    }
    if (tid == 0) {
        stats[5] = cnt;
        mail[5] = cnt;
        // the device's error words ride along: a wait that gave up (stats[7]) or a factorisation that met a non-positive pivot (stats[6]) in the
        // launches in front of this one -- the step's own LI update -- fails the call that reads this count, not some later pre3_get_state
        mail[6] = stats[6]; mail[7] = stats[7];
        __threadfence_system();
        __hip_atomic_store(&mail[9], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// the HI collection as a launch of its own (one wave), behind the rescue gate's launch
__global__ __launch_bounds__(64) void k_collect_hi(int m, const int32_t *__restrict__ meas, const int32_t *__restrict__ lm_ic, const int32_t *__restrict__ lm_li,
                                                   const int32_t *lm_hi, int32_t *__restrict__ hi_meas, int32_t *__restrict__ sel_rows,
                                                   int32_t *__restrict__ stats, int32_t *mail, int seq)
{
    collect_hi_body(m, meas, lm_ic, lm_li, lm_hi, hi_meas, sel_rows, stats, mail, seq);
}
// The collection can ride in the gate's last workgroup (PRE3_HI_FUSE=1, round 1's form) or follow as its own launch (default): like the
// RANSAC selection (pre3_api.hip, pre3_ransac), the in-launch form makes every workgroup pay a device-scope release and a ticket.
static bool hi_fuse() { static const int e = getenv("PRE3_HI_FUSE") ? atoi(getenv("PRE3_HI_FUSE")) : 0; return e != 0; }
static int launch_collect_hi(pre3_ctx *c, const HiArgs &ha)
{
    hipLaunchKernelGGL(k_collect_hi, dim3(1), dim3(64), 0, c->stream, ha.m, ha.meas, c->lm.ic, c->lm.li, c->lm.hi, ha.hi_meas, ha.sel_rows, ha.stats, ha.mail, ha.seq);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// x_out = x_prior + W' y  (update.m:36), then Jn at the un-normalised quaternion (update.m:42) -> params,
// then normalise (update.m:48).  W: r_pad x ldw, y = column `ld` of W.
template <typename T>
__global__ __launch_bounds__(1024) void k_update_x(int n, int r, const T *__restrict__ W, int ldw, int ld,
                                                   const double *__restrict__ x_prior, double *__restrict__ x_out,
                                                   double *__restrict__ params, unsigned int *__restrict__ tile_ctr)
{
    if (blockIdx.x == 0 && threadIdx.x < 8) tile_ctr[threadIdx.x] = 0;       // tickets of the K9 launch that follows
    __shared__ double red[16][64];
    __shared__ double q[4];
    const int ci = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + ci;            // i < ldw always (ldw >= ld + 64 > n rounded up)
    double s = 0;
#pragma unroll 8
    for (int a = rg; a < r; a += 16) s = fma((double)W[(size_t)a * ldw + i], (double)W[(size_t)a * ldw + ld], s);      // (chain g = rows g mod 16: update_x_block's order)
    red[rg][ci] = s;
    __syncthreads();
    if (rg == 0) {
        s = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][ci];
        if (i < n) s += x_prior[i];
    }
    if (blockIdx.x == 0) {          // block-uniform branch: every thread of block 0 reaches the barrier
        if (rg == 0 && i >= 3 && i < 7) q[i - 3] = s;
        __syncthreads();
        if (rg == 0 && i == 0) { double Jn[16]; d_normjac(q, Jn); for (int t = 0; t < 16; ++t) { params[16 + t] = Jn[t]; params[96 + t] = Jn[t]; } }
        if (rg == 0 && i >= 3 && i < 7) s = s / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    }
    if (rg == 0 && i < n) x_out[i] = s;
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static CamD to_camd(const pre3_cam &c) { return CamD{ c.f, c.Cx, c.Cy, c.k1, c.k2, c.nRows, c.nCols }; }

#define DISPATCH_T(c, expr_f64, expr_f32) do { if ((c)->dtype == PRE3_F64) { expr_f64; } else { expr_f32; } } while (0)

ProjRide make_proj_ride(pre3_ctx *c, int which, int clear_first, int slot, int n_producers)
{
    ProjRide pr{};
    pr.n_blocks = ceil_div(c->N, 64); pr.N = c->N; pr.clear_first = clear_first;
    pr.lm_type = c->lm.type; pr.lm_off = c->lm.off; pr.x = which == PRE3_X_K_K ? c->x_kk : c->x_km1; pr.cam = to_camd(c->cam);
    pr.h = c->lm.h; pr.has_h = c->lm.has_h; pr.Hc = c->lm.Hc; pr.Hl = c->lm.Hl;
    pr.ctr = c->chol_arrive + 3 + slot;
    pr.guard = c->stats + 7;
    c->ride_target[slot] += (unsigned)n_producers;
    pr.target = c->ride_target[slot];
    return pr;
}

// with_projection: the IC-search projection at x_k_km1 (clear_first = 1) rides in the same launch
int launch_predict_impl(pre3_ctx *c, const double u[7], bool with_projection, size_t inbox_n16, int32_t inbox_seq)
{
    U7 uu; for (int i = 0; i < 7; ++i) uu.v[i] = u[i];
    int blocks = ceil_div(c->n, 256);
    ProjRide pr{};
    static const int own_pose = getenv("PRE3_RIDE_POSE") ? atoi(getenv("PRE3_RIDE_POSE")) : 1;      // 0: the riders wait for block 0's pose (rounds 2-4)
    if (with_projection && c->N > 0) {
        pr = make_proj_ride(c, PRE3_X_K_KM1, 1, 0, own_pose ? 0 : 1); pr.x_lm = c->x_kk;      // (one producer: block 0 -- or none: the riders compute the pose themselves)
        pr.own_pose = own_pose ? 1 : 0; pr.x_prev = c->x_kk; pr.u = uu;
    }
    InboxRide ib{ (const int4 *)c->inbox_host_dev, (int4 *)c->inbox_dev, (int)inbox_n16, c->mail_dev, inbox_seq };     // one more block when inbox_n16 > 0
    const int nb = blocks + pr.n_blocks + (inbox_n16 ? 1 : 0);
    const int fuse_jn = c->jn_pending ? 1 : 0;          // the pending update.m:42-46 pass of the update in front (run_update left it to this launch)
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_predict<double>, dim3(nb), dim3(256), 0, c->stream, c->x_kk, c->x_km1, (double *)c->P, c->n, c->ld, uu, c->pred_params, blocks, pr, ib, fuse_jn, PendW{}),
        hipLaunchKernelGGL(k_predict<float>, dim3(nb), dim3(256), 0, c->stream, c->x_kk, c->x_km1, (float *)c->P, c->n, c->ld, uu, c->pred_params, blocks, pr, ib, fuse_jn, pend_args(c)));
    PRE3_HIP(hipGetLastError());
    c->jn_pending = false;
    return PRE3_OK;
}

int launch_jnorm(pre3_ctx *c, int)
{
    PRE3_TRY(pend_flush(c));                    // (reads P: a pending HI down-date goes first)
    int blocks = ceil_div(c->n, 256);
    ProjRide pr{};
    GateRide<float> gr{};
    static const int gate_env = getenv("PRE3_GATE_RIDE") ? atoi(getenv("PRE3_GATE_RIDE")) : 1;      // 0: the gate as a launch of its own behind this one (rounds 3-4)
    if (c->proj_with_jnorm && c->N > 0) {        // (no producer to wait for: x_k_k is complete)
        if (gate_env && c->want_gate_ride && c->jn_q_valid && c->dtype == PRE3_F32) {
            // pre3_step behind a persistent launch whose consumers wrote all of P: projection AND chi2 gate ride here (GateRide)
            gr.n_blocks = ceil_div(c->N, 16); gr.N = c->N; gr.Q = c->jn_q; gr.chi2 = c->rescue_chi2; gr.project = c->proj_in_cholp ? 0 : 1;
            gr.lm_type = c->lm.type; gr.lm_off = c->lm.off; gr.x = c->x_kk; gr.cam = to_camd(c->cam);
            gr.h = c->lm.h; gr.has_h = c->lm.has_h; gr.Hc = c->lm.Hc; gr.Hl = c->lm.Hl; gr.z = c->lm.z; gr.ic = c->lm.ic; gr.li = c->lm.li; gr.hi = c->lm.hi;
            c->rescue_gated = true;
        } else {
            pr = make_proj_ride(c, PRE3_X_K_K, 0, 1, 0);
        }
        c->rescue_projected = true;
    }
    c->proj_with_jnorm = false; c->proj_in_cholp = false;
    c->jn_q_valid = false;                        // (Q belongs to the update that has just been normalised)
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_jnorm_P<double>, dim3(blocks + pr.n_blocks), dim3(256), 0, c->stream, (double *)c->P, c->n, c->ld, c->pred_params, blocks, pr, GateRide<double>{}),
        hipLaunchKernelGGL(k_jnorm_P<float>, dim3(blocks + pr.n_blocks + gr.n_blocks), dim3(256), 0, c->stream, (float *)c->P, c->n, c->ld, c->pred_params, blocks, pr, gr));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_project(pre3_ctx *c, int which, int clear_first)
{
    const double *x = which == PRE3_X_K_K ? c->x_kk : c->x_km1;
    hipLaunchKernelGGL(k_project, dim3(ceil_div(c->N, 64)), dim3(64), 0, c->stream, c->N, c->lm.type, c->lm.off, x, to_camd(c->cam),
                       clear_first, c->lm.h, c->lm.has_h, c->lm.Hc, c->lm.Hl);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}


// project + innovation (+ the HI collection in mode 1) with one kernel boundary less
int launch_project_innovation(pre3_ctx *c, int which, int clear_first, int mode, double chi2, bool collect, bool clear_ic, const IcMatchRide *ride)
{
    PRE3_TRY(pend_flush(c));                    // (reads P: a pending HI down-date goes first)
    const IcMatchRide mr = ride ? *ride : IcMatchRide{};
    const double *x = which == PRE3_X_K_K ? c->x_kk : c->x_km1;
    int32_t *clr = (int32_t *)((unsigned char *)c->inbox_dev + c->off_flags);
    const int n_clr = mode == 0 ? (int)(c->flags_bytes / sizeof(int32_t)) : 0;
    HiArgs ha{};
    if (mode == 1 && collect) ha = HiArgs{ hi_fuse() ? 1 : 0, c->m, ++c->seq_collect, c->meas, c->hi_meas, c->sel_rows, c->stats, c->mail_dev, c->chol_arrive + 2 };
    dim3 g(ceil_div(c->N * 16, 256) + mr.n_blocks), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_project_innovation<double>, g, b, 0, c->stream, c->N, c->lm.type, c->lm.off, x, to_camd(c->cam), clear_first,
                           (const double *)c->P, c->ld, c->lm.Hc, c->lm.Hl, c->lm.has_h, mode, chi2, c->lm.h, c->lm.z, c->lm.ic, c->lm.li,
                           c->lm.hi, c->lm.S, c->lm.has_S, clr, n_clr, ha, clear_ic ? c->lm.ic : nullptr, clear_ic ? c->N : 0, mr),
        hipLaunchKernelGGL(k_project_innovation<float>, g, b, 0, c->stream, c->N, c->lm.type, c->lm.off, x, to_camd(c->cam), clear_first,
                           (const float *)c->P, c->ld, c->lm.Hc, c->lm.Hl, c->lm.has_h, mode, chi2, c->lm.h, c->lm.z, c->lm.ic, c->lm.li,
                           c->lm.hi, c->lm.S, c->lm.has_S, clr, n_clr, ha, clear_ic ? c->lm.ic : nullptr, clear_ic ? c->N : 0, mr));
    PRE3_HIP(hipGetLastError());
    if (mode == 1 && collect && !ha.fuse) PRE3_TRY(launch_collect_hi(c, ha));
    return PRE3_OK;
}

int launch_innovation(pre3_ctx *c, int mode, double chi2, bool clear_flags, bool collect)
{
    PRE3_TRY(pend_flush(c));                    // (reads P: a pending HI down-date goes first)
    int32_t *clr = (int32_t *)((unsigned char *)c->inbox_dev + c->off_flags);
    const int n_clr = clear_flags ? (int)(c->flags_bytes / sizeof(int32_t)) : 0;
    HiArgs ha{};
    if (mode == 1 && collect) ha = HiArgs{ hi_fuse() ? 1 : 0, c->m, ++c->seq_collect, c->meas, c->hi_meas, c->sel_rows, c->stats, c->mail_dev, c->chol_arrive + 2 };
    dim3 g(ceil_div(c->N * 16, 256)), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_innovation<double>, g, b, 0, c->stream, c->N, c->lm.type, c->lm.off, (const double *)c->P, c->ld, c->lm.Hc,
                           c->lm.Hl, c->lm.has_h, mode, chi2, c->lm.h, c->lm.z, c->lm.ic, c->lm.li, c->lm.hi, c->lm.S, c->lm.has_S, clr, n_clr, ha),
        hipLaunchKernelGGL(k_innovation<float>, g, b, 0, c->stream, c->N, c->lm.type, c->lm.off, (const float *)c->P, c->ld, c->lm.Hc,
                           c->lm.Hl, c->lm.has_h, mode, chi2, c->lm.h, c->lm.z, c->lm.ic, c->lm.li, c->lm.hi, c->lm.S, c->lm.has_S, clr, n_clr, ha));
    PRE3_HIP(hipGetLastError());
    if (mode == 1 && collect && !ha.fuse) PRE3_TRY(launch_collect_hi(c, ha));
    return PRE3_OK;
}

int launch_window_gate(pre3_ctx *c, int M, const int32_t *pred_idx_dev, const int32_t *k1_dev, const double *zc_dev, int strict,
                       int32_t *accept_dev)
{
    hipLaunchKernelGGL(k_window_gate, dim3(ceil_div(M, 64)), dim3(64), 0, c->stream, M, pred_idx_dev, k1_dev, zc_dev, c->lm.h, c->lm.S,
                       c->lm.has_S, strict, c->lm.z, c->lm.ic, accept_dev);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// sel_dev == nullptr: all measured rows
int launch_build_rows_impl(pre3_ctx *c, int nsel, const int32_t *sel_dev, int r_pad)
{
    dim3 g(ceil_div(r_pad, 64)), b(64);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_build_rows<double>, g, b, 0, c->stream, nsel, sel_dev, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl,
                           c->lm.z, c->lm.h, c->row_col, (double *)c->row_val, c->row_nu, r_pad),
        hipLaunchKernelGGL(k_build_rows<float>, g, b, 0, c->stream, nsel, sel_dev, c->meas, c->lm.type, c->lm.off, c->lm.Hc, c->lm.Hl,
                           c->lm.z, c->lm.h, c->row_col, (float *)c->row_val, c->row_nu, r_pad));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

constexpr int SCORE_THREADS = 512;      // one measurement per lane up to m = 512 (the per-measurement geometry is the long pole)

template <typename T>
static void launch_score_k(pre3_ctx *c, int k, int nb, size_t shm, int hyp_begin, double threshold, int ldg, int32_t *support_dev,
                           uint32_t *mask_dev, int mask_words, const SelArgs &sel)
{
#define SCORE_ARGS hyp_begin, c->hyp, c->m, c->meas, c->lm.type, c->lm.off, c->x_km1, (const T *)c->HP, c->ldw, c->g_valid ? (const T *)c->G : (const T *)nullptr, ldg, \
                   c->row_nu, c->lm.z, to_camd(c->cam), threshold, support_dev, mask_dev, mask_words, sel, c->row_col, (const T *)c->row_val
    switch (k) {
    case 1: hipLaunchKernelGGL((k_ransac_score<T, 1>), dim3(nb), dim3(SCORE_THREADS), shm, c->stream, SCORE_ARGS); break;
    case 2: hipLaunchKernelGGL((k_ransac_score<T, 2>), dim3(nb), dim3(SCORE_THREADS), shm, c->stream, SCORE_ARGS); break;
    case 3: hipLaunchKernelGGL((k_ransac_score<T, 3>), dim3(nb), dim3(SCORE_THREADS), shm, c->stream, SCORE_ARGS); break;
    default: hipLaunchKernelGGL((k_ransac_score<T, 4>), dim3(nb), dim3(SCORE_THREADS), shm, c->stream, SCORE_ARGS); break;
    }
#undef SCORE_ARGS
}

// select_n_draw > 0: the selection stage (k_ransac_select's work) runs in the last workgroup of this launch
int launch_ransac_score_impl(pre3_ctx *c, int k, double threshold, int hyp_begin, int hyp_end, int ldg, int32_t *support_dev,
                             uint32_t *mask_dev, int mask_words, int select_n_draw, int early_exit)
{
    int nb = hyp_end - hyp_begin;
    if (nb <= 0) return PRE3_OK;
    size_t shm = sizeof(double) * c->m + sizeof(uint32_t) * mask_words + 16;
    SelArgs sel{};
    if (select_n_draw > 0) {
        sel = SelArgs{ 1, select_n_draw, k, early_exit, ++c->seq_select, c->li_meas, c->lm.li, c->sel_rows, c->stats, c->mail_dev, c->chol_arrive + 1 };
    }
    DISPATCH_T(c,
        launch_score_k<double>(c, k, nb, shm, hyp_begin, threshold, ldg, support_dev, mask_dev, mask_words, sel),
        launch_score_k<float>(c, k, nb, shm, hyp_begin, threshold, ldg, support_dev, mask_dev, mask_words, sel));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_ransac_select_impl(pre3_ctx *c, int n_draw, int k, int early_exit, int32_t *support_dev, const uint32_t *mask_dev, int mask_words, int err_idx)
{
    hipLaunchKernelGGL(k_ransac_select, dim3(1), dim3(256), 0, c->stream, n_draw, k, early_exit, c->m, c->meas, support_dev, mask_dev,
                       mask_words, c->li_meas, c->lm.li, c->sel_rows, c->stats, c->mail_dev, ++c->seq_select, err_idx);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

bool select_gather_usable(const pre3_ctx *c)
{
    static const int env = getenv("PRE3_SELECT_GATHER") ? atoi(getenv("PRE3_SELECT_GATHER")) : 1;
    return env && c->m > 0 && ceil_div(c->m, 32) <= SG_MAXW;
}

int launch_select_gather(pre3_ctx *c, int n_draw, int k, int early_exit, int mask_words)
{
    const int r_pad_max = round_up(2 * c->m, NB);
    const int seq = ++c->seq_select;
    static const int lds_env = getenv("PRE3_SELECT_GATHER_LDS") ? atoi(getenv("PRE3_SELECT_GATHER_LDS")) : 1;
    const size_t stage = (size_t)SGL_RB * c->ldw * (c->dtype == PRE3_F64 ? 8 : 4);
    const size_t stage1 = stage / SGL_RB;
    if (lds_env && stage > 55 * 1024 && stage1 <= 55 * 1024 && c->dtype == PRE3_F32) {
        // long rows: one row per workgroup
        const int ny = r_pad_max;
        dim3 g(ny + 1), b(256);
        hipLaunchKernelGGL((k_select_gather_lds<float, 1, 12>), g, b, stage1, c->stream, n_draw, k, early_exit, c->m, c->meas, c->support, c->masks, mask_words, c->li_meas, c->lm.li,
                           c->sel_rows, c->stats, c->mail_dev, seq, ny, (const float *)c->HP, (float *)c->W, c->ldw, c->row_col, (const float *)c->row_val, (float *)c->Smat);
        PRE3_HIP(hipGetLastError());
        return PRE3_OK;
    }
    if (lds_env && stage <= 55 * 1024) {          // (+ 8.8 KB of static LDS: inside the 64 KB a launch gets without opting in)
        const int ny = r_pad_max / SGL_RB;
        dim3 g(ny + 1), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL((k_select_gather_lds<double, SGL_RB, 4>), g, b, stage, c->stream, n_draw, k, early_exit, c->m, c->meas, c->support, c->masks, mask_words, c->li_meas, c->lm.li,
                               c->sel_rows, c->stats, c->mail_dev, seq, ny, (const double *)c->HP, (double *)c->W, c->ldw, c->row_col, (const double *)c->row_val, (double *)c->Smat),
            hipLaunchKernelGGL((k_select_gather_lds<float, SGL_RB, 4>), g, b, stage, c->stream, n_draw, k, early_exit, c->m, c->meas, c->support, c->masks, mask_words, c->li_meas, c->lm.li,
                               c->sel_rows, c->stats, c->mail_dev, seq, ny, (const float *)c->HP, (float *)c->W, c->ldw, c->row_col, (const float *)c->row_val, (float *)c->Smat));
        PRE3_HIP(hipGetLastError());
        return PRE3_OK;
    }
    const int ny = r_pad_max / SG_RB;
    const int gxW = ceil_div(c->ldw / 4, 256);
    dim3 g(gxW + ceil_div(r_pad_max, 256), ny + 1), b(256);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_select_gather<double>, g, b, 0, c->stream, n_draw, k, early_exit, c->m, c->meas, c->support, c->masks, mask_words, c->li_meas, c->lm.li,
                           c->sel_rows, c->stats, c->mail_dev, seq, ny, gxW, (const double *)c->HP, (double *)c->W, c->ldw, c->row_col,
                           (const double *)c->row_val, (double *)c->Smat),
        hipLaunchKernelGGL(k_select_gather<float>, g, b, 0, c->stream, n_draw, k, early_exit, c->m, c->meas, c->support, c->masks, mask_words, c->li_meas, c->lm.li,
                           c->sel_rows, c->stats, c->mail_dev, seq, ny, gxW, (const float *)c->HP, (float *)c->W, c->ldw, c->row_col,
                           (const float *)c->row_val, (float *)c->Smat));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int launch_update_x(pre3_ctx *c, int which_prior, int r)
{
    const double *xp = which_prior == PRE3_X_K_K ? c->x_kk : c->x_km1;
    dim3 g(ceil_div(c->n, 64)), b(1024);
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_update_x<double>, g, b, 0, c->stream, c->n, r, (const double *)c->W, c->ldw, c->ld, xp, c->x_kk, c->pred_params, c->tile_ctr),
        hipLaunchKernelGGL(k_update_x<float>, g, b, 0, c->stream, c->n, r, (const float *)c->W, c->ldw, c->ld, xp, c->x_kk, c->pred_params, c->tile_ctr));
    c->tile_ctr_clean = true;
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}


__global__ __launch_bounds__(1024) void k_inbox_pull(InboxRide ib) { inbox_pull_block(ib); }

// A rank's slice of a sharded RANSAC round starts with three small things -- the supports + masks zeroed (the slices are disjoint: the
// all-reduce's integer sum is their union), the draw table pulled out of the pinned inbox, and the measurements its hypotheses draw marked
// (need[s] = tag: a fresh tag per round, nothing to clear) -- which were a memset, k_inbox_pull and k_mark_needed: one launch (workgroup 0: pull + marks).
__global__ __launch_bounds__(1024) void k_slice_prepare(InboxRide ib, int32_t *__restrict__ zero, int n_zero, const int32_t *hyp, int k, int lo, int hi,
                                                        int32_t *__restrict__ need, int tag)
{
    if (blockIdx.x > 0) {                           // workgroups 1..: the clear, 16 bytes per lane
        int4 *z4 = reinterpret_cast<int4 *>(zero);
        for (int i = (blockIdx.x - 1) * 1024 + threadIdx.x; i < n_zero / 4; i += (gridDim.x - 1) * 1024) z4[i] = int4{ 0, 0, 0, 0 };
        if (blockIdx.x == 1 && (int)threadIdx.x < (n_zero & 3)) zero[(n_zero & ~3) + threadIdx.x] = 0;
        return;
    }
    if (ib.n16 > 0) inbox_pull_block(ib);          // (ends in a barrier: the table is in device memory, written by this workgroup)
    else __syncthreads();
    for (int t = threadIdx.x; t < (hi - lo) * k; t += 1024) need[hyp[lo * k + t]] = tag;
}
int launch_slice_prepare(pre3_ctx *c, const void *src_host_mapped, size_t n16, int32_t seq, int n_zero, int k, int lo, int hi, int tag)
{
    hipLaunchKernelGGL(k_slice_prepare, dim3(1 + (n_zero > 0 ? std::min(16, ceil_div(n_zero, 4096)) : 0)), dim3(1024), 0, c->stream, InboxRide{ (const int4 *)src_host_mapped, (int4 *)c->hyp, (int)n16, c->mail_dev, seq, 10 },
                       c->support, n_zero, c->hyp, k, lo, hi, c->need, tag);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}
int launch_inbox_pull(pre3_ctx *c, const void *src_host_mapped, void *dst_dev, size_t n16, int32_t seq, int slot, int32_t *clear, int n_clear)
{
    hipLaunchKernelGGL(k_inbox_pull, dim3(1), dim3(1024), 0, c->stream, InboxRide{ (const int4 *)src_host_mapped, (int4 *)dst_dev, (int)n16, c->mail_dev, seq, slot, clear, n_clear });
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// ---- stateless compute_hypothesis_support_fast.m:27-116 ------------------------------------------------------------------
// One workgroup (the reference evaluates one hypothesis per call; a frame has a few hundred measurements): lane j projects
// measurement j of the hypothesis state xi exactly as k_ransac_score does (un-normalised quaternion, quirk Q4), the workgroup takes
// min(residual) over the inverse-depth measurements and applies `residual < min + threshold` (:70) / `residual < threshold` (:109).
// i1/i2/i3/i4: the state entries the four columns of state_vector_pattern select, in order (xi(logical(pattern(:,c)))).
__global__ __launch_bounds__(256) void k_hyp_support(const double *__restrict__ xi, CamD cam, int n_id, const int32_t *__restrict__ i1,
                                                     const int32_t *__restrict__ i2, const int32_t *__restrict__ i3, const double *__restrict__ z_id,
                                                     int n_euc, const int32_t *__restrict__ i4, const double *__restrict__ z_euc, double threshold,
                                                     double *__restrict__ res, int32_t *__restrict__ out /* [0] support, then pos_id[n_id], pos_euc[n_euc] */)
{
    __shared__ double s_red[4];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double rot[9];
    d_q2r(xi + 3, rot);
    double lmin = INFINITY;
    for (int j = tid; j < n_id + n_euc; j += 256) {
        const bool id = j < n_id;
        const int e = j - n_id;
        double y[6];
        if (id) { y[0] = xi[i1[3 * j]]; y[1] = xi[i1[3 * j + 1]]; y[2] = xi[i1[3 * j + 2]]; y[3] = xi[i2[2 * j]]; y[4] = xi[i2[2 * j + 1]]; y[5] = xi[i3[j]]; }
        else { y[0] = xi[i4[3 * e]]; y[1] = xi[i4[3 * e + 1]]; y[2] = xi[i4[3 * e + 2]]; y[3] = y[4] = y[5] = 0; }
        double v[3], hc[3], uvd[2];
        d_ray(id ? PRE3_INVDEPTH : PRE3_CARTESIAN, y, xi, v);
#pragma unroll
        for (int c = 0; c < 3; ++c) hc[c] = rot[0 * 3 + c] * v[0] + rot[1 * 3 + c] * v[1] + rot[2 * 3 + c] * v[2];
        d_pinhole_distort(hc, cam, uvd);
        const double *z = id ? z_id + 2 * j : z_euc + 2 * e;
        const double n0 = z[0] - uvd[0], n1 = z[1] - uvd[1];
        const double r = sqrt(n0 * n0 + n1 * n1);
        res[j] = r;
        if (id) lmin = fmin(lmin, r);
    }
    lmin = wave_min(lmin);
    if (lane == 0) s_red[wv] = lmin;
    __syncthreads();
    const double minres = fmin(fmin(s_red[0], s_red[1]), fmin(s_red[2], s_red[3]));
    int cnt = 0;
    for (int j = tid; j < n_id + n_euc; j += 256) {
        const int in = j < n_id ? (res[j] < (minres + threshold)) : (res[j] < threshold);      // NaN compares false, as in MATLAB
        out[1 + j] = in; cnt += in;
    }
    cnt = wave_sum(cnt);
    if (lane == 0) s_cnt[wv] = cnt;
    __syncthreads();
    if (tid == 0) out[0] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

int run_hypothesis_support(int n, const double *xi, const pre3_cam &cam, int n_id, const int32_t *i1, const int32_t *i2, const int32_t *i3,
                           const double *z_id, int n_euc, const int32_t *i4, const double *z_euc, double threshold, int32_t *out_host)
{
    const int m = n_id + n_euc;
    // one pooled device block: [xi n | z 2m | res m] doubles, then [i1 3n_id | i2 2n_id | i3 n_id | i4 3n_euc | out 1+m] int32
    const size_t nd = (size_t)n + 3 * (size_t)m, ni = 6 * (size_t)n_id + 3 * (size_t)n_euc + 1 + m;
    void *blk = nullptr; int slot = -1;
    PRE3_TRY(scratch_acquire(sizeof(double) * nd + sizeof(int32_t) * ni, &blk, &slot));
    struct Rel { int slot; void *p; ~Rel() { scratch_release(slot, p); } } rel{ slot, blk };
    std::vector<double> hd(nd, 0.0);
    std::vector<int32_t> hi(ni, 0);
    memcpy(hd.data(), xi, sizeof(double) * n);
    if (n_id) memcpy(hd.data() + n, z_id, sizeof(double) * 2 * n_id);
    if (n_euc) memcpy(hd.data() + n + 2 * (size_t)n_id, z_euc, sizeof(double) * 2 * n_euc);
    if (n_id) { memcpy(hi.data(), i1, sizeof(int32_t) * 3 * n_id); memcpy(hi.data() + 3 * (size_t)n_id, i2, sizeof(int32_t) * 2 * n_id); memcpy(hi.data() + 5 * (size_t)n_id, i3, sizeof(int32_t) * n_id); }
    if (n_euc) memcpy(hi.data() + 6 * (size_t)n_id, i4, sizeof(int32_t) * 3 * n_euc);
    double *dd = (double *)blk; int32_t *di = (int32_t *)(dd + nd);
    PRE3_HIP(hipMemcpy(dd, hd.data(), sizeof(double) * nd, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(di, hi.data(), sizeof(int32_t) * ni, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_hyp_support, dim3(1), dim3(256), 0, 0, dd, to_camd(cam), n_id, di, di + 3 * (size_t)n_id, di + 5 * (size_t)n_id, dd + n,
                       n_euc, di + 6 * (size_t)n_id, dd + n + 2 * (size_t)n_id, threshold, dd + n + 2 * (size_t)m, di + 6 * (size_t)n_id + 3 * (size_t)n_euc);
    PRE3_HIP(hipGetLastError());
    PRE3_HIP(hipMemcpy(out_host, di + 6 * (size_t)n_id + 3 * (size_t)n_euc, sizeof(int32_t) * (1 + m), hipMemcpyDeviceToHost));
    return PRE3_OK;
}

}  // namespace pre3
