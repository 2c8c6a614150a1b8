/*
 * pre3_oracle.c -- CPU restatement (plain C, IEEE fp64, scalar, single-threaded) of the
 * per-step hot path of the MATLAB reference ahtamjidi/3PRE (1-point-RANSAC EKF SLAM, SR4000).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path (3pre_amd/) never does.
 *
 * Every function follows one reference function statement by statement and cites it
 * (paths relative to /root/reference/matlab_code/).  MATLAB arrays are column-major; the
 * covariance is symmetric, so row-/column-major coincide for P.  Indices here are 0-based.
 *
 * Parity pin: tests/golden/sr4000_step3.npz (MATLAB-produced snapshot of a real SR4000 step)
 * pins orc_project / orc_jacobian / orc_innovation / orc_update / orc_rescue / orc_support;
 * tests/golden/siftmatch_kat.json pins orc_siftmatch_* on the reference's own box.sift.
 * UNPINNED by any reference artefact (checked against the independent numpy twin and by
 * property tests only): orc_predict, the Cartesian-landmark branches, the RANSAC hypothesis
 * sequence (MATLAB's legacy RNG is an input here), orc_knn beyond its docstring example.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define ORC_API __attribute__((visibility("default")))

/* camera: f Cx Cy k1 k2 nRows nCols (initialize_cam.m:69-78) */
typedef struct { double f, Cx, Cy, k1, k2, nRows, nCols; } orc_cam;

enum { ORC_INVDEPTH = 0, ORC_CARTESIAN = 1 };

/* ------------------------------------------------------------------ small dense helpers */

/* C(m x n) = A(m x k) * B(k x n), row-major.  Every C(i,j) is the sum over t = 0..k-1 in that order; B is transposed once so the
 * inner loop is contiguous, and rows are independent (OpenMP over i): the bits of C do not depend on the thread count. */
static void mm(const double *A, const double *B, double *C, int m, int k, int n)
{
    double *Bt = (double *)malloc(sizeof(double) * (size_t)k * n);
    for (int t = 0; t < k; ++t) for (int j = 0; j < n; ++j) Bt[(size_t)j * k + t] = B[(size_t)t * n + j];
#pragma omp parallel for schedule(static) if ((long)m * n * k > 2000000L)
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            const double *a = A + (size_t)i * k, *b = Bt + (size_t)j * k;
            for (int t = 0; t < k; ++t) s += a[t] * b[t];
            C[(size_t)i * n + j] = s;
        }
    free(Bt);
}

/* thread count of the OpenMP loops (bench.py times the restatement at 1 thread and at all cores; results are identical) */
ORC_API void orc_set_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t > 0 ? t : omp_get_num_procs());
#else
    (void)t;
#endif
}
ORC_API int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* positions of the non-zeros of each of the r rows of a dense r x n matrix, ascending: MATLAB's sparse H*P visits exactly these, and
 * skipping the zeros leaves every partial sum unchanged (0*x contributes an exact 0) */
typedef struct { int *idx; int *start; } nzrows;
static nzrows nz_build(const double *H, int r, int n)
{
    nzrows z; z.start = (int *)malloc(sizeof(int) * (r + 1)); int cnt = 0;
    for (int a = 0; a < r; ++a) for (int t = 0; t < n; ++t) if (H[(size_t)a * n + t] != 0.0) ++cnt;
    z.idx = (int *)malloc(sizeof(int) * (cnt ? cnt : 1)); cnt = 0;
    for (int a = 0; a < r; ++a) { z.start[a] = cnt; for (int t = 0; t < n; ++t) if (H[(size_t)a * n + t] != 0.0) z.idx[cnt++] = t; }
    z.start[r] = cnt;
    return z;
}
static void nz_free(nzrows z) { free(z.idx); free(z.start); }

/* general inverse by LU with partial pivoting (MATLAB `inv` is LAPACK dgetrf+dgetri) */
static int inv_lu(const double *A, double *Ainv, int n)
{
    double *M = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    if (!M || !piv) { free(M); free(piv); return -1; }
    memcpy(M, A, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(M[c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(M[r * n + c]) > best) { best = fabs(M[r * n + c]); p = r; }
        if (best == 0.0) { free(M); free(piv); return -2; }
        if (p != c) {
            for (int j = 0; j < n; ++j) { double t = M[c * n + j]; M[c * n + j] = M[p * n + j]; M[p * n + j] = t; }
            int t = piv[c]; piv[c] = piv[p]; piv[p] = t;
        }
        double d = M[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            double l = M[r * n + c] / d;
            M[r * n + c] = l;
            if (l != 0.0) for (int j = c + 1; j < n; ++j) M[r * n + j] -= l * M[c * n + j];
        }
    }
    /* solve for each unit vector (columns are independent: OpenMP over e, same arithmetic per column) */
#pragma omp parallel if (n > 128)
    {
        double *y = (double *)malloc(sizeof(double) * n), *c = (double *)malloc(sizeof(double) * n);
#pragma omp for schedule(static)
        for (int e = 0; e < n; ++e) {
            for (int i = 0; i < n; ++i) {
                double s = (piv[i] == e) ? 1.0 : 0.0;
                for (int j = 0; j < i; ++j) s -= M[i * n + j] * y[j];
                y[i] = s;
            }
            for (int i = n - 1; i >= 0; --i) {
                double s = y[i];
                for (int j = i + 1; j < n; ++j) s -= M[i * n + j] * c[j];
                c[i] = s / M[i * n + i];
            }
            for (int i = 0; i < n; ++i) Ainv[i * n + e] = c[i];
        }
        free(y); free(c);
    }
    free(M); free(piv);
    return 0;
}

static void inv2(const double A[4], double B[4])
{
    /* 2x2 inverse through the same pivoted LU so rounding follows inv() */
    inv_lu(A, B, 2);
}

static void inv3(const double A[9], double B[9]) { inv_lu(A, B, 3); }

/* q2r.m:29-36 (Civera) */
static void q2r(const double q[4], double R[9])
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = r * r + x * x - y * y - z * z; R[1] = 2 * (x * y - r * z);           R[2] = 2 * (z * x + r * y);
    R[3] = 2 * (x * y + r * z);           R[4] = r * r - x * x + y * y - z * z; R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (z * x - r * y);           R[7] = 2 * (y * z + r * x);           R[8] = r * r - x * x - y * y + z * z;
}

/* slamToolbox_11_02_18/FrameTransforms/Rotations/q2R.m:18-34 (Sola) -- same algebra, different rounding */
static void q2R_sola(const double q[4], double R[9])
{
    double a = q[0], b = q[1], c = q[2], d = q[3];
    double aa = a * a, ab = 2 * a * b, ac = 2 * a * c, ad = 2 * a * d;
    double bb = b * b, bc = 2 * b * c, bd = 2 * b * d, cc = c * c, cd = 2 * c * d, dd = d * d;
    R[0] = aa + bb - cc - dd; R[1] = bc - ad;           R[2] = bd + ac;
    R[3] = bc + ad;           R[4] = aa - bb + cc - dd; R[5] = cd - ab;
    R[6] = bd - ac;           R[7] = cd + ab;           R[8] = aa - bb - cc + dd;
}

/* qProd.m:16-33 */
static void qprod(const double q1[4], const double q2[4], double q[4], double Qq1[16], double Qq2[16])
{
    double a = q1[0], b = q1[1], c = q1[2], d = q1[3];
    double w = q2[0], x = q2[1], y = q2[2], z = q2[3];
    q[0] = a * w - b * x - c * y - d * z;
    q[1] = a * x + b * w + c * z - d * y;
    q[2] = a * y - b * z + c * w + d * x;
    q[3] = a * z + b * y - c * x + d * w;
    if (Qq1) {
        double M[16] = { w, -x, -y, -z,  x, w, z, -y,  y, -z, w, x,  z, y, -x, w };
        memcpy(Qq1, M, sizeof M);
    }
    if (Qq2) {
        double M[16] = { a, -b, -c, -d,  b, a, -d, c,  c, d, a, -b,  d, -c, b, a };
        memcpy(Qq2, M, sizeof M);
    }
}

/* normJac.m:27-38 */
static void normjac(const double q[4], double J[16])
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    double s = pow(r * r + x * x + y * y + z * z, -1.5);
    double M[16] = {
        x * x + y * y + z * z, -r * x, -r * y, -r * z,
        -x * r, r * r + y * y + z * z, -x * y, -x * z,
        -y * r, -y * x, r * r + x * x + z * z, -y * z,
        -z * r, -z * x, -z * y, r * r + x * x + y * y };
    for (int i = 0; i < 16; ++i) J[i] = s * M[i];
}

/* rows/cols 4:7 <- Jnorm (update.m:42-46, predict_state_and_covariance.m:137-141).
 * The reference rebuilds the matrix from blocks of the OLD matrix; blocks (4:7,4:7) get
 * Jnorm*P*Jnorm'. */
static void apply_jnorm(double *P, int n, const double J[16])
{
    double *rows = (double *)malloc(sizeof(double) * 4 * n);
    double *cols = (double *)malloc(sizeof(double) * 4 * n);
    /* Jnorm*P(4:7,:) and P(:,4:7)*Jnorm' from the old matrix */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0, c = 0;
            for (int t = 0; t < 4; ++t) { s += J[i * 4 + t] * P[(3 + t) * n + j]; c += P[j * n + 3 + t] * J[i * 4 + t]; }
            rows[i * n + j] = s; cols[i * n + j] = c;
        }
    /* corner: Jnorm*P(4:7,4:7)*Jnorm' evaluated left to right */
    double JP[16], C[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) JP[i * 4 + j] = rows[i * n + 3 + j];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += JP[i * 4 + t] * J[j * 4 + t];
            C[i * 4 + j] = s;
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < n; ++j) { P[(3 + i) * n + j] = rows[i * n + j]; P[j * n + 3 + i] = cols[i * n + j]; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) P[(3 + i) * n + 3 + j] = C[i * 4 + j];
    free(rows); free(cols);
}

/* ------------------------------------------------------------------ a2: prediction */

/* e2q.m:14-36 Jacobian Qe only (the quaternion itself is unused by the caller) */
static void e2q_jac(const double e[3], double Qe[12])
{
    double sr = sin(e[0] / 2), sp = sin(e[1] / 2), sy = sin(e[2] / 2);
    double cr = cos(e[0] / 2), cp = cos(e[1] / 2), cy = cos(e[2] / 2);
    double M[12] = {
        -cy * cp * sr + sy * sp * cr, -cy * sp * cr + sy * cp * sr, -sy * cp * cr + cy * sp * sr,
         cy * cp * cr + sy * sp * sr, -cy * sp * sr - sy * cp * cr, -sy * cp * sr - cy * sp * cr,
        -cy * sp * sr + sy * cp * cr,  cy * cp * cr - sy * sp * sr, -sy * sp * cr + cy * cp * sr,
        -sy * cp * sr - cy * sp * cr, -cy * cp * sr - sy * sp * cr,  cy * cp * cr + sy * sp * sr };
    for (int i = 0; i < 12; ++i) Qe[i] = 0.5 * M[i];
}

/* constant process noise Pn (7x7), predict_state_and_covariance.m:98-102 */
ORC_API void orc_process_noise(double Pn[49])
{
    memset(Pn, 0, sizeof(double) * 49);
    double sx = 0.01 / 3; sx = sx * sx;
    Pn[0] = Pn[8] = Pn[16] = sx;
    double a = 0.24 / 2 * M_PI / 180;
    double e[3] = { a * 1, a * 0.1, a * 1 };
    double Qe[12];
    e2q_jac(e, Qe);
    double D[3] = { e[0] * e[0], e[1] * e[1], e[2] * e[2] };
    /* cov_dq = Qe*diag(.)*Qe' evaluated left to right */
    double QD[12];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) QD[i * 3 + j] = Qe[i * 3 + j] * D[j];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += QD[i * 3 + t] * Qe[j * 3 + t];
            Pn[(3 + i) * 7 + 3 + j] = s;
        }
}

/* predict_state_and_covariance.m:59-143 with aux_code/odometry_model.m:44-68.
 * u = [dX(3); dq(4)] is an input (the reference fetches it from disk through fv.m:47). */
ORC_API int orc_predict(int n, const double *x, const double *P, const double u[7], double *x_out, double *P_out)
{
    if (n < 13) return -1;
    const double *q = x + 3;
    double R[9], qn[4], Qq1[16], Qq2[16];
    q2R_sola(q, R);
    /* odometry_model.m:50-51: x = x + q2R(q)*dx ; [q,Qq1,Qq2] = qProd(q,dq) */
    for (int i = 0; i < 3; ++i) x_out[i] = x[i] + (R[i * 3] * u[0] + R[i * 3 + 1] * u[1] + R[i * 3 + 2] * u[2]);
    qprod(q, u + 3, qn, Qq1, Qq2);
    for (int i = 0; i < 4; ++i) x_out[3 + i] = qn[i];
    /* :79  velocities zeroed, landmarks copied */
    for (int i = 7; i < 13; ++i) x_out[i] = 0;
    for (int i = 13; i < n; ++i) x_out[i] = x[i];
    /* :83-86 F = [Xo_x 0; 0 I6] (13x13), G = [Xo_u; 0] (13x7) */
    double F[169] = { 0 }, G[91] = { 0 };
    for (int i = 0; i < 3; ++i) F[i * 13 + i] = 1;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) F[(3 + i) * 13 + 3 + j] = Qq1[i * 4 + j];
    for (int i = 7; i < 13; ++i) F[i * 13 + i] = 1;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) G[i * 7 + j] = R[i * 3 + j];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) G[(3 + i) * 7 + 3 + j] = Qq2[i * 4 + j];
    /* :98-102,120  Q = G*Pn*G' */
    double Pn[49], GP[91], Q[169];
    orc_process_noise(Pn);
    mm(G, Pn, GP, 13, 7, 7);
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) {
            double s = 0;
            for (int t = 0; t < 7; ++t) s += GP[i * 7 + t] * G[j * 7 + t];
            Q[i * 13 + j] = s;
        }
    /* :131-132 block rebuild */
    memcpy(P_out, P, sizeof(double) * (size_t)n * n);
    double FP[169];
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) {
            double s = 0;
            for (int t = 0; t < 13; ++t) s += F[i * 13 + t] * P[t * n + j];
            FP[i * 13 + j] = s;
        }
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) {
            double s = 0;
            for (int t = 0; t < 13; ++t) s += FP[i * 13 + t] * F[j * 13 + t];
            P_out[i * n + j] = s + Q[i * 13 + j];
        }
    for (int i = 0; i < 13; ++i)
        for (int j = 13; j < n; ++j) {
            double s = 0, c = 0;
            for (int t = 0; t < 13; ++t) { s += F[i * 13 + t] * P[t * n + j]; c += P[j * n + t] * F[i * 13 + t]; }
            P_out[i * n + j] = s;   /* F*P(1:13,14:end) */
            P_out[j * n + i] = c;   /* P(14:end,1:13)*F' */
        }
    /* :137-143 quaternion normalisation Jacobian at the un-normalised q, then normalise */
    double J[16];
    normjac(x_out + 3, J);
    apply_jnorm(P_out, n, J);
    double nq = sqrt(x_out[3] * x_out[3] + x_out[4] * x_out[4] + x_out[5] * x_out[5] + x_out[6] * x_out[6]);
    for (int i = 0; i < 4; ++i) x_out[3 + i] /= nq;
    return 0;
}

/* ------------------------------------------------------------------ a3: measurement model */

/* m.m:38-40 */
static void m_dir(double theta, double phi, double m[3])
{
    double cphi = cos(phi);
    m[0] = cphi * sin(theta); m[1] = -sin(phi); m[2] = cphi * cos(theta);
}

/* hu_my_version.m:41-42 then distort_fm_my_version.m:52-61 */
static void pinhole_distort(const double hrl[3], const orc_cam *cam, double uvd[2])
{
    double uu = cam->Cx + (hrl[0] / hrl[2]) * cam->f;
    double vu = cam->Cy + (hrl[1] / hrl[2]) * cam->f;
    double xu = (uu - cam->Cx) / cam->f, yu = (vu - cam->Cy) / cam->f;
    double ru = sqrt(xu * xu + yu * yu);
    double D = 1 + cam->k1 * (ru * ru) + cam->k2 * pow(ru, 4);
    double xd = xu * D, yd = yu * D;
    uvd[0] = xd * cam->f + cam->Cx;
    uvd[1] = yd * cam->f + cam->Cy;
}

/* hi_inverse_depth.m:33-85 / hi_cartesian.m:33-81.  Returns 1 and writes zi when predicted. */
static int hi_landmark(int type, const double *y, const double t_wc[3], const double r_wc[9], const orc_cam *cam, double zi[2])
{
    double v[3], hrl[3];
    if (type == ORC_INVDEPTH) {
        double mi[3];
        m_dir(y[3], y[4], mi);
        for (int i = 0; i < 3; ++i) v[i] = (y[i] - t_wc[i]) * y[5] + mi[i];
        /* r_cw = r_wc' */
        for (int i = 0; i < 3; ++i) hrl[i] = r_wc[0 * 3 + i] * v[0] + r_wc[1 * 3 + i] * v[1] + r_wc[2 * 3 + i] * v[2];
    } else {
        double rcw[9];
        inv3(r_wc, rcw);            /* hi_cartesian.m:33 uses inv(r_wc) */
        for (int i = 0; i < 3; ++i) v[i] = y[i] - t_wc[i];
        for (int i = 0; i < 3; ++i) hrl[i] = rcw[i * 3] * v[0] + rcw[i * 3 + 1] * v[1] + rcw[i * 3 + 2] * v[2];
    }
    double ax = atan2(hrl[0], hrl[2]) * 180 / M_PI, ay = atan2(hrl[1], hrl[2]) * 180 / M_PI;
    if (ax < -60 || ax > 60 || ay < -60 || ay > 60) return 0;
    double uvd[2];
    pinhole_distort(hrl, cam, uvd);
    if (uvd[0] > 0 && uvd[0] < cam->nCols && uvd[1] > 0 && uvd[1] < cam->nRows) { zi[0] = uvd[0]; zi[1] = uvd[1]; return 1; }
    return 0;
}

/* predict_camera_measurements.m:27-68.  lm_type[N]; landmark i's parameters start at x[lm_off[i]].
 * has_h[i] is in/out: a landmark that is not predicted now keeps its previous h (quirk Q7). */
ORC_API int orc_project(int N, const int *lm_type, const int *lm_off, const double *x, const orc_cam *cam, double *h, int *has_h)
{
    double r_wc[9];
    q2r(x + 3, r_wc);
    for (int i = 0; i < N; ++i) {
        double zi[2];
        if (hi_landmark(lm_type[i], x + lm_off[i], x, r_wc, cam, zi)) { h[2 * i] = zi[0]; h[2 * i + 1] = zi[1]; has_h[i] = 1; }
    }
    return 0;
}

/* ------------------------------------------------------------------ a4: Jacobians */

/* jacob_distor_fm_my_version.m:47-61 */
static void jacob_distor(const orc_cam *cam, const double uv[2], double J[4])
{
    double Cx = cam->Cx, Cy = cam->Cy, k1 = cam->k1, k2 = cam->k2, f = cam->f;
    double u = uv[0], v = uv[1];
    double x = u - Cx, y = v - Cy;
    double r2 = (x * x + y * y) / (f * f);
    double r4 = r2 * r2;
    J[0] = (1 + k1 * r2 + k2 * r4) + (u - Cx) * (k1 + 2 * k2 * r2) * (2 * (u - Cx) / (f * f));
    J[3] = (1 + k1 * r2 + k2 * r4) + (v - Cy) * (k1 + 2 * k2 * r2) * (2 * (v - Cy) / (f * f));
    J[1] = (u - Cx) * (k1 + 2 * k2 * r2) * (2 * (v - Cy) / (f * f));
    J[2] = (v - Cy) * (k1 + 2 * k2 * r2) * (2 * (u - Cx) / (f * f));
}

/* dRq_times_a_by_dq.m:29-101 */
static void dRq_times_a_by_dq(const double q[4], const double a[3], double out[12])
{
    double q0 = q[0], qx = q[1], qy = q[2], qz = q[3];
    double d0[9] = { 2 * q0, -2 * qz, 2 * qy,  2 * qz, 2 * q0, -2 * qx,  -2 * qy, 2 * qx, 2 * q0 };
    double dx[9] = { 2 * qx, 2 * qy, 2 * qz,  2 * qy, -2 * qx, -2 * q0,  2 * qz, 2 * q0, -2 * qx };
    double dy[9] = { -2 * qy, 2 * qx, 2 * q0,  2 * qx, 2 * qy, 2 * qz,  -2 * q0, 2 * qz, -2 * qy };
    double dz[9] = { -2 * qz, -2 * q0, 2 * qx,  2 * q0, -2 * qz, 2 * qy,  2 * qx, 2 * qy, 2 * qz };
    const double *D[4] = { d0, dx, dy, dz };
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 3; ++i)
            out[i * 4 + c] = D[c][i * 3] * a[0] + D[c][i * 3 + 1] * a[1] + D[c][i * 3 + 2] * a[2];
}

/* calculate_Hi_inverse_depth_my_version.m:27-192 / calculate_Hi_cartesian_my_version.m:27-169.
 * Output: compact Hc (2x7: d h / d r_wc, d h / d q_wc; the 6 velocity columns are zero, :89)
 * and Hl (2x6; for a Cartesian landmark only the first 3 columns are used). zi = the landmark's
 * stored h (possibly stale, quirk Q7/Q8): the distortion Jacobian is evaluated there. */
static void Hi_landmark(int type, const double *xv, const double *y, const orc_cam *cam, const double zi[2], double Hc[14], double Hl[12])
{
    double Rwc[9], Rrw[9];
    q2r(xv + 3, Rwc);
    inv3(Rwc, Rrw);                                         /* Rrw = inv(q2r(q)), :73,171 */
    /* dhd_dhu = inv(inv(jacob_distor(cam, zi))), :152-153 + jacob_undistor_fm_my_version.m:38 */
    double Jd[4], Ju[4], dhd_dhu[4];
    jacob_distor(cam, zi, Jd);
    inv2(Jd, Ju);
    inv2(Ju, dhd_dhu);
    double hc[3], a_vec[3];
    double f = cam->f;
    if (type == ORC_INVDEPTH) {
        double theta = y[3], phi = y[4], rho = y[5];
        double mi[3] = { cos(phi) * sin(theta), -sin(phi), cos(phi) * cos(theta) };
        for (int i = 0; i < 3; ++i) a_vec[i] = (y[i] - xv[i]) * rho + mi[i];
    } else {
        for (int i = 0; i < 3; ++i) a_vec[i] = y[i] - xv[i];
    }
    for (int i = 0; i < 3; ++i) hc[i] = Rrw[i * 3] * a_vec[0] + Rrw[i * 3 + 1] * a_vec[1] + Rrw[i * 3 + 2] * a_vec[2];
    /* dhu_dhrl, :182-183 */
    double dhu_dhrl[6] = { f / hc[2], 0, -hc[0] * f / (hc[2] * hc[2]),  0, f / hc[2], -hc[1] * f / (hc[2] * hc[2]) };
    double dh_dhrl[6];
    mm(dhd_dhu, dhu_dhrl, dh_dhrl, 2, 2, 3);
    /* dhrl_drw = -inv(R)*rho (:136) / -inv(R) (cartesian) */
    double dhrl_drw[9];
    double sc = (type == ORC_INVDEPTH) ? y[5] : 1.0;
    for (int i = 0; i < 9; ++i) dhrl_drw[i] = -(Rrw[i]) * sc;
    double H11[6];
    mm(dh_dhrl, dhrl_drw, H11, 2, 3, 3);
    /* dhrl_dqwr = dRq_times_a_by_dq(qconj(q), a)*diag(1,-1,-1,-1), :119 */
    double qc[4] = { xv[3], -xv[4], -xv[5], -xv[6] };
    double dRa[12];
    dRq_times_a_by_dq(qc, a_vec, dRa);
    for (int i = 0; i < 3; ++i) { dRa[i * 4 + 1] = dRa[i * 4 + 1] * -1; dRa[i * 4 + 2] = dRa[i * 4 + 2] * -1; dRa[i * 4 + 3] = dRa[i * 4 + 3] * -1; }
    double H12[8];
    mm(dh_dhrl, dRa, H12, 2, 3, 4);
    for (int r = 0; r < 2; ++r) {
        for (int j = 0; j < 3; ++j) Hc[r * 7 + j] = H11[r * 3 + j];
        for (int j = 0; j < 4; ++j) Hc[r * 7 + 3 + j] = H12[r * 4 + j];
    }
    memset(Hl, 0, sizeof(double) * 12);
    if (type == ORC_INVDEPTH) {
        /* dhrl_dy = [rho*Rrw, Rrw*dm/dtheta, Rrw*dm/dphi, Rrw*(y(1:3)-rw)], :66-81 */
        double theta = y[3], phi = y[4], lambda = y[5];
        double dth[3] = { cos(phi) * cos(theta), 0, -cos(phi) * sin(theta) };
        double dph[3] = { -sin(phi) * sin(theta), -cos(phi), -sin(phi) * cos(theta) };
        double d3[3] = { y[0] - xv[0], y[1] - xv[1], y[2] - xv[2] };
        double A[18];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) A[i * 6 + j] = lambda * Rrw[i * 3 + j];
            A[i * 6 + 3] = Rrw[i * 3] * dth[0] + Rrw[i * 3 + 1] * dth[1] + Rrw[i * 3 + 2] * dth[2];
            A[i * 6 + 4] = Rrw[i * 3] * dph[0] + Rrw[i * 3 + 1] * dph[1] + Rrw[i * 3 + 2] * dph[2];
            A[i * 6 + 5] = Rrw[i * 3] * d3[0] + Rrw[i * 3 + 1] * d3[1] + Rrw[i * 3 + 2] * d3[2];
        }
        mm(dh_dhrl, A, Hl, 2, 3, 6);
    } else {
        double T[6];
        mm(dh_dhrl, Rrw, T, 2, 3, 3);              /* dhrl_dy = inv(q2r(q)) */
        for (int r = 0; r < 2; ++r) for (int j = 0; j < 3; ++j) Hl[r * 6 + j] = T[r * 3 + j];
    }
}

/* calculate_derivatives.m:27-60: H only for landmarks whose h is non-empty */
ORC_API int orc_jacobian(int N, const int *lm_type, const int *lm_off, const double *x, const orc_cam *cam,
                         const double *h, const int *has_h, double *Hc, double *Hl)
{
    for (int i = 0; i < N; ++i)
        if (has_h[i]) Hi_landmark(lm_type[i], x, x + lm_off[i], cam, h + 2 * i, Hc + 14 * i, Hl + 12 * i);
    return 0;
}

/* dense H row pair of landmark i (2 x n) from the compact form */
static void expand_H(int n, int type, int off, const double *Hc, const double *Hl, double *Hd)
{
    memset(Hd, 0, sizeof(double) * 2 * n);
    int d = (type == ORC_INVDEPTH) ? 6 : 3;
    for (int r = 0; r < 2; ++r) {
        for (int j = 0; j < 7; ++j) Hd[r * n + j] = Hc[r * 7 + j];
        for (int j = 0; j < d; ++j) Hd[r * n + off + j] = Hl[r * 6 + j];
    }
}

/* H*P*H' for a stack of dense rows: S(r x r) = H(r x n) P(n x n) H' evaluated as (H*P)*H' */
static void HPHt(int n, int r, const double *H, const double *P, double *S, double *HP_out)
{
    double *HP = HP_out ? HP_out : (double *)malloc(sizeof(double) * (size_t)r * n);
    nzrows nz = nz_build(H, r, n);
#pragma omp parallel for schedule(static) if ((long)r * n > 100000L)
    for (int a = 0; a < r; ++a) {
        const double *Ha = H + (size_t)a * n;
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int q = nz.start[a]; q < nz.start[a + 1]; ++q) { const int t = nz.idx[q]; s += Ha[t] * P[(size_t)t * n + j]; }   /* sparse rows: zeros contribute exact 0 */
            HP[(size_t)a * n + j] = s;
        }
    }
    for (int a = 0; a < r; ++a)
        for (int b = 0; b < r; ++b) {
            double s = 0;
            const double *Hb = H + (size_t)b * n;
            for (int q = nz.start[b]; q < nz.start[b + 1]; ++q) { const int t = nz.idx[q]; s += HP[(size_t)a * n + t] * Hb[t]; }
            S[a * r + b] = s;
        }
    nz_free(nz);
    if (!HP_out) free(HP);
}

/* ------------------------------------------------------------------ a5: S_i and gates */

/* search_IC_matches.m:33-44: S_i = H_i*P*H_i' + R_i, R_i = eye(2) */
ORC_API int orc_innovation(int n, int N, const int *lm_type, const int *lm_off, const double *P,
                           const double *Hc, const double *Hl, const int *has_h, double *S)
{
    double *Hd = (double *)malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < N; ++i) {
        if (!has_h[i]) continue;
        expand_H(n, lm_type[i], lm_off[i], Hc + 14 * i, Hl + 12 * i, Hd);
        double Si[4];
        HPHt(n, 2, Hd, P, Si, NULL);
        S[4 * i + 0] = Si[0] + 1; S[4 * i + 1] = Si[1]; S[4 * i + 2] = Si[2]; S[4 * i + 3] = Si[3] + 1;
    }
    free(Hd);
    return 0;
}

/* matching_sift_based.m:119-134: candidate c pairs landmark list entry k1[c] (index into the
 * predicted-landmark list `pred_idx`) with measured pixel zc.  strict_reference=1 reproduces
 * quirk Q5 (S taken from pred_idx[c], the c-th predicted landmark, not from the matched one). */
ORC_API int orc_window_gate(int M, const int *pred_idx, int n_pred, const int *k1, const double *zc, const double *h,
                            const double *S, const int *has_S, int strict_reference, int *accept)
{
    for (int c = 0; c < M; ++c) {
        int lm = pred_idx[k1[c]];
        int slm = strict_reference ? pred_idx[c < n_pred ? c : n_pred - 1] : lm;
        double half = has_S[slm] ? ceil(3 * sqrt(S[4 * slm])) : 40;
        double dx = zc[2 * c] - h[2 * lm], dy = zc[2 * c + 1] - h[2 * lm + 1];
        double dist = sqrt(dx * dx + dy * dy);
        accept[c] = dist <= half;
    }
    return 0;
}

/* @ekf_filter/rescue_hi_inliers.m:35-46: for IC and not LI landmarks:
 * nu'*inv(H*P*H')*nu < chi2  (no +R, quirk Q6) */
ORC_API int orc_rescue(int n, int N, const int *lm_type, const int *lm_off, const double *P,
                       const double *Hc, const double *Hl, const double *h, const double *z,
                       const int *ic, const int *li, double chi2, int *hi, double *d2_out)
{
    double *Hd = (double *)malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < N; ++i) {
        if (!(ic[i] == 1 && li[i] == 0)) continue;
        expand_H(n, lm_type[i], lm_off[i], Hc + 14 * i, Hl + 12 * i, Hd);
        double Si[4], Sinv[4];
        HPHt(n, 2, Hd, P, Si, NULL);
        inv2(Si, Sinv);
        double nu[2] = { z[2 * i] - h[2 * i], z[2 * i + 1] - h[2 * i + 1] };
        /* nui'*inv(Si)*nui evaluated left to right */
        double t0 = nu[0] * Sinv[0] + nu[1] * Sinv[2], t1 = nu[0] * Sinv[1] + nu[1] * Sinv[3];
        double d2 = t0 * nu[0] + t1 * nu[1];
        if (d2_out) d2_out[i] = d2;
        hi[i] = d2 < chi2 ? 1 : 0;
    }
    free(Hd);
    return 0;
}

/* ------------------------------------------------------------------ a9: update */

/* update.m:27-56.  H dense r x n (row-major), R dense r x r or NULL for eye(r).
 * K_out (n x r) may be NULL.  r == 0 leaves (x,P) untouched (update.m:50-55). */
ORC_API int orc_update(int n, int r, const double *x, const double *P, const double *H, const double *R,
                       const double *z, const double *h, double *x_out, double *P_out, double *K_out)
{
    if (r == 0) {
        memcpy(x_out, x, sizeof(double) * n);
        memcpy(P_out, P, sizeof(double) * (size_t)n * n);
        return 0;
    }
    double *HP = (double *)malloc(sizeof(double) * (size_t)r * n);
    double *S = (double *)malloc(sizeof(double) * (size_t)r * r);
    double *Sinv = (double *)malloc(sizeof(double) * (size_t)r * r);
    double *K = (double *)malloc(sizeof(double) * (size_t)n * r);
    double *KS = (double *)malloc(sizeof(double) * (size_t)n * r);
    if (!HP || !S || !Sinv || !K || !KS) return -1;
    /* :32 S = full(H*P*H' + R) */
    HPHt(n, r, H, P, S, HP);
    for (int a = 0; a < r; ++a) for (int b = 0; b < r; ++b) S[a * r + b] += R ? R[a * r + b] : (a == b ? 1.0 : 0.0);
    /* :33 K = P*H'*inv(S).  (P*H')(i,a) = sum_t P(i,t) H(a,t) */
    if (inv_lu(S, Sinv, r)) return -2;
    double *PHt = (double *)malloc(sizeof(double) * (size_t)n * r);
    nzrows nz = nz_build(H, r, n);
#pragma omp parallel for schedule(static) if ((long)r * n > 100000L)
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < r; ++a) {
            double s = 0;
            const double *Ha = H + (size_t)a * n;
            for (int q = nz.start[a]; q < nz.start[a + 1]; ++q) { const int t = nz.idx[q]; s += P[(size_t)i * n + t] * Ha[t]; }
            PHt[(size_t)i * r + a] = s;
        }
    nz_free(nz);
    mm(PHt, Sinv, K, n, r, r);
    free(PHt);
    /* :36 x = x + K*(z-h) */
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int a = 0; a < r; ++a) s += K[(size_t)i * r + a] * (z[a] - h[a]);
        x_out[i] = x[i] + s;
    }
    /* :37 P = P - K*S*K' */
    mm(K, S, KS, n, r, r);
#pragma omp parallel for schedule(static) if ((long)n * n > 100000L)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int a = 0; a < r; ++a) s += KS[(size_t)i * r + a] * K[(size_t)j * r + a];
            P_out[(size_t)i * n + j] = P[(size_t)i * n + j] - s;
        }
    /* :38 P = 0.5*P + 0.5*P' */
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double a = 0.5 * P_out[(size_t)i * n + j] + 0.5 * P_out[(size_t)j * n + i];
            double b = 0.5 * P_out[(size_t)j * n + i] + 0.5 * P_out[(size_t)i * n + j];
            P_out[(size_t)i * n + j] = a; P_out[(size_t)j * n + i] = b;
        }
    /* :42-48 */
    double J[16];
    normjac(x_out + 3, J);
    apply_jnorm(P_out, n, J);
    double nq = sqrt(x_out[3] * x_out[3] + x_out[4] * x_out[4] + x_out[5] * x_out[5] + x_out[6] * x_out[6]);
    for (int i = 0; i < 4; ++i) x_out[3 + i] /= nq;
    if (K_out) memcpy(K_out, K, sizeof(double) * (size_t)n * r);
    free(HP); free(S); free(Sinv); free(K); free(KS);
    return 0;
}

/* stack the landmark rows selected by sel[] (landmark indices, in landmark order) as the
 * @ekf_filter/ekf_update_{li,hi}_inliers.m:45-58 / ekf_update_all.m:46-62 wrappers do, then update */
ORC_API int orc_update_landmarks(int n, int N, const int *lm_type, const int *lm_off, int nsel, const int *sel,
                                 const double *x, const double *P, const double *Hc, const double *Hl,
                                 const double *zl, const double *hl, double *x_out, double *P_out)
{
    (void)N;
    int r = 2 * nsel;
    double *H = (double *)malloc(sizeof(double) * (size_t)(r ? r : 1) * n);
    double *z = (double *)malloc(sizeof(double) * (r ? r : 1));
    double *h = (double *)malloc(sizeof(double) * (r ? r : 1));
    for (int s = 0; s < nsel; ++s) {
        int i = sel[s];
        expand_H(n, lm_type[i], lm_off[i], Hc + 14 * i, Hl + 12 * i, H + (size_t)2 * s * n);
        z[2 * s] = zl[2 * i]; z[2 * s + 1] = zl[2 * i + 1];
        h[2 * s] = hl[2 * i]; h[2 * s + 1] = hl[2 * i + 1];
    }
    int rc = orc_update(n, r, x, P, H, NULL, z, h, x_out, P_out, NULL);
    free(H); free(z); free(h);
    return rc;
}

/* ------------------------------------------------------------------ a6-a8: 1-point RANSAC */

/* compute_hypothesis_support_fast.m:33-110.  meas[m] = landmark indices with non-empty z in
 * landmark order (generate_state_vector_pattern.m:34-52); z = their pixels (2 per landmark).
 * mask[m] receives the inlier flags in that order. */
ORC_API int orc_support(int m, const int *meas, const int *lm_type, const int *lm_off, const double *xi,
                        const orc_cam *cam, const double *z, double threshold, int *mask, double *resid_out)
{
    double rot[9];
    q2r(xi + 3, rot);                      /* un-normalised quaternion, quirk Q4 */
    double *res = (double *)malloc(sizeof(double) * (m ? m : 1));
    double minres = INFINITY;
    int n_id = 0;
    for (int j = 0; j < m; ++j) {
        int i = meas[j];
        const double *y = xi + lm_off[i];
        double v[3], hc[3];
        if (lm_type[i] == ORC_INVDEPTH) {
            double mi[3];
            m_dir(y[3], y[4], mi);
            for (int c = 0; c < 3; ++c) v[c] = (y[c] - xi[c]) * y[5] + mi[c];   /* :44-55 */
        } else {
            for (int c = 0; c < 3; ++c) v[c] = y[c] - xi[c];                      /* :88-90 */
        }
        for (int c = 0; c < 3; ++c) hc[c] = rot[0 * 3 + c] * v[0] + rot[1 * 3 + c] * v[1] + rot[2 * 3 + c] * v[2];
        /* :57-66: h_image = f*h_norm + [u0;v0]; distort */
        double hn0 = hc[0] / hc[2], hn1 = hc[1] / hc[2];
        double ui = cam->f * hn0 + cam->Cx, vi = cam->f * hn1 + cam->Cy;
        double xu = (ui - cam->Cx) / cam->f, yu = (vi - cam->Cy) / cam->f;
        double ru = sqrt(xu * xu + yu * yu);
        double D = 1 + cam->k1 * (ru * ru) + cam->k2 * pow(ru, 4);
        double ud = xu * D * cam->f + cam->Cx, vd = yu * D * cam->f + cam->Cy;
        double n0 = z[2 * j] - ud, n1 = z[2 * j + 1] - vd;
        res[j] = sqrt(n0 * n0 + n1 * n1);
        if (lm_type[i] == ORC_INVDEPTH) { ++n_id; if (res[j] < minres) minres = res[j]; }
    }
    int support = 0;
    for (int j = 0; j < m; ++j) {
        int i = meas[j];
        /* :70 inverse depth: residual < min(residual)+threshold ; :109 cartesian: residual < threshold */
        int in = (lm_type[i] == ORC_INVDEPTH) ? (res[j] < (minres + threshold)) : (res[j] < threshold);
        mask[j] = in;
        support += in;
        if (resid_out) resid_out[j] = res[j];
    }
    (void)n_id;
    free(res);
    return support;
}

/* ransac_hypotheses.m:51-63: state-only k-landmark update.  sel[k] index the landmark table. */
ORC_API int orc_hypothesis_state(int n, int k, const int *sel, const int *lm_type, const int *lm_off,
                                 const double *x, const double *P, const double *Hc, const double *Hl,
                                 const double *zl, const double *hl, double *xi)
{
    int r = 2 * k;
    double *H = (double *)malloc(sizeof(double) * (size_t)r * n);
    double *S = (double *)malloc(sizeof(double) * r * r), *Sinv = (double *)malloc(sizeof(double) * r * r);
    double *PHt = (double *)malloc(sizeof(double) * (size_t)n * r);
    for (int s = 0; s < k; ++s) expand_H(n, lm_type[sel[s]], lm_off[sel[s]], Hc + 14 * sel[s], Hl + 12 * sel[s], H + (size_t)2 * s * n);
    HPHt(n, r, H, P, S, NULL);
    for (int a = 0; a < r; ++a) S[a * r + a] += 1.0;         /* kalman_R = blkdiag(R_j) = I */
    inv_lu(S, Sinv, r);
    nzrows nz = nz_build(H, r, n);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < r; ++a) {
            double s = 0;
            const double *Ha = H + (size_t)a * n;
            for (int q = nz.start[a]; q < nz.start[a + 1]; ++q) { const int t = nz.idx[q]; s += P[(size_t)i * n + t] * Ha[t]; }
            PHt[(size_t)i * r + a] = s;
        }
    nz_free(nz);
    for (int i = 0; i < n; ++i) {
        double acc = 0;
        for (int b = 0; b < r; ++b) {
            double kib = 0;
            for (int a = 0; a < r; ++a) kib += PHt[(size_t)i * r + a] * Sinv[a * r + b];
            int lm = sel[b / 2];
            acc += kib * (zl[2 * lm + (b & 1)] - hl[2 * lm + (b & 1)]);
        }
        xi[i] = x[i] + acc;
    }
    free(H); free(S); free(Sinv); free(PHt);
    return 0;
}

/* ransac_hypotheses.m:27-85 with the hypothesis draws as an INPUT (hyp[n_draw*k] = positions in
 * the IC list, replacing select_random_match.m's randperm).  ic_list[num_ic] = landmark indices
 * with individually_compatible==1; meas[m] = landmarks with non-empty z.
 * early_exit=1 replays the reference's termination rule (quirk Q1); 0 evaluates all n_draw.
 * Outputs: support[n_draw] (-1 where not evaluated), li_mask[m], best index, iterations run,
 * final n_hyp (StatData.RANSAC_ITER), max support. */
ORC_API int orc_ransac(int n, int N, const int *lm_type, const int *lm_off, const double *x, const double *P,
                       const double *Hc, const double *Hl, const double *zl, const double *hl,
                       int num_ic, const int *ic_list, int m, const int *meas, const orc_cam *cam,
                       int n_draw, int k, const int *hyp, double threshold, int early_exit,
                       int *support, int *li_mask, int *best_out, int *iters_out, int *n_hyp_out, int *max_support_out)
{
    (void)N;
    double *xi = (double *)malloc(sizeof(double) * n);
    double *zm = (double *)malloc(sizeof(double) * 2 * (m ? m : 1));
    int *mask = (int *)malloc(sizeof(int) * (m ? m : 1));
    for (int j = 0; j < m; ++j) { zm[2 * j] = zl[2 * meas[j]]; zm[2 * j + 1] = zl[2 * meas[j] + 1]; }
    for (int j = 0; j < m; ++j) li_mask[j] = 0;
    for (int i = 0; i < n_draw; ++i) support[i] = -1;
    int n_hyp = 1000, max_support = 0, best = -1, iters = 0;
    int limit = early_exit ? (n_draw < 1000 ? n_draw : 1000) : n_draw;
    for (int it = 0; it < limit; ++it) {
        if (early_exit && n_hyp == 0) break;                                   /* :41-46 */
        int sel[8];
        for (int s = 0; s < k; ++s) sel[s] = ic_list[hyp[it * k + s]];
        orc_hypothesis_state(n, k, sel, lm_type, lm_off, x, P, Hc, Hl, zl, hl, xi);
        int sup = orc_support(m, meas, lm_type, lm_off, xi, cam, zm, threshold, mask, NULL);
        support[it] = sup;
        ++iters;
        if (sup > max_support) {                                                /* :74-79 */
            max_support = sup; best = it;
            memcpy(li_mask, mask, sizeof(int) * m);
            double epsilon = 1 - ((double)sup / (double)num_ic);
            n_hyp = (int)ceil(log(1 - 0.99) / log(1 - (1 - epsilon)));
        }
        if (early_exit && n_hyp <= k) break;                                    /* :80, `i` is the inner loop's k */
    }
    *best_out = best; *iters_out = iters; *n_hyp_out = n_hyp; *max_support_out = max_support;
    free(xi); free(zm); free(mask);
    return 0;
}

/* ------------------------------------------------------------------ a10: siftmatch */

/* sift/siftmatch.c:83-132 for each class.  L1: ND x K1, L2: ND x K2 column-major (one
 * descriptor per column).  pairs[2*M] receives 1-based (k1,k2) as the gateway writes them
 * (:241-242), score[M] the best squared distance.  Returns M. */
#define ORC_SIFTMATCH(NAME, T, ACC, MAXV)                                                          \
    ORC_API int NAME(int ND, int K1, const T *L1, int K2, const T *L2, double thresh_d,             \
                     double *pairs, double *score)                                                  \
    {                                                                                               \
        float thresh = (float)thresh_d; /* the gateway passes a double into a float parameter */    \
        int M = 0;                                                                                  \
        for (int k1 = 0; k1 < K1; ++k1) {                                                           \
            const T *a = L1 + (size_t)k1 * ND;                                                      \
            ACC best = MAXV, second = MAXV;                                                         \
            int bestk = -1;                                                                         \
            for (int k2 = 0; k2 < K2; ++k2) {                                                       \
                const T *b = L2 + (size_t)k2 * ND;                                                  \
                ACC acc = 0;                                                                        \
                for (int bin = 0; bin < ND; ++bin) {                                                \
                    ACC delta = ((ACC)a[bin]) - ((ACC)b[bin]);                                      \
                    acc += delta * delta;                                                           \
                }                                                                                   \
                if (acc < best) { second = best; best = acc; bestk = k2; }                          \
                else if (acc < second) { second = acc; }                                            \
            }                                                                                       \
            if (thresh * (float)best <= (float)second && bestk != -1) {                             \
                pairs[2 * M] = k1 + 1; pairs[2 * M + 1] = bestk + 1;                                \
                if (score) score[M] = (double)best;                                                 \
                ++M;                                                                                \
            }                                                                                       \
        }                                                                                           \
        return M;                                                                                   \
    }

ORC_SIFTMATCH(orc_siftmatch_f64, double, double, INFINITY)
ORC_SIFTMATCH(orc_siftmatch_f32, float, float, INFINITY)
ORC_SIFTMATCH(orc_siftmatch_i8, signed char, int, 0x7fffffff)
ORC_SIFTMATCH(orc_siftmatch_u8, unsigned char, int, 0x7fffffff)

/* ------------------------------------------------------------------ a11: kNearestNeighbors */

typedef struct { double d; int idx; } orc_kv;
static int kv_cmp(const void *a, const void *b)
{
    const orc_kv *x = (const orc_kv *)a, *y = (const orc_kv *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->idx - y->idx;          /* MATLAB sort is stable: lowest index first on ties */
}

/* kNearestNeighbors.m:29-39.  data N x D, query M x D, row-major here (one vector per row).
 * ids (M x k, 1-based), dist (M x k, Euclidean). */
ORC_API int orc_knn(int D, int N, const double *data, int M, const double *query, int k, double *ids, double *dist)
{
    if (k > N) return -1;
    orc_kv *kv = (orc_kv *)malloc(sizeof(orc_kv) * N);
    for (int i = 0; i < M; ++i) {
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int d = 0; d < D; ++d) { double t = query[(size_t)i * D + d] - data[(size_t)j * D + d]; s += t * t; }
            kv[j].d = s; kv[j].idx = j;
        }
        qsort(kv, N, sizeof(orc_kv), kv_cmp);
        for (int c = 0; c < k; ++c) { ids[(size_t)i * k + c] = kv[c].idx + 1; dist[(size_t)i * k + c] = sqrt(kv[c].d); }
    }
    free(kv);
    return 0;
}

/* ------------------------------------------------------------------ map management
 * (map_management.m:27-79).  UNPINNED: no reference artefact holds a before/after pair of these
 * operations; checked against the numpy twin and by round-trip / symmetry properties only. */

/* undistort_fm_my_version.m:27-48 (10 Newton steps) */
static void undistort_fm(const orc_cam *cam, const double uvd[2], double uv[2])
{
    double xd = (uvd[0] - cam->Cx) / cam->f, yd = (uvd[1] - cam->Cy) / cam->f;
    double rd = sqrt(xd * xd + yd * yd);
    double ru = rd / (1 + cam->k1 * rd * rd + cam->k2 * pow(rd, 4));
    for (int k = 0; k < 10; ++k) {
        double f1 = ru + cam->k1 * pow(ru, 3) + cam->k2 * pow(ru, 5) - rd;
        double f1p = 1 + 3 * cam->k1 * ru * ru + 5 * cam->k2 * pow(ru, 4);
        ru = ru - f1 / f1p;
    }
    double D = 1 + cam->k1 * ru * ru + cam->k2 * pow(ru, 4);
    uv[0] = cam->f * xd / D + cam->Cx;
    uv[1] = cam->f * yd / D + cam->Cy;
}

/* delete_a_feature.m:47-51 applied to the landmarks del[n_del] (ascending; delete_features.m:54-74 deletes
 * from the back so that indices stay valid).  Returns the new state size. */
ORC_API int orc_map_delete(int n, int N, const int *lm_type, const int *lm_off, int n_del, const int *del,
                           const double *x, const double *P, double *x_out, double *P_out)
{
    int *keep = (int *)malloc(sizeof(int) * n);
    int nn = 0;
    for (int i = 0; i < 13; ++i) keep[nn++] = i;
    int d = 0;
    for (int i = 0; i < N; ++i) {
        int dim = lm_type[i] == ORC_INVDEPTH ? 6 : 3;
        if (d < n_del && del[d] == i) { ++d; continue; }
        for (int c = 0; c < dim; ++c) keep[nn++] = lm_off[i] + c;
    }
    for (int a = 0; a < nn; ++a) {
        x_out[a] = x[keep[a]];
        for (int b = 0; b < nn; ++b) P_out[(size_t)a * nn + b] = P[(size_t)keep[a] * n + keep[b]];
    }
    free(keep);
    return nn;
}

/* add_features_inverse_depth.m:27-47 -> hinv_my_version.m:26-53 + add_a_feature_covariance_inverse_depth.m:27-90,
 * one feature appended after the other.  x_out/P_out hold n + 6*n_new entries per dimension. */
ORC_API int orc_map_add(int n, int n_new, const double *uvd, double std_pxl, const double *initial_rho, const orc_cam *cam,
                        const double *x, const double *P, double *x_out, double *P_out)
{
    int cur = n;
    int nmax = n + 6 * n_new;
    double *Pc = (double *)calloc((size_t)nmax * nmax, sizeof(double));
    for (int i = 0; i < n; ++i) { x_out[i] = x[i]; for (int j = 0; j < n; ++j) Pc[(size_t)i * nmax + j] = P[(size_t)i * n + j]; }
    double fku = cam->f, fkv = cam->f, U0 = cam->Cx, V0 = cam->Cy;      /* cam.K = [f 0 Cx; 0 f Cy; 0 0 1] */
    for (int f = 0; f < n_new; ++f) {
        const double *pix = uvd + 2 * f;
        double uvu[2];
        undistort_fm(cam, pix, uvu);
        double Rwc[9];
        q2r(x + 3, Rwc);
        double hc[3] = { -(U0 - uvu[0]) / fku, -(V0 - uvu[1]) / fkv, 1.0 };
        double nw[3];
        for (int i = 0; i < 3; ++i) nw[i] = Rwc[i * 3] * hc[0] + Rwc[i * 3 + 1] * hc[1] + Rwc[i * 3 + 2] * hc[2];
        /* hinv_my_version.m:51 */
        double ynew[6] = { x[0], x[1], x[2], atan2(nw[0], nw[2]), atan2(-nw[1], sqrt(nw[0] * nw[0] + nw[2] * nw[2])), initial_rho[f] };
        for (int i = 0; i < 6; ++i) x_out[cur + i] = ynew[i];
        double std_rho = initial_rho[f] * initial_rho[f] * 0.01;            /* add_features_inverse_depth.m:41 overrides its argument */
        double Xw = nw[0], Yw = nw[1], Zw = nw[2];
        double dth[3] = { Zw / (Xw * Xw + Zw * Zw), 0, -Xw / (Xw * Xw + Zw * Zw) };
        double s2 = Xw * Xw + Yw * Yw + Zw * Zw, sxz = sqrt(Xw * Xw + Zw * Zw);
        double dph[3] = { (Xw * Yw) / (s2 * sxz), -sxz / s2, (Zw * Yw) / (s2 * sxz) };
        double dg_dq[12];
        dRq_times_a_by_dq(x + 3, hc, dg_dq);
        double dth_dq[4], dph_dq[4];
        for (int c = 0; c < 4; ++c) {
            dth_dq[c] = dth[0] * dg_dq[0 * 4 + c] + dth[1] * dg_dq[1 * 4 + c] + dth[2] * dg_dq[2 * 4 + c];
            dph_dq[c] = dph[0] * dg_dq[0 * 4 + c] + dph[1] * dg_dq[1 * 4 + c] + dph[2] * dg_dq[2 * 4 + c];
        }
        double J[6 * 13] = { 0 };                                          /* dy_dxv */
        J[0 * 13 + 0] = J[1 * 13 + 1] = J[2 * 13 + 2] = 1;
        for (int c = 0; c < 4; ++c) { J[3 * 13 + 3 + c] = dth_dq[c]; J[4 * 13 + 3 + c] = dph_dq[c]; }
        /* dy_dhd (6x3) = [dyprima_dgw*dgw_dgc*dgc_dhu*dhu_dhd  0; 0 0 1] */
        double Jd[4], dhu_dhd[4];
        jacob_distor(cam, pix, Jd);
        inv2(Jd, dhu_dhd);                                                  /* jacob_undistor_fm_my_version.m:38 */
        double dgc_dhu[6] = { 1 / fku, 0,  0, 1 / fkv,  0, 0 };             /* 3x2 */
        double dyp_dgw[15] = { 0 };                                         /* 5x3: rows 3,4 = dtheta_dgw, dphi_dgw */
        for (int c = 0; c < 3; ++c) { dyp_dgw[3 * 3 + c] = dth[c]; dyp_dgw[4 * 3 + c] = dph[c]; }
        double T1[15], T2[10], T3[10];
        mm(dyp_dgw, Rwc, T1, 5, 3, 3);
        mm(T1, dgc_dhu, T2, 5, 3, 2);
        mm(T2, dhu_dhd, T3, 5, 2, 2);
        double dy_dhd[18] = { 0 };
        for (int r_ = 0; r_ < 5; ++r_) { dy_dhd[r_ * 3] = T3[r_ * 2]; dy_dhd[r_ * 3 + 1] = T3[r_ * 2 + 1]; }
        dy_dhd[5 * 3 + 2] = 1;
        double Padd[9] = { std_pxl * std_pxl, 0, 0,  0, std_pxl * std_pxl, 0,  0, 0, std_rho * std_rho };
        double T4[18], Nn[36];
        mm(dy_dhd, Padd, T4, 6, 3, 3);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double s = 0; for (int t = 0; t < 3; ++t) s += T4[i * 3 + t] * dy_dhd[j * 3 + t]; Nn[i * 6 + j] = s; }
        /* new rows: dy_dxv*[P_xv P_xvy] ; new cols: [P_xv; P_yxv]*dy_dxv' ; corner: dy_dxv*P_xv*dy_dxv' + Nn */
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < cur; ++j) {
                double s = 0, c2 = 0;
                for (int t = 0; t < 13; ++t) { s += J[i * 13 + t] * Pc[(size_t)t * nmax + j]; c2 += Pc[(size_t)j * nmax + t] * J[i * 13 + t]; }
                Pc[(size_t)(cur + i) * nmax + j] = s;
                Pc[(size_t)j * nmax + cur + i] = c2;
            }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0;
                for (int t = 0; t < 13; ++t) s += Pc[(size_t)(cur + i) * nmax + t] * J[j * 13 + t];       /* (dy_dxv*P_xv)*dy_dxv' */
                Pc[(size_t)(cur + i) * nmax + cur + j] = s + Nn[i * 6 + j];
            }
        cur += 6;
    }
    for (int i = 0; i < cur; ++i) for (int j = 0; j < cur; ++j) P_out[(size_t)i * cur + j] = Pc[(size_t)i * nmax + j];
    free(Pc);
    return cur;
}

/* inversedepth_2_cartesian.m:27-76.  converted[N] receives the flags; lm_type_out the new types.  Landmarks are
 * processed in order on the evolving (X, P) exactly as the reference does.  Returns the new state size. */
ORC_API int orc_map_convert(int n, int N, const int *lm_type, double threshold, const double *x, const double *P,
                            double *x_out, double *P_out, int *lm_type_out, int *converted)
{
    int cur = n;
    double *X = (double *)malloc(sizeof(double) * n), *Pc = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *Pn = (double *)malloc(sizeof(double) * (size_t)n * n), *T = (double *)malloc(sizeof(double) * (size_t)n * n);
    memcpy(X, x, sizeof(double) * n);
    memcpy(Pc, P, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < N; ++i) { lm_type_out[i] = lm_type[i]; converted[i] = 0; }
    for (int i = 0; i < N; ++i) {
        if (lm_type_out[i] != ORC_INVDEPTH) continue;
        int o = 13;
        for (int j = 0; j < i; ++j) o += lm_type_out[j] == ORC_INVDEPTH ? 6 : 3;
        double std_rho = sqrt(Pc[(size_t)(o + 5) * cur + o + 5]);
        double rho = X[o + 5], std_d = std_rho / (rho * rho), theta = X[o + 3], phi = X[o + 4];
        double mi[3];
        m_dir(theta, phi, mi);
        double p[3] = { X[o] + (1 / rho) * mi[0], X[o + 1] + (1 / rho) * mi[1], X[o + 2] + (1 / rho) * mi[2] };   /* inversedepth2cartesian.m */
        double a[3] = { p[0] - X[o], p[1] - X[o + 1], p[2] - X[o + 2] }, c2[3] = { p[0] - X[0], p[1] - X[1], p[2] - X[2] };
        double d_c2p = sqrt(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
        double cos_alpha = (a[0] * c2[0] + a[1] * c2[1] + a[2] * c2[2]) / (sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * d_c2p);
        double li = 4 * std_d * cos_alpha / d_c2p;
        if (!(li < threshold)) continue;
        double dmt[3] = { cos(phi) * cos(theta), 0, -cos(phi) * sin(theta) };
        double dmp[3] = { -sin(phi) * sin(theta), -cos(phi), -sin(phi) * cos(theta) };
        double J[18];
        for (int r_ = 0; r_ < 3; ++r_) {
            for (int c = 0; c < 3; ++c) J[r_ * 6 + c] = r_ == c ? 1 : 0;
            J[r_ * 6 + 3] = (1 / rho) * dmt[r_]; J[r_ * 6 + 4] = (1 / rho) * dmp[r_]; J[r_ * 6 + 5] = -mi[r_] / (rho * rho);
        }
        int nn = cur - 3;
        /* T = J_all * P (nn x cur), Pn = T * J_all' (nn x nn) */
        for (int a2 = 0; a2 < nn; ++a2)
            for (int j = 0; j < cur; ++j) {
                double s;
                if (a2 < o) s = Pc[(size_t)a2 * cur + j];
                else if (a2 < o + 3) { s = 0; for (int t = 0; t < 6; ++t) s += J[(a2 - o) * 6 + t] * Pc[(size_t)(o + t) * cur + j]; }
                else s = Pc[(size_t)(a2 + 3) * cur + j];
                T[(size_t)a2 * cur + j] = s;
            }
        for (int a2 = 0; a2 < nn; ++a2)
            for (int b2 = 0; b2 < nn; ++b2) {
                double s;
                if (b2 < o) s = T[(size_t)a2 * cur + b2];
                else if (b2 < o + 3) { s = 0; for (int t = 0; t < 6; ++t) s += T[(size_t)a2 * cur + o + t] * J[(b2 - o) * 6 + t]; }
                else s = T[(size_t)a2 * cur + b2 + 3];
                Pn[(size_t)a2 * nn + b2] = s;
            }
        for (int t = 0; t < 3; ++t) X[o + t] = p[t];
        memmove(X + o + 3, X + o + 6, sizeof(double) * (cur - o - 6));
        memcpy(Pc, Pn, sizeof(double) * (size_t)nn * nn);
        cur = nn;
        lm_type_out[i] = ORC_CARTESIAN;
        converted[i] = 1;
    }
    memcpy(x_out, X, sizeof(double) * cur);
    memcpy(P_out, Pc, sizeof(double) * (size_t)cur * cur);
    free(X); free(Pc); free(Pn); free(T);
    return cur;
}

/* =====================================================================================================================
 * the visual-odometry front end's 4-point 3D-3D RANSAC.  PARITY UNPINNED against reference outputs: the
 * reference ships no VO golden data (no d1_*.dat range files, no RANSAC_RESULT_*.mat).  The restatement is pinned to
 * LAPACK's svd (what MATLAB calls) through the numpy twin and to rigid-motion known answers (tests/test_vo_oracle.py).
 * ===================================================================================================================== */

/* svd(H) for a 3x3 by one-sided Jacobi (Hestenes): H = U diag(sv) V'.  Replaces MATLAB's `[U,S,V] = svd(H)`
 * (find_transform_matrix_dr_ye.m:19); only V*U' and the singular values are used downstream, and those do not depend
 * on the ordering / sign conventions of the factorisation when H has full rank. */
static void orc_svd3(const double H[9], double U[9], double sv[3], double V[9])
{
    double A[9];
    memcpy(A, H, sizeof A);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) { alpha += A[3 * i + p] * A[3 * i + p]; beta += A[3 * i + q] * A[3 * i + q]; gamma += A[3 * i + p] * A[3 * i + q]; }
                if (gamma == 0.0 || fabs(gamma) <= 2.2e-16 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    double ap = A[3 * i + p], aq = A[3 * i + q];
                    A[3 * i + p] = c * ap - s * aq; A[3 * i + q] = s * ap + c * aq;
                    double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = c * vp - s * vq; V[3 * i + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int ok[3];
    double big = 0;
    for (int j = 0; j < 3; ++j) { sv[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]); if (sv[j] > big) big = sv[j]; }
    for (int j = 0; j < 3; ++j) {
        ok[j] = sv[j] > 1e-300 && sv[j] > 1e-18 * big;
        if (ok[j]) for (int i = 0; i < 3; ++i) U[3 * i + j] = A[3 * i + j] / sv[j];
    }
    /* complete U for (numerically) zero singular values: any orthonormal completion, as LAPACK's is */
    int nok = ok[0] + ok[1] + ok[2];
    if (nok == 2) {
        int j = !ok[0] ? 0 : (!ok[1] ? 1 : 2), a = (j + 1) % 3, b = (j + 2) % 3;
        U[j] = U[3 + a] * U[6 + b] - U[6 + a] * U[3 + b];
        U[3 + j] = U[6 + a] * U[b] - U[a] * U[6 + b];
        U[6 + j] = U[a] * U[3 + b] - U[3 + a] * U[b];
    } else if (nok < 2) {
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;      /* rank <= 1: the caller's state logic rejects it */
        if (nok == 1) {
            int j = ok[0] ? 0 : (ok[1] ? 1 : 2);
            double u[3] = { A[j] / sv[j], A[3 + j] / sv[j], A[6 + j] / sv[j] };
            int m = fabs(u[0]) < fabs(u[1]) ? (fabs(u[0]) < fabs(u[2]) ? 0 : 2) : (fabs(u[1]) < fabs(u[2]) ? 1 : 2);
            double e[3] = { 0, 0, 0 }; e[m] = 1;
            double w[3] = { u[1] * e[2] - u[2] * e[1], u[2] * e[0] - u[0] * e[2], u[0] * e[1] - u[1] * e[0] };
            double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            for (int i = 0; i < 3; ++i) w[i] /= nw;
            double x[3] = { u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0] };
            int a = (j + 1) % 3, b = (j + 2) % 3;
            for (int i = 0; i < 3; ++i) { U[3 * i + j] = u[i]; U[3 * i + a] = w[i]; U[3 * i + b] = x[i]; }
        }
    }
}

static double orc_det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* find_transform_matrix_dr_ye.m:8-41.  pset1, pset2: 3 x pnum column-major; idx (may be NULL) selects columns.
 * rot row-major 3x3; returns state (1, 2, -1, 0). */
ORC_API int orc_vo_find_transform(int pnum, const int *idx, const double *pset1, const double *pset2, double *rot, double *trans)
{
    double ct1[3] = { 0, 0, 0 }, ct2[3] = { 0, 0, 0 }, H[9] = { 0 };
    for (int k = 0; k < pnum; ++k) {
        int c = idx ? idx[k] : k;
        for (int i = 0; i < 3; ++i) { ct1[i] += pset1[3 * c + i]; ct2[i] += pset2[3 * c + i]; }
    }
    for (int i = 0; i < 3; ++i) { ct1[i] /= pnum; ct2[i] /= pnum; }                       /* :12 */
    for (int k = 0; k < pnum; ++k) {                                                      /* :13-16  H += q2*q1' */
        int c = idx ? idx[k] : k;
        double q1[3], q2[3];
        for (int i = 0; i < 3; ++i) { q1[i] = pset1[3 * c + i] - ct1[i]; q2[i] = pset2[3 * c + i] - ct2[i]; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[3 * i + j] += q2[i] * q1[j];
    }
    double U[9], sv[3], V[9], Xq[9];
    orc_svd3(H, U, sv, V);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Xq[3 * i + j] = V[3 * i] * U[3 * j] + V[3 * i + 1] * U[3 * j + 1] + V[3 * i + 2] * U[3 * j + 2];   /* :20 */
    double mdet = orc_det3(Xq);
    int state;
    if (round(mdet) == 1) state = 1;                                                       /* :23-26 */
    else if (round(mdet) == -1) {                                                          /* :27-39 */
        int zn = -1, cnt = 0;
        for (int j = 0; j < 3; ++j) if (fabs(sv[j]) < 0.00000000000001) { zn = j; ++cnt; }
        if (cnt == 1) {
            for (int i = 0; i < 3; ++i) V[3 * i + zn] = -V[3 * i + zn];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Xq[3 * i + j] = V[3 * i] * U[3 * j] + V[3 * i + 1] * U[3 * j + 1] + V[3 * i + 2] * U[3 * j + 2];
            state = 2;
        } else state = -1;
    } else state = 0;                                                                      /* :40-44 */
    if (state >= 1) {
        memcpy(rot, Xq, sizeof Xq);
        for (int i = 0; i < 3; ++i) trans[i] = ct1[i] - (rot[3 * i] * ct2[0] + rot[3 * i + 1] * ct2[1] + rot[3 * i + 2] * ct2[2]);
    } else {
        memcpy(rot, H, sizeof H);                                                          /* rot = H, trans = 0 */
        trans[0] = trans[1] = trans[2] = 0;
    }
    return state;
}

/* ransac_dr_ye.m:13-19: pset(:,i) = [-x(ROW,COL); -y(ROW,COL); z(ROW,COL)], ROW = round(pix(2)), COL = round(pix(1)).
 * frm: ldf x K column-major (rows 1:2 = pixel), sel[pnum]: 1-based keypoint numbers (one row of `match`);
 * x, y, z: rows x cols column-major.  Returns 0, or -1 when a pixel falls outside the image (MATLAB would error). */
ORC_API int orc_vo_gather(int rows, int cols, const double *x, const double *y, const double *z, int ldf, const double *frm,
                          int pnum, const double *sel, double *pset)
{
    for (int i = 0; i < pnum; ++i) {
        int k = (int)sel[i] - 1;
        int COL = (int)round(frm[(size_t)ldf * k]), ROW = (int)round(frm[(size_t)ldf * k + 1]);
        if (ROW < 1 || ROW > rows || COL < 1 || COL > cols) return -1;
        size_t o = (size_t)(COL - 1) * rows + (ROW - 1);
        pset[3 * i] = -x[o]; pset[3 * i + 1] = -y[o]; pset[3 * i + 2] = z[o];
    }
    return 0;
}

/* ransac_dr_ye.m:20-23: the inlier radius scale.  Returns -1 when no point is farther than 0.4 m (MATLAB errors there). */
ORC_API int orc_vo_dist(int pnum, const double *pset2, double *dist)
{
    double minZ = 0; int have = 0;
    for (int k = 0; k < pnum; ++k) {
        double nr = sqrt(pset2[3 * k + 2] * pset2[3 * k + 2] + pset2[3 * k + 1] * pset2[3 * k + 1] + pset2[3 * k] * pset2[3 * k]);
        if (nr > 0.4 && (!have || pset2[3 * k + 2] < minZ)) { minZ = pset2[3 * k + 2]; have = 1; }
    }
    if (!have) return -1;
    for (int k = 0; k < pnum; ++k)
        if (pset2[3 * k + 2] == minZ) {                                                    /* pmZ(1): over ALL points, not only the far ones */
            *dist = sqrt(pset2[3 * k] * pset2[3 * k] + pset2[3 * k + 1] * pset2[3 * k + 1] + pset2[3 * k + 2] * pset2[3 * k + 2]);
            return 0;
        }
    return -1;
}

/* One hypothesis: ransac_dr_ye.m:48-72 with the 4 sample positions given (0-based).  inl[pnum] flags; returns cnum. */
ORC_API int orc_vo_hypothesis(int pnum, const double *pset1, const double *pset2, const int *draw, double dist, int *inl, int *state_out)
{
    double rot[9], tr[3];
    int st = orc_vo_find_transform(4, draw, pset1, pset2, rot, tr);
    if (state_out) *state_out = st;
    int cnum = 0;
    for (int k = 0; k < pnum; ++k) {
        double d = 0;
        for (int i = 0; i < 3; ++i) {
            double v = rot[3 * i] * pset2[3 * k] + rot[3 * i + 1] * pset2[3 * k + 1] + rot[3 * i + 2] * pset2[3 * k + 2];
            v = v + tr[i];
            d = d + (v - pset1[3 * k + i]) * (v - pset1[3 * k + i]);
        }
        inl[k] = d < 0.001 * dist;                                                        /* :67 */
        cnum += inl[k];
    }
    return cnum;
}

/* vodometry_dr_ye.m:162-236.  draws: n_hyp x 4 (0-based).  out[16]: rot(9, row-major) trans(3) euler(3) errmean;
 * out2: [errstd, dist]; iout[6]: sta, n_support, n_iterations, best, all-evaluated count, reserved.
 * cnum_out[n_hyp]; inl_out[pnum] = the winner's inliers.  Returns 0, -1 (dist undefined), -2 (pnum < 4). */
ORC_API int orc_vo_ransac(int pnum, const double *pset1, const double *pset2, int n_hyp, const int *draws, int *cnum_out, int *inl_out,
                          double *out, double *out2, int *iout)
{
    if (pnum < 4) return -2;
    double dist;
    if (orc_vo_dist(pnum, pset2, &dist)) return -1;
    int *inl = (int *)malloc(sizeof(int) * pnum);
    int maxC = 0, best = -1, bestc = -1;
    double nIter = n_hyp;                                                                  /* nIterations = rst (:174) */
    for (int i = 0; i < n_hyp; ++i) {                                                      /* the for-range is fixed at entry: every draw is evaluated */
        int c = orc_vo_hypothesis(pnum, pset1, pset2, draws + 4 * i, dist, inl, NULL);
        cnum_out[i] = c;
        if (c > maxC) {                                                                    /* :185-188 */
            maxC = c;
            nIter = 5 * ceil(log(0.01) / log(1 - pow((double)maxC / pnum, 4)));
        }
        if (c > bestc) { bestc = c; best = i; }                                            /* [rs_max, rs_ind] = max(tmp_cnum): first maximum */
    }
    iout[2] = (int)(nIter < n_hyp ? nIter : n_hyp); iout[3] = best; iout[4] = n_hyp; iout[5] = 0;
    out2[1] = dist;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    out2[0] = 0;
    if (bestc < 3) { iout[0] = 4; iout[1] = bestc < 0 ? 0 : bestc; memset(inl_out, 0, sizeof(int) * pnum); free(inl); return 0; }   /* :198-205 */
    orc_vo_hypothesis(pnum, pset1, pset2, draws + 4 * best, dist, inl_out, NULL);
    int *idx = (int *)malloc(sizeof(int) * pnum), m = 0;
    for (int k = 0; k < pnum; ++k) if (inl_out[k]) idx[m++] = k;
    double *rot = out, *tr = out + 9;
    int sta = orc_vo_find_transform(m, idx, pset1, pset2, rot, tr);                        /* :221 */
    double sum = 0, *en = (double *)malloc(sizeof(double) * m);
    for (int a = 0; a < m; ++a) {                                                          /* :222-225 */
        int k = idx[a]; double s2 = 0;
        for (int i = 0; i < 3; ++i) {
            double v = rot[3 * i] * pset2[3 * k] + rot[3 * i + 1] * pset2[3 * k + 1] + rot[3 * i + 2] * pset2[3 * k + 2] + tr[i] - pset1[3 * k + i];
            s2 += v * v;
        }
        en[a] = sqrt(s2); sum += en[a];
    }
    double mean = sum / m, var = 0;
    for (int a = 0; a < m; ++a) var += (en[a] - mean) * (en[a] - mean);
    out[15] = mean; out2[0] = m > 1 ? sqrt(var / (m - 1)) : 0;
    if (sta >= 1) {                                                                        /* R2e.m:21-23 */
        out[12] = atan2(rot[7], rot[8]); out[13] = asin(-rot[6]); out[14] = atan2(rot[3], rot[0]);
    }
    iout[0] = sta; iout[1] = m;
    free(en); free(idx); free(inl);
    return 0;
}

/* R2q.m (slamToolbox): q = [a -b -c -d]'.  R row-major. */
ORC_API void orc_R2q(const double *R, double *q)
{
    double T = R[0] + R[4] + R[8] + 1, a, b, c, d, S;
    if (T > 0.00000001) {
        S = 2 * sqrt(T); a = 0.25 * S; b = (R[5] - R[7]) / S; c = (R[6] - R[2]) / S; d = (R[1] - R[3]) / S;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        S = 2 * sqrt(1.0 + R[0] - R[4] - R[8]); a = (R[5] - R[7]) / S; b = 0.25 * S; c = (R[1] + R[3]) / S; d = (R[6] + R[2]) / S;
    } else if (R[4] > R[8]) {
        S = 2 * sqrt(1.0 + R[4] - R[0] - R[8]); a = (R[6] - R[2]) / S; b = (R[1] + R[3]) / S; c = 0.25 * S; d = (R[5] + R[7]) / S;
    } else {
        S = 2 * sqrt(1.0 + R[8] - R[0] - R[4]); a = (R[1] - R[3]) / S; b = (R[6] + R[2]) / S; c = (R[5] + R[7]) / S; d = 0.25 * S;
    }
    q[0] = a; q[1] = -b; q[2] = -c; q[3] = -d;
}
