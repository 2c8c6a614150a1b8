// pre3_map.hip -- SURVEY 8(f)-1: map management on the device (map_management.m:27-79), so that P never leaves HBM.
//
// All three operations of the reference are congruences  P <- A P A' (+ D)  with a very sparse A:
//   delete_a_feature.m:47-51                       A = row selection
//   add_a_feature_covariance_inverse_depth.m:83-90  A = [I ; dy_dxv], D = dy_dhd Padd dy_dhd' on the new 6x6 block
//   inversedepth_2_cartesian.m:58-72                A = blkdiag(I, J(3x6), I)
// A is built row by row on the device (k_map_fill: <= 6 non-zeros per row) from a host-made row descriptor list, then
// applied in ONE pass into a second ld x ld buffer that becomes P (k_map_one: out[a][b] = sum_t val[b][t] * T[a][col[b][t]] with
// T[a][c] = sum_t val[a][t] * P[col[a][t]][c] formed on the fly and rounded as a stored T would be; almost every entry is a plain copy
// P[src(a)][src(b)] -- 36 MB read + 36 MB written at N = 500 instead of the two gather passes T = A P, P = T A' of rounds 1-3, which moved
// twice that at half the rate: 53 -> ~17 us per call).  Rows that are plain copies (coefficient 1) reproduce their source bit for bit.
#include <algorithm>

#include "pre3_internal.h"

namespace pre3 {

struct CamM { double f, Cx, Cy, k1, k2; };
constexpr int MAPW = 8;          // ELL width of A
constexpr int FEATW = 64;        // doubles per new feature: y[6], dth_dq[4], dph_dq[4], Nn[36]
constexpr int CONVW = 24;        // doubles per landmark for the conversion: p[3], J[18]

__device__ inline void m_q2r(const double *q, double *R)
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = r * r + x * x - y * y - z * z; R[1] = 2 * (x * y - r * z);           R[2] = 2 * (z * x + r * y);
    R[3] = 2 * (x * y + r * z);           R[4] = r * r - x * x + y * y - z * z; R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (z * x - r * y);           R[7] = 2 * (y * z + r * x);           R[8] = r * r - x * x - y * y + z * z;
}

// hinv_my_version.m:26-53 and the Jacobians of add_a_feature_covariance_inverse_depth.m:29-82, one lane per new feature
__global__ void k_map_new_features(int n_new, const double *__restrict__ uvd, const double *__restrict__ rho0, double std_pxl,
                                   const double *__restrict__ x, CamM cam, double *__restrict__ feat)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_new) return;
    const double ud = uvd[2 * f], vd = uvd[2 * f + 1];
    // undistort_fm_my_version.m:27-48
    double xd = (ud - cam.Cx) / cam.f, yd = (vd - cam.Cy) / cam.f;
    double rd = sqrt(xd * xd + yd * yd);
    double ru = rd / (1 + cam.k1 * rd * rd + cam.k2 * rd * rd * rd * rd);
    for (int k = 0; k < 10; ++k) {
        double f1 = ru + cam.k1 * ru * ru * ru + cam.k2 * ru * ru * ru * ru * ru - rd;
        double f1p = 1 + 3 * cam.k1 * ru * ru + 5 * cam.k2 * ru * ru * ru * ru;
        ru = ru - f1 / f1p;
    }
    const double Dd = 1 + cam.k1 * ru * ru + cam.k2 * ru * ru * ru * ru;
    const double uu = cam.f * xd / Dd + cam.Cx, vu = cam.f * yd / Dd + cam.Cy;
    double R[9];
    m_q2r(x + 3, R);
    const double hc[3] = { -(cam.Cx - uu) / cam.f, -(cam.Cy - vu) / cam.f, 1.0 };
    double nw[3];
    for (int i = 0; i < 3; ++i) nw[i] = R[i * 3] * hc[0] + R[i * 3 + 1] * hc[1] + R[i * 3 + 2] * hc[2];
    double *o = feat + (size_t)f * FEATW;
    o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
    o[3] = atan2(nw[0], nw[2]); o[4] = atan2(-nw[1], sqrt(nw[0] * nw[0] + nw[2] * nw[2])); o[5] = rho0[f];
    const double Xw = nw[0], Yw = nw[1], Zw = nw[2];
    const double dth[3] = { Zw / (Xw * Xw + Zw * Zw), 0, -Xw / (Xw * Xw + Zw * Zw) };
    const double s2 = Xw * Xw + Yw * Yw + Zw * Zw, sxz = sqrt(Xw * Xw + Zw * Zw);
    const double dph[3] = { (Xw * Yw) / (s2 * sxz), -sxz / s2, (Zw * Yw) / (s2 * sxz) };
    // dRq_times_a_by_dq(q_wc, XYZ_c)  (dRq_times_a_by_dq.m:29-101)
    const double q0 = x[3], qx = x[4], qy = x[5], qz = x[6], a0 = hc[0], a1 = hc[1], a2 = hc[2];
    double dq[12];
    dq[0] = 2 * q0 * a0 - 2 * qz * a1 + 2 * qy * a2;  dq[4] = 2 * qz * a0 + 2 * q0 * a1 - 2 * qx * a2;  dq[8]  = -2 * qy * a0 + 2 * qx * a1 + 2 * q0 * a2;
    dq[1] = 2 * qx * a0 + 2 * qy * a1 + 2 * qz * a2;  dq[5] = 2 * qy * a0 - 2 * qx * a1 - 2 * q0 * a2;  dq[9]  = 2 * qz * a0 + 2 * q0 * a1 - 2 * qx * a2;
    dq[2] = -2 * qy * a0 + 2 * qx * a1 + 2 * q0 * a2; dq[6] = 2 * qx * a0 + 2 * qy * a1 + 2 * qz * a2;  dq[10] = -2 * q0 * a0 + 2 * qz * a1 - 2 * qy * a2;
    dq[3] = -2 * qz * a0 - 2 * q0 * a1 + 2 * qx * a2; dq[7] = 2 * q0 * a0 - 2 * qz * a1 + 2 * qy * a2;  dq[11] = 2 * qx * a0 + 2 * qy * a1 + 2 * qz * a2;
    for (int c = 0; c < 4; ++c) {
        o[6 + c] = dth[0] * dq[c] + dth[1] * dq[4 + c] + dth[2] * dq[8 + c];
        o[10 + c] = dph[0] * dq[c] + dph[1] * dq[4 + c] + dph[2] * dq[8 + c];
    }
    // dy_dhd = [dyprima_dgw * R_wc * dgc_dhu * dhu_dhd , 0 ; 0 0 1],  dhu_dhd = inv(jacob_distor(uvd))
    const double xx = ud - cam.Cx, yy = vd - cam.Cy, f2 = cam.f * cam.f;
    const double r2 = (xx * xx + yy * yy) / f2, r4 = r2 * r2, g = cam.k1 + 2 * cam.k2 * r2, D0 = 1 + cam.k1 * r2 + cam.k2 * r4;
    const double Jd[4] = { D0 + xx * g * (2 * xx / f2), xx * g * (2 * yy / f2), yy * g * (2 * xx / f2), D0 + yy * g * (2 * yy / f2) };
    const double det = Jd[0] * Jd[3] - Jd[1] * Jd[2];
    const double Ji[4] = { Jd[3] / det, -Jd[1] / det, -Jd[2] / det, Jd[0] / det };
    double A[6][3];                         // dy_dhd
    for (int r_ = 0; r_ < 6; ++r_) for (int c = 0; c < 3; ++c) A[r_][c] = 0;
    for (int r_ = 3; r_ < 5; ++r_) {
        const double *dg = r_ == 3 ? dth : dph;
        double t1[3];                       // row * R_wc
        for (int c = 0; c < 3; ++c) t1[c] = dg[0] * R[c] + dg[1] * R[3 + c] + dg[2] * R[6 + c];
        const double t2[2] = { t1[0] / cam.f, t1[1] / cam.f };          // * dgc_dhu
        A[r_][0] = t2[0] * Ji[0] + t2[1] * Ji[2];
        A[r_][1] = t2[0] * Ji[1] + t2[1] * Ji[3];
    }
    A[5][2] = 1;
    const double std_rho = rho0[f] * rho0[f] * 0.01;                    // add_features_inverse_depth.m:41
    const double pd[3] = { std_pxl * std_pxl, std_pxl * std_pxl, std_rho * std_rho };
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += (A[i][t] * pd[t]) * A[j][t];
            o[14 + i * 6 + j] = s;
        }
}

// inversedepth_2_cartesian.m:38-57: linearity index per inverse-depth landmark, plus the point p and the 3x6 Jacobian
template <typename T>
__global__ void k_map_convert_flags(int N, const int32_t *__restrict__ lm_type, const int32_t *__restrict__ lm_off,
                                    const double *__restrict__ x, const T *__restrict__ P, int ld, double threshold,
                                    int32_t *__restrict__ flags, double *__restrict__ conv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int flag = 0;
    if (lm_type[i] == PRE3_INVDEPTH) {
        const int o = lm_off[i];
        const double std_rho = sqrt((double)P[(size_t)(o + 5) * ld + o + 5]);
        const double rho = x[o + 5], std_d = std_rho / (rho * rho), theta = x[o + 3], phi = x[o + 4];
        const double cphi = cos(phi);
        const double mi[3] = { cphi * sin(theta), -sin(phi), cphi * cos(theta) };
        const double p[3] = { x[o] + (1 / rho) * mi[0], x[o + 1] + (1 / rho) * mi[1], x[o + 2] + (1 / rho) * mi[2] };
        const double a[3] = { p[0] - x[o], p[1] - x[o + 1], p[2] - x[o + 2] }, c2[3] = { p[0] - x[0], p[1] - x[1], p[2] - x[2] };
        const double d_c2p = sqrt(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
        const double cos_alpha = (a[0] * c2[0] + a[1] * c2[1] + a[2] * c2[2]) / (sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * d_c2p);
        flag = (4 * std_d * cos_alpha / d_c2p) < threshold ? 1 : 0;
        double *cv = conv + (size_t)i * CONVW;
        const double dmt[3] = { cos(phi) * cos(theta), 0, -cos(phi) * sin(theta) };
        const double dmp[3] = { -sin(phi) * sin(theta), -cos(phi), -sin(phi) * cos(theta) };
        for (int r_ = 0; r_ < 3; ++r_) {
            cv[r_] = p[r_];
            for (int c = 0; c < 3; ++c) cv[3 + r_ * 6 + c] = r_ == c ? 1.0 : 0.0;
            cv[3 + r_ * 6 + 3] = (1 / rho) * dmt[r_]; cv[3 + r_ * 6 + 4] = (1 / rho) * dmp[r_]; cv[3 + r_ * 6 + 5] = -mi[r_] / (rho * rho);
        }
    }
    flags[i] = flag;
}

// Row a of A and entry a of the new state from its descriptor (kind, p0, p1):
//   0: copy of old row p0;  1: row p1 (0..5) of new feature p0;  2: row p1 (0..2) of the conversion of landmark p0 (old offset in desc[3a+... see host)
template <typename T>
__global__ void k_map_fill(int n_new, const int32_t *__restrict__ desc, const double *__restrict__ x_old, const double *__restrict__ feat,
                           const double *__restrict__ conv, const int32_t *__restrict__ lm_off_old, int32_t *__restrict__ col,
                           T *__restrict__ val, double *__restrict__ x_new, int32_t *__restrict__ src0)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_new) return;
    const int kind = desc[3 * a], p0 = desc[3 * a + 1], p1 = desc[3 * a + 2];
    src0[a] = kind == 0 ? p0 : -1;                  // k_map_one: the old row / column a plain copy comes from
    int32_t *cc = col + a * MAPW;
    T *vv = val + a * MAPW;
    for (int t = 0; t < MAPW; ++t) { cc[t] = 0; vv[t] = (T)0; }
    if (kind == 0) {
        cc[0] = p0; vv[0] = (T)1; x_new[a] = x_old[p0];
    } else if (kind == 1) {
        const double *o = feat + (size_t)p0 * FEATW;
        x_new[a] = o[p1];
        if (p1 < 3) { cc[0] = p1; vv[0] = (T)1; }
        else if (p1 < 5) { for (int c = 0; c < 4; ++c) { cc[c] = 3 + c; vv[c] = (T)o[(p1 == 3 ? 6 : 10) + c]; } }
    } else {
        const double *cv = conv + (size_t)p0 * CONVW;
        const int o = lm_off_old[p0];
        x_new[a] = cv[p1];
        for (int c = 0; c < 6; ++c) { cc[c] = o + c; vv[c] = (T)cv[3 + p1 * 6 + c]; }
    }
}

// T = A P : dst[a][j] = sum_t val[a][t] * P[col[a][t]][j]   (a < n_new; zero rows beyond)
template <typename T>
__global__ __launch_bounds__(256) void k_map_rows(int n_new, const int32_t *__restrict__ desc, const int32_t *__restrict__ col,
                                                  const T *__restrict__ val, const T *__restrict__ P, int ld, T *__restrict__ dst)
{
    const int a = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    T s = (T)0;
    if (a < n_new) {
        if (desc[3 * a] == 0) s = P[(size_t)col[a * MAPW] * ld + j];          // plain copy row (almost all of them): one load
        else {
#pragma unroll
            for (int t = 0; t < MAPW; ++t) s += val[a * MAPW + t] * P[(size_t)col[a * MAPW + t] * ld + j];
        }
    }
    dst[(size_t)a * ld + j] = s;
}

// P = T A' : dst[a][b] = sum_t val[b][t] * Tm[a][col[b][t]]   (a, b < n_new; zero elsewhere)
template <typename T>
__global__ __launch_bounds__(256) void k_map_cols(int n_new, const int32_t *__restrict__ desc, const int32_t *__restrict__ col,
                                                  const T *__restrict__ val, const T *__restrict__ Tm, int ld, T *__restrict__ dst)
{
    const int a = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ld) return;
    T s = (T)0;
    if (a < n_new && b < n_new) {
        if (desc[3 * b] == 0) s = Tm[(size_t)a * ld + col[b * MAPW]];
        else {
#pragma unroll
            for (int t = 0; t < MAPW; ++t) s += val[b * MAPW + t] * Tm[(size_t)a * ld + col[b * MAPW + t]];
        }
    }
    dst[(size_t)a * ld + b] = s;
}

// Everything that follows the congruence as riders of the congruence's launch (k_map_one; round 4: it was one copy, two uploads, three fills and the bank gather with two
// stream synchronisations): the new state, the new landmark table, the per-landmark fields cleared (update_features_info.m:30-44), the
// inbox cleared, and the descriptor bank re-laid-out (src[i] = old index of new landmark i, -1: a new landmark, zero descriptor).
struct MapFinish {
    int n_new; const double *x_alt; double *x_kk; int N; const int32_t *types_src, *off_src; int32_t *lm_type, *lm_off; int capN;
    int32_t *has_h, *has_S, *inbox; int inbox_words; const int32_t *src; const double *bank; double *bank_out;
};
__device__ __forceinline__ void map_finish_block(const MapFinish &f, int t, int nt)
{
    for (int i = t; i < f.n_new; i += nt) f.x_kk[i] = f.x_alt[i];
    for (int i = t; i < f.N; i += nt) { f.lm_type[i] = f.types_src[i]; f.lm_off[i] = f.off_src[i]; }
    for (int i = t; i < f.capN; i += nt) { f.has_h[i] = 0; f.has_S[i] = 0; }
    for (int i = t; i < f.inbox_words; i += nt) f.inbox[i] = 0;
    if (f.bank != nullptr)
        for (int i = t; i < f.N * 128; i += nt) { const int sidx = f.src[i >> 7]; f.bank_out[i] = sidx >= 0 ? f.bank[(size_t)sidx * 128 + (i & 127)] : 0.0; }
}
__global__ __launch_bounds__(256) void k_map_finish(MapFinish f) { map_finish_block(f, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }

// out = A P A' (+ D) in one pass: out[a][b] = sum_t val[b][t] * T[a][col[b][t]],  T[a][c] = sum_t val[a][t] * P[col[a][t]][c] -- the sums of
// k_map_rows / k_map_cols term for term, T rounded to the storage type as the stored intermediate was.  A workgroup writes 8 rows x 1024
// columns; where row and columns are plain copies (src0 >= 0: everything but the few new / converted rows) an entry is one load.
// feat != nullptr: the new features' 6x6 noise blocks (add_a_feature_covariance_inverse_depth.m:88-90, k_map_add_noise) are added here.
template <typename T>
__device__ __forceinline__ T map_tm(const int32_t *__restrict__ col, const T *__restrict__ val, const T *__restrict__ P, int ld, int a, int ka, int c)
{
    if (ka == 0) return P[(size_t)col[a * MAPW] * ld + c];
    T s = (T)0;
#pragma unroll
    for (int t = 0; t < MAPW; ++t) s = ell_fma(val[a * MAPW + t], P[(size_t)col[a * MAPW + t] * ld + c], s);
    return s;
}
// an entry whose row AND column are computed (the new features' own blocks, a converted landmark's block: a few dozen entries per call)
template <typename T>
__device__ __attribute__((noinline)) T map_entry(int a, int b, const int32_t *__restrict__ desc, const int32_t *__restrict__ col, const T *__restrict__ val,
                                                 const T *__restrict__ P, int ld, const double *__restrict__ feat)
{
    const int ka = desc[3 * a];
    T tm[MAPW];                                     // (all 64 loads in flight: the few lanes that come here are the longest chain of the launch)
#pragma unroll
    for (int t = 0; t < MAPW; ++t) tm[t] = map_tm(col, val, P, ld, a, ka, col[b * MAPW + t]);
    T s = (T)0;
#pragma unroll
    for (int t = 0; t < MAPW; ++t) s = ell_fma(val[b * MAPW + t], tm[t], s);
    if (feat != nullptr && ka == 1 && desc[3 * b] == 1 && desc[3 * a + 1] == desc[3 * b + 1])
        s = (T)((double)s + feat[(size_t)desc[3 * a + 1] * FEATW + 14 + desc[3 * a + 2] * 6 + desc[3 * b + 2]]);
    return s;
}
template <typename T>
__global__ __launch_bounds__(256) void k_map_one(int n_new, const int32_t *__restrict__ desc, const int32_t *__restrict__ col, const T *__restrict__ val,
                                                 const int32_t *__restrict__ src0, const T *__restrict__ P, int ld, T *__restrict__ dst,
                                                 const double *__restrict__ feat, int ny, MapFinish fin)
{
    constexpr int RB = 8, CB = 4;                                                     // a workgroup writes RB rows x CB * 256 columns
    // the block rows behind the congruence's own: what follows it (new state, landmark table, cleared fields, inbox, descriptor bank) -- it
    // needs nothing of this launch, so it rides here instead of going out as k_map_finish
    if ((int)blockIdx.y >= ny) { map_finish_block(fin, (((int)blockIdx.y - ny) * gridDim.x + blockIdx.x) * 256 + threadIdx.x, ((int)gridDim.y - ny) * gridDim.x * 256); return; }
    // (ld is a multiple of 128: all RB rows exist.)  Last rows first: the computed rows of an add sit at the end of the state and take a dozen
    // microseconds of dependent loads -- dispatched first they hide behind the copies, dispatched last they were the tail of the launch
    const int a0 = (ny - 1 - (int)blockIdx.y) * RB;
    int sa[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) sa[r] = a0 + r < n_new ? src0[a0 + r] : -2;
    bool rows_plain = true;
#pragma unroll
    for (int r = 0; r < RB; ++r) rows_plain = rows_plain && sa[r] != -1;
    if (rows_plain) {
        // Round 5 -- the common case (workgroup-uniform: no computed row among the RB) at HBM rate: a lane takes FOUR consecutive destination
        // columns.  Behind a deleted landmark their sources are four consecutive columns too, shifted by a multiple of 3 entries: element-aligned
        // only, which a 16-byte global load takes (the hardware needs dword alignment; tools/probe_unaligned.hip), so a run moves as 16-byte loads
        // and aligned 16-byte stores instead of the dword accesses of round 4 (2.9 TB/s).  A lane whose four columns are not such a run (the edge
        // of a deletion, a computed column of a new / converted landmark, the end of the state) takes them one by one, as before.
        typedef T tv4_t __attribute__((ext_vector_type(4), aligned(sizeof(T))));
        typedef T tv4a_t __attribute__((ext_vector_type(4)));
        const int c4 = blockIdx.x * CB * 256 + threadIdx.x * 4;
        if (c4 >= ld) return;
        int sb4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sb4[j] = c4 + j < n_new ? src0[c4 + j] : -2;     // -1: a computed column, -2: beyond the new state (zero)
        const bool run = sb4[0] >= 0 && sb4[1] == sb4[0] + 1 && sb4[2] == sb4[0] + 2 && sb4[3] == sb4[0] + 3;
        tv4a_t o[RB];
        if (run) {
            tv4_t t[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) t[r] = *reinterpret_cast<const tv4_t *>(P + (size_t)(sa[r] >= 0 ? sa[r] : 0) * ld + sb4[0]);
#pragma unroll
            for (int r = 0; r < RB; ++r) o[r] = sa[r] >= 0 ? tv4a_t{ t[r].x, t[r].y, t[r].z, t[r].w } : tv4a_t{ (T)0, (T)0, (T)0, (T)0 };
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sb4[j] != -1) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const T v = P[(size_t)(sa[r] >= 0 ? sa[r] : 0) * ld + (sb4[j] >= 0 ? sb4[j] : 0)];
                        o[r][j] = (sa[r] >= 0 && sb4[j] >= 0) ? v : (T)0;
                    }
                } else {
                    // a computed column (a new feature's, a converted landmark's: a handful of lanes per call): its <= 8 terms for all RB rows at once
                    T vv[MAPW]; int cc[MAPW];
#pragma unroll
                    for (int t = 0; t < MAPW; ++t) { vv[t] = val[(c4 + j) * MAPW + t]; cc[t] = col[(c4 + j) * MAPW + t]; }
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        T e = (T)0;
#pragma unroll
                        for (int t = 0; t < MAPW; ++t) e = ell_fma(vv[t], P[(size_t)(sa[r] >= 0 ? sa[r] : 0) * ld + cc[t]], e);
                        o[r][j] = sa[r] >= 0 ? e : (T)0;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) *reinterpret_cast<tv4a_t *>(dst + (size_t)(a0 + r) * ld + c4) = o[r];
        return;
    }
    int cb[CB], sb[CB];                                                               // lane-consecutive columns: every access of a wave is one contiguous run
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        cb[k] = (blockIdx.x * CB + k) * 256 + threadIdx.x;
        sb[k] = cb[k] < n_new ? src0[cb[k]] : -2;                                    // -1: a computed column, -2: beyond the new state (zero)
    }
#pragma unroll 1
    for (int r = 0; r < RB; ++r) {
        const int a = a0 + r;
        const int ka = sa[r] == -1 ? desc[3 * a] : 0;
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (cb[k] >= ld) continue;
            T e = (T)0;
            if (sa[r] != -2 && sb[k] != -2) {
                if (sb[k] >= 0) e = map_tm(col, val, P, ld, a, ka, sb[k]);           // (a copied row: one load; a computed row: its <= 8 terms, unrolled)
                else if (sa[r] >= 0) {                                               // a computed column of a copied row
#pragma unroll
                    for (int t = 0; t < MAPW; ++t) e = ell_fma(val[cb[k] * MAPW + t], P[(size_t)sa[r] * ld + col[cb[k] * MAPW + t]], e);
                } else e = map_entry(a, cb[k], desc, col, val, P, ld, feat);
            }
            dst[(size_t)a * ld + cb[k]] = e;
        }
    }
}

template <typename T>
__global__ void k_map_add_noise(int n_feat, int first_off, const double *__restrict__ feat, T *__restrict__ P, int ld)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_feat * 36) return;
    const int f = t / 36, i = (t % 36) / 6, j = t % 6;
    const size_t o = (size_t)(first_off + 6 * f + i) * ld + first_off + 6 * f + j;
    P[o] = (T)((double)P[o] + feat[(size_t)f * FEATW + 14 + i * 6 + j]);
}

#define DISPATCH_T(c, expr_f64, expr_f32) do { if ((c)->dtype == PRE3_F64) { expr_f64; } else { expr_f32; } } while (0)

static int ensure_map_buffers(pre3_ctx *c)
{
    if (c->P_alt) return PRE3_OK;
    auto bytes = [&](void **p, size_t b) { return hipMalloc(p, b ? b : 16) == hipSuccess; };
    bool ok = bytes(&c->P_alt, (size_t)c->ld * c->ld * c->esz) && bytes((void **)&c->x_alt, sizeof(double) * c->capn) &&
              bytes((void **)&c->map_col, sizeof(int32_t) * (size_t)c->capn * MAPW) && bytes(&c->map_val, c->esz * (size_t)c->capn * MAPW) &&
              bytes((void **)&c->map_desc, sizeof(int32_t) * (3 * (size_t)c->capn + 3 * (size_t)c->capN + 16) + sizeof(double) * 3 * (size_t)c->capN + 16) &&
              bytes((void **)&c->map_feat, sizeof(double) * (size_t)c->capN * (FEATW > CONVW ? FEATW : CONVW)) &&
              bytes((void **)&c->map_flags, sizeof(int32_t) * c->capN) && bytes((void **)&c->map_src0, sizeof(int32_t) * (size_t)c->ld) &&
              bytes((void **)&c->map_conv, sizeof(double) * (size_t)c->capN * CONVW);
    if (!ok) { set_error("map management: device allocation failed"); return PRE3_E_NOMEM; }
    // two pinned staging blocks ([desc | types | off | src | uvd, rho]: ONE upload per call), used alternately: a block is written again only
    // after the call before last has been consumed (its event), so a call does not end in a stream synchronisation
    c->map_stage_bytes = (sizeof(int32_t) * (3 * (size_t)c->capn + 3 * (size_t)c->capN + 16) + sizeof(double) * 3 * (size_t)c->capN + 15) & ~(size_t)15;
    for (int k = 0; k < 2; ++k) {
        if (hipHostMalloc(&c->map_stage[k], c->map_stage_bytes, hipHostMallocMapped) != hipSuccess || hipEventCreateWithFlags(&c->map_stage_ev[k], hipEventDisableTiming) != hipSuccess) {
            set_error("map management: pinned staging allocation failed"); return PRE3_E_NOMEM;
        }
    }
    return PRE3_OK;
}

// apply the state map described by desc (3 ints per new row), then install the new landmark table.  uvd_rho: 3 n_feat doubles ([u v] per new
// feature, then rho0 per feature) for pre3_map_add_inverse_depth, staged with everything else.
static int apply_map(pre3_ctx *c, const std::vector<int32_t> &desc, int n_new, const std::vector<int32_t> &new_types,
                     int n_feat, int first_new_off, const std::vector<int32_t> &lm_src, const double *uvd = nullptr, const double *rho0 = nullptr,
                     double std_pxl = 0.0)
{
    PRE3_CHECK(n_new <= c->capn && (int)new_types.size() <= c->capN, PRE3_E_ARG, "map management: the new map (N=%zu, n=%d) exceeds the context capacity (N=%d, n=%d)",
               new_types.size(), n_new, c->capN, c->capn);
    const int N = (int)new_types.size();
    std::vector<int32_t> off(N ? N : 1);
    int n = 13;
    for (int i = 0; i < N; ++i) { off[i] = n; n += new_types[i] == PRE3_INVDEPTH ? 6 : 3; }
    PRE3_CHECK(n == n_new, PRE3_E_STATE, "map management: internal size mismatch (%d vs %d)", n, n_new);
    // ---- stage: [desc 3 n_new | types N | off N | src N | pad to 8 bytes | uvd 2 n_feat, rho n_feat]
    const int k = c->map_stage_next; c->map_stage_next ^= 1;
    if (c->map_stage_used[k]) PRE3_TRY(stage_wait(c, 2 + k));
    int32_t *st = static_cast<int32_t *>(c->map_stage[k]);
    const size_t o_types = desc.size(), o_off = o_types + (size_t)N, o_src = o_off + (size_t)N, o_end = (o_src + (size_t)N + 1) & ~(size_t)1;
    memcpy(st, desc.data(), sizeof(int32_t) * desc.size());
    if (N) { memcpy(st + o_types, new_types.data(), sizeof(int32_t) * N); memcpy(st + o_off, off.data(), sizeof(int32_t) * N); memcpy(st + o_src, lm_src.data(), sizeof(int32_t) * N); }
    size_t bytes = sizeof(int32_t) * o_end;
    if (n_feat > 0) {
        double *sd = reinterpret_cast<double *>(st + o_end);
        memcpy(sd, uvd, sizeof(double) * 2 * n_feat); memcpy(sd + 2 * n_feat, rho0, sizeof(double) * n_feat);
        bytes += sizeof(double) * 3 * (size_t)n_feat;
    }
    PRE3_CHECK(bytes <= c->map_stage_bytes, PRE3_E_ARG, "map management: staging block too small");
    PRE3_TRY(launch_pull(c, st, c->map_desc, bytes, 2 + k));        // (read over PCIe by the device: no DMA-engine copy; announces itself: no event)
    c->map_stage_used[k] = true;
    const int32_t *d_types = c->map_desc + o_types, *d_off = c->map_desc + o_off, *d_src = c->map_desc + o_src;
    if (n_feat > 0) {
        const double *d_uvd = reinterpret_cast<const double *>(c->map_desc + o_end);
        CamM cam{ c->cam.f, c->cam.Cx, c->cam.Cy, c->cam.k1, c->cam.k2 };
        hipLaunchKernelGGL(k_map_new_features, dim3(ceil_div(n_feat, 64)), dim3(64), 0, c->stream, n_feat, d_uvd, d_uvd + 2 * n_feat, std_pxl, c->x_kk, cam, c->map_feat);
    }
    const double *feat = c->map_feat, *conv = c->map_conv;
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_map_fill<double>, dim3(ceil_div(n_new, 256)), dim3(256), 0, c->stream, n_new, c->map_desc, c->x_kk, feat, conv, c->lm.off, c->map_col, (double *)c->map_val, c->x_alt, c->map_src0),
        hipLaunchKernelGGL(k_map_fill<float>, dim3(ceil_div(n_new, 256)), dim3(256), 0, c->stream, n_new, c->map_desc, c->x_kk, feat, conv, c->lm.off, c->map_col, (float *)c->map_val, c->x_alt, c->map_src0));
    // new state, landmark table, cleared per-landmark fields and inbox, re-laid-out descriptor bank: riders of the congruence's launch (or one launch behind the two-pass form); no synchronisation
    const bool with_bank = c->bank != nullptr && c->bank_alt != nullptr && N > 0;
    const int fin_work = std::max(std::max(n_new, (int)(c->inbox_bytes / 4)), with_bank ? N * 128 : 0);
    const MapFinish fin{ n_new, c->x_alt, c->x_kk, N, d_types, d_off, c->lm.type, c->lm.off, c->capN, c->lm.has_h, c->lm.has_S, (int32_t *)c->inbox_dev, (int)(c->inbox_bytes / 4),
                         d_src, with_bank ? c->bank : nullptr, c->bank_alt };
    static const int one_pass = getenv("PRE3_MAP_ONE_PASS") ? atoi(getenv("PRE3_MAP_ONE_PASS")) : 1;
    if (one_pass) {
        // one pass into the second buffer, which becomes P
        const int gx = ceil_div(c->ld, 1024), ny = c->ld / 8, fin_rows = std::min(64, ceil_div(ceil_div(fin_work, 256), gx));
        dim3 g1(gx, ny + fin_rows), b(256);
        const double *noise = n_feat > 0 ? feat : nullptr;
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_map_one<double>, g1, b, 0, c->stream, n_new, c->map_desc, c->map_col, (const double *)c->map_val, c->map_src0, (const double *)c->P, c->ld, (double *)c->P_alt, noise, ny, fin),
            hipLaunchKernelGGL(k_map_one<float>, g1, b, 0, c->stream, n_new, c->map_desc, c->map_col, (const float *)c->map_val, c->map_src0, (const float *)c->P, c->ld, (float *)c->P_alt, noise, ny, fin));
        std::swap(c->P, c->P_alt);
    } else {
        dim3 g(ceil_div(c->ld, 256), c->ld), b(256);
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_map_rows<double>, g, b, 0, c->stream, n_new, c->map_desc, c->map_col, (const double *)c->map_val, (const double *)c->P, c->ld, (double *)c->P_alt),
            hipLaunchKernelGGL(k_map_rows<float>, g, b, 0, c->stream, n_new, c->map_desc, c->map_col, (const float *)c->map_val, (const float *)c->P, c->ld, (float *)c->P_alt));
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_map_cols<double>, g, b, 0, c->stream, n_new, c->map_desc, c->map_col, (const double *)c->map_val, (const double *)c->P_alt, c->ld, (double *)c->P),
            hipLaunchKernelGGL(k_map_cols<float>, g, b, 0, c->stream, n_new, c->map_desc, c->map_col, (const float *)c->map_val, (const float *)c->P_alt, c->ld, (float *)c->P));
        if (n_feat > 0) {
            DISPATCH_T(c,
                hipLaunchKernelGGL(k_map_add_noise<double>, dim3(ceil_div(n_feat * 36, 256)), dim3(256), 0, c->stream, n_feat, first_new_off, feat, (double *)c->P, c->ld),
                hipLaunchKernelGGL(k_map_add_noise<float>, dim3(ceil_div(n_feat * 36, 256)), dim3(256), 0, c->stream, n_feat, first_new_off, feat, (float *)c->P, c->ld));
        }
        hipLaunchKernelGGL(k_map_finish, dim3(std::min(1024, ceil_div(fin_work, 256))), dim3(256), 0, c->stream, fin);
    }
    PRE3_HIP(hipGetLastError());
    if (with_bank) std::swap(c->bank, c->bank_alt);
    c->N = N; c->n = n; c->lm_type_host = new_types;
    c->m = 0; c->meas_host.clear(); c->measurements_set = false; c->projected = false; c->innovated = false; c->hp_all_valid = false;
    c->li_from_host = c->hi_from_host = -1; c->li_kernel = c->hi_kernel = false;
    c->x_valid[PRE3_X_K_KM1] = false;
    return PRE3_OK;
}

static int map_precheck(pre3_ctx *c, const char *who)
{
    PRE3_CHECK(c != nullptr, PRE3_E_ARG, "null context");
    PRE3_HIP(hipSetDevice(c->device));
    PRE3_CHECK(c->x_valid[PRE3_X_K_K] && c->p_which == PRE3_X_K_K, PRE3_E_STATE, "%s: needs (x_k_k, p_k_k) on the device (map management runs between steps)", who);
    return ensure_map_buffers(c);
}

}  // namespace pre3

using namespace pre3;

extern "C" {

int pre3_get_map(pre3_ctx *c, int32_t *lm_type_out)
{
    if (!c) return PRE3_E_ARG;
    if (lm_type_out) for (int i = 0; i < c->N; ++i) lm_type_out[i] = c->lm_type_host[i];
    return c->N;
}

int pre3_map_delete(pre3_ctx *c, int n_del, const int32_t *del_idx)
{
    PRE3_TRY(map_precheck(c, "pre3_map_delete"));
    PRE3_CHECK(n_del >= 0 && (n_del == 0 || del_idx), PRE3_E_ARG, "pre3_map_delete: bad arguments");
    for (int d = 0; d < n_del; ++d)
        PRE3_CHECK(del_idx[d] >= 0 && del_idx[d] < c->N && (d == 0 || del_idx[d] > del_idx[d - 1]), PRE3_E_ARG, "pre3_map_delete: indices must be ascending and in range");
    if (n_del == 0) return PRE3_OK;
    std::vector<int32_t> desc, types, src;
    for (int i = 0; i < 13; ++i) { desc.push_back(0); desc.push_back(i); desc.push_back(0); }
    int d = 0, off = 13;
    for (int i = 0; i < c->N; ++i) {
        const int dim = c->lm_type_host[i] == PRE3_INVDEPTH ? 6 : 3;
        if (d < n_del && del_idx[d] == i) { ++d; off += dim; continue; }
        for (int q = 0; q < dim; ++q) { desc.push_back(0); desc.push_back(off + q); desc.push_back(0); }
        types.push_back(c->lm_type_host[i]); src.push_back(i);
        off += dim;
    }
    return apply_map(c, desc, (int)desc.size() / 3, types, 0, 0, src);
}

int pre3_map_add_inverse_depth(pre3_ctx *c, int n_new, const double *uvd, double std_pxl, const double *initial_rho)
{
    PRE3_TRY(map_precheck(c, "pre3_map_add_inverse_depth"));
    PRE3_CHECK(c->have_cam, PRE3_E_STATE, "pre3_map_add_inverse_depth: camera not set");
    PRE3_CHECK(n_new >= 0 && (n_new == 0 || (uvd && initial_rho)), PRE3_E_ARG, "pre3_map_add_inverse_depth: bad arguments");
    if (n_new == 0) return PRE3_OK;
    PRE3_CHECK(c->N + n_new <= c->capN, PRE3_E_ARG, "pre3_map_add_inverse_depth: %d + %d landmarks exceed the capacity %d", c->N, n_new, c->capN);
    PRE3_CHECK((size_t)n_new * FEATW <= (size_t)c->capN * FEATW, PRE3_E_ARG, "pre3_map_add_inverse_depth: too many new features in one call");
    std::vector<int32_t> desc, types(c->lm_type_host), src;
    for (int i = 0; i < c->N; ++i) src.push_back(i);
    for (int i = 0; i < c->n; ++i) { desc.push_back(0); desc.push_back(i); desc.push_back(0); }
    for (int f = 0; f < n_new; ++f) {
        for (int q = 0; q < 6; ++q) { desc.push_back(1); desc.push_back(f); desc.push_back(q); }
        types.push_back(PRE3_INVDEPTH); src.push_back(-1);
    }
    return apply_map(c, desc, c->n + 6 * n_new, types, n_new, c->n, src, uvd, initial_rho, std_pxl);
}

int pre3_map_inversedepth_2_cartesian(pre3_ctx *c, double thr, int32_t *converted_out)
{
    PRE3_TRY(map_precheck(c, "pre3_map_inversedepth_2_cartesian"));
    const int N = c->N;
    if (N == 0) return PRE3_OK;
    DISPATCH_T(c,
        hipLaunchKernelGGL(k_map_convert_flags<double>, dim3(ceil_div(N, 64)), dim3(64), 0, c->stream, N, c->lm.type, c->lm.off, c->x_kk, (const double *)c->P, c->ld, thr, c->map_flags, c->map_conv),
        hipLaunchKernelGGL(k_map_convert_flags<float>, dim3(ceil_div(N, 64)), dim3(64), 0, c->stream, N, c->lm.type, c->lm.off, c->x_kk, (const float *)c->P, c->ld, thr, c->map_flags, c->map_conv));
    PRE3_HIP(hipGetLastError());
    std::vector<int32_t> flags(N);
    PRE3_TRY(stream_drain(c, __func__));
    PRE3_HIP(hipMemcpy(flags.data(), c->map_flags, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    if (converted_out) for (int i = 0; i < N; ++i) converted_out[i] = flags[i];
    bool any = false;
    for (int i = 0; i < N; ++i) any |= flags[i] != 0;
    if (!any) return PRE3_OK;
    std::vector<int32_t> desc, types, src;
    for (int i = 0; i < N; ++i) src.push_back(i);
    for (int i = 0; i < 13; ++i) { desc.push_back(0); desc.push_back(i); desc.push_back(0); }
    int off = 13;
    for (int i = 0; i < N; ++i) {
        const int dim = c->lm_type_host[i] == PRE3_INVDEPTH ? 6 : 3;
        if (flags[i]) {
            for (int q = 0; q < 3; ++q) { desc.push_back(2); desc.push_back(i); desc.push_back(q); }
            types.push_back(PRE3_CARTESIAN);
        } else {
            for (int q = 0; q < dim; ++q) { desc.push_back(0); desc.push_back(off + q); desc.push_back(0); }
            types.push_back(c->lm_type_host[i]);
        }
        off += dim;
    }
    return apply_map(c, desc, (int)desc.size() / 3, types, 0, 0, src);
}

// map_management.m:27-79 as ONE call and ONE congruence: delete_features (:33), inversedepth_2_cartesian (:48, convert_threshold < 0: skipped),
// initialize_features' add (:58-66).  All three only select / recombine rows of the OLD state (the new features' Jacobians read the camera
// block, which neither a deletion nor a conversion touches), so their row maps compose into one descriptor list and P makes one pass
// through k_map_one instead of up to three.  converted_out (may be null): per OLD landmark, 1 = converted (a deleted landmark reads 0).
int pre3_map_management(pre3_ctx *c, int n_del, const int32_t *del_idx, double convert_threshold, int32_t *converted_out,
                        int n_new, const double *uvd, double std_pxl, const double *initial_rho)
{
    PRE3_TRY(map_precheck(c, "pre3_map_management"));
    PRE3_CHECK(n_del >= 0 && (n_del == 0 || del_idx), PRE3_E_ARG, "pre3_map_management: bad deletion list");
    for (int d = 0; d < n_del; ++d)
        PRE3_CHECK(del_idx[d] >= 0 && del_idx[d] < c->N && (d == 0 || del_idx[d] > del_idx[d - 1]), PRE3_E_ARG, "pre3_map_management: deletion indices must be ascending and in range");
    PRE3_CHECK(n_new >= 0 && (n_new == 0 || (uvd && initial_rho)), PRE3_E_ARG, "pre3_map_management: bad new-feature arguments");
    PRE3_CHECK(n_new == 0 || c->have_cam, PRE3_E_STATE, "pre3_map_management: camera not set");
    PRE3_CHECK(c->N - n_del + n_new <= c->capN, PRE3_E_ARG, "pre3_map_management: %d - %d + %d landmarks exceed the capacity %d", c->N, n_del, n_new, c->capN);
    const int N = c->N;
    std::vector<int32_t> flags(N ? N : 1, 0);
    if (convert_threshold >= 0 && N > 0) {
        DISPATCH_T(c,
            hipLaunchKernelGGL(k_map_convert_flags<double>, dim3(ceil_div(N, 64)), dim3(64), 0, c->stream, N, c->lm.type, c->lm.off, c->x_kk, (const double *)c->P, c->ld, convert_threshold, c->map_flags, c->map_conv),
            hipLaunchKernelGGL(k_map_convert_flags<float>, dim3(ceil_div(N, 64)), dim3(64), 0, c->stream, N, c->lm.type, c->lm.off, c->x_kk, (const float *)c->P, c->ld, convert_threshold, c->map_flags, c->map_conv));
        PRE3_HIP(hipGetLastError());
        PRE3_TRY(stream_drain(c, __func__));
        PRE3_HIP(hipMemcpy(flags.data(), c->map_flags, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> desc, types, src;
    desc.reserve(3 * (size_t)(c->n + 6 * n_new)); types.reserve(N + n_new); src.reserve(N + n_new);
    for (int i = 0; i < 13; ++i) { desc.push_back(0); desc.push_back(i); desc.push_back(0); }
    int d = 0, off = 13, n_out = 13;
    bool any = n_del > 0 || n_new > 0;
    for (int i = 0; i < N; ++i) {
        const int dim = c->lm_type_host[i] == PRE3_INVDEPTH ? 6 : 3;
        if (d < n_del && del_idx[d] == i) { ++d; off += dim; flags[i] = 0; continue; }
        if (flags[i]) {
            for (int q = 0; q < 3; ++q) { desc.push_back(2); desc.push_back(i); desc.push_back(q); }
            types.push_back(PRE3_CARTESIAN); n_out += 3; any = true;
        } else {
            for (int q = 0; q < dim; ++q) { desc.push_back(0); desc.push_back(off + q); desc.push_back(0); }
            types.push_back(c->lm_type_host[i]); n_out += dim;
        }
        src.push_back(i);
        off += dim;
    }
    if (converted_out) for (int i = 0; i < N; ++i) converted_out[i] = flags[i];
    if (!any) return PRE3_OK;
    const int first_new_off = n_out;
    for (int f = 0; f < n_new; ++f) {
        for (int q = 0; q < 6; ++q) { desc.push_back(1); desc.push_back(f); desc.push_back(q); }
        types.push_back(PRE3_INVDEPTH); src.push_back(-1); n_out += 6;
    }
    return apply_map(c, desc, n_out, types, n_new, first_new_off, src, uvd, initial_rho, std_pxl);
}

}  // extern "C"
