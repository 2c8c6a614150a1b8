"""numpy twin of the CPU oracle: an INDEPENDENT second restatement of the same reference functions,
structured the way the MATLAB code is (interpreted per-landmark / per-hypothesis loops around BLAS-backed
dense algebra: explicit inv(S), K*S*K' as two GEMMs, full n x n temporaries).

TEST INFRASTRUCTURE ONLY (same rule as oracle/__init__.py).  Two uses:
  * cross-check of the C oracle where no reference artefact pins it (prediction, Cartesian landmarks,
    hypothesis loop) -- tests/test_oracle.py;
  * bench.py's cpu_baseline leg ("MATLAB-equivalent restatement": interpreter + multithreaded BLAS, as
    MATLAB R2011a + MKL would run it; MATLAB itself is not available in either container).

Citations: paths relative to /root/reference/matlab_code/.
"""
import numpy as np

INVDEPTH, CARTESIAN = 0, 1


# ---- leaf helpers ----------------------------------------------------------------------------------
def q2r(q):                       # q2r.m:29-36
    r, x, y, z = q
    return np.array([[r * r + x * x - y * y - z * z, 2 * (x * y - r * z), 2 * (z * x + r * y)],
                     [2 * (x * y + r * z), r * r - x * x + y * y - z * z, 2 * (y * z - r * x)],
                     [2 * (z * x - r * y), 2 * (y * z + r * x), r * r - x * x - y * y + z * z]])


def q2R(q):                       # slamToolbox .../Rotations/q2R.m:18-34
    a, b, c, d = q
    aa, ab, ac, ad = a * a, 2 * a * b, 2 * a * c, 2 * a * d
    bb, bc, bd, cc, cd, dd = b * b, 2 * b * c, 2 * b * d, c * c, 2 * c * d, d * d
    return np.array([[aa + bb - cc - dd, bc - ad, bd + ac], [bc + ad, aa - bb + cc - dd, cd - ab], [bd - ac, cd + ab, aa - bb - cc + dd]])


def qProd(q1, q2):                # qProd.m:16-33
    a, b, c, d = q1
    w, x, y, z = q2
    q = np.array([a * w - b * x - c * y - d * z, a * x + b * w + c * z - d * y, a * y - b * z + c * w + d * x, a * z + b * y - c * x + d * w])
    Qq1 = np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])
    Qq2 = np.array([[a, -b, -c, -d], [b, a, -d, c], [c, d, a, -b], [d, -c, b, a]])
    return q, Qq1, Qq2


def normJac(q):                   # normJac.m:27-38
    r, x, y, z = q
    return (r * r + x * x + y * y + z * z) ** (-1.5) * np.array(
        [[x * x + y * y + z * z, -r * x, -r * y, -r * z], [-x * r, r * r + y * y + z * z, -x * y, -x * z],
         [-y * r, -y * x, r * r + x * x + z * z, -y * z], [-z * r, -z * x, -z * y, r * r + x * x + y * y]])


def e2q_jac(e):                   # e2q.m:18-30
    sr, sp, sy = np.sin(e / 2)
    cr, cp, cy = np.cos(e / 2)
    return 0.5 * np.array([[-cy * cp * sr + sy * sp * cr, -cy * sp * cr + sy * cp * sr, -sy * cp * cr + cy * sp * sr],
                           [cy * cp * cr + sy * sp * sr, -cy * sp * sr - sy * cp * cr, -sy * cp * sr - cy * sp * cr],
                           [-cy * sp * sr + sy * cp * cr, cy * cp * cr - sy * sp * sr, -sy * sp * cr + cy * cp * sr],
                           [-sy * cp * sr - cy * sp * cr, -cy * cp * sr - sy * sp * cr, cy * cp * cr + sy * sp * sr]])


def m_dir(theta, phi):            # m.m:38-40
    cphi = np.cos(phi)
    return np.array([cphi * np.sin(theta), -np.sin(phi), cphi * np.cos(theta)])


def distort(uv, cam):             # distort_fm_my_version.m:52-61 (uv: 2 x K)
    f, Cx, Cy, k1, k2 = cam[:5]
    xu, yu = (uv[0] - Cx) / f, (uv[1] - Cy) / f
    ru = np.sqrt(xu * xu + yu * yu)
    D = 1 + k1 * ru ** 2 + k2 * ru ** 4
    return np.stack([xu * D * f + Cx, yu * D * f + Cy])


def jnorm_rebuild(P, J):          # update.m:42-46 / predict_state_and_covariance.m:137-141
    n = P.shape[0]
    return np.block([[P[0:3, 0:3], P[0:3, 3:7] @ J.T, P[0:3, 7:n]],
                     [J @ P[3:7, 0:3], J @ P[3:7, 3:7] @ J.T, J @ P[3:7, 7:n]],
                     [P[7:n, 0:3], P[7:n, 3:7] @ J.T, P[7:n, 7:n]]])


# ---- a2 ------------------------------------------------------------------------------------------------
def process_noise():              # predict_state_and_covariance.m:98-102
    cov_dX = np.diag((0.01 / 3 * np.ones(3)) ** 2)
    e = 0.24 / 2 * np.pi / 180 * np.array([1, 0.1, 1])
    Qe = e2q_jac(e)
    cov_dq = Qe @ np.diag(e ** 2) @ Qe.T
    Pn = np.zeros((7, 7))
    Pn[:3, :3] = cov_dX
    Pn[3:, 3:] = cov_dq
    return Pn


def predict(X_k, P_k, u):         # predict_state_and_covariance.m:59-143
    n = X_k.shape[0]
    q = X_k[3:7]
    R = q2R(q)
    xp = X_k[0:3] + R @ u[0:3]                       # odometry_model.m:50
    qn, Qq1, Qq2 = qProd(q, u[3:7])
    X = np.concatenate([xp, qn, np.zeros(6), X_k[13:]])
    Xo_x = np.block([[np.eye(3), np.zeros((3, 4))], [np.zeros((4, 3)), Qq1]])
    Xo_u = np.block([[R, np.zeros((3, 4))], [np.zeros((4, 3)), Qq2]])
    F = np.block([[Xo_x, np.zeros((7, 6))], [np.zeros((6, 7)), np.eye(6)]])
    G = np.vstack([Xo_u, np.zeros((6, 7))])
    Q = G @ process_noise() @ G.T
    P = np.block([[F @ P_k[0:13, 0:13] @ F.T + Q, F @ P_k[0:13, 13:n]], [P_k[13:n, 0:13] @ F.T, P_k[13:n, 13:n]]])
    J = normJac(X[3:7])
    P = jnorm_rebuild(P, J)
    X[3:7] = X[3:7] / np.linalg.norm(X[3:7])
    return X, P


# ---- a3 / a4 ----------------------------------------------------------------------------------------------
def hi_landmark(t, y, t_wc, r_wc, cam):       # hi_inverse_depth.m:33-85 / hi_cartesian.m:33-81
    if t == INVDEPTH:
        hrl = r_wc.T @ ((y[0:3] - t_wc) * y[5] + m_dir(y[3], y[4]))
    else:
        hrl = np.linalg.inv(r_wc) @ (y[0:3] - t_wc)
    ax, ay = np.degrees(np.arctan2(hrl[0], hrl[2])), np.degrees(np.arctan2(hrl[1], hrl[2]))
    if ax < -60 or ax > 60 or ay < -60 or ay > 60:
        return None
    uv_u = np.array([[cam[1] + (hrl[0] / hrl[2]) * cam[0]], [cam[2] + (hrl[1] / hrl[2]) * cam[0]]])   # hu_my_version.m:41-42
    uv_d = distort(uv_u, cam)[:, 0]
    if 0 < uv_d[0] < cam[6] and 0 < uv_d[1] < cam[5]:
        return uv_d
    return None


def project(types, off, x, cam, h=None, has_h=None):   # predict_camera_measurements.m:27-68
    N = len(types)
    h = np.zeros((N, 2)) if h is None else h.copy()
    has_h = np.zeros(N, np.int32) if has_h is None else has_h.copy()
    r_wc = q2r(x[3:7])
    for i in range(N):
        d = 6 if types[i] == INVDEPTH else 3
        zi = hi_landmark(types[i], x[off[i]:off[i] + d], x[0:3], r_wc, cam)
        if zi is not None:
            h[i] = zi
            has_h[i] = 1
    return h, has_h


def jacob_distor(cam, uv):        # jacob_distor_fm_my_version.m:47-61
    f, Cx, Cy, k1, k2 = cam[:5]
    u, v = uv
    x, y = u - Cx, v - Cy
    r2 = (x * x + y * y) / f ** 2
    r4 = r2 * r2
    return np.array([[(1 + k1 * r2 + k2 * r4) + (u - Cx) * (k1 + 2 * k2 * r2) * (2 * (u - Cx) / f ** 2), (u - Cx) * (k1 + 2 * k2 * r2) * (2 * (v - Cy) / f ** 2)],
                     [(v - Cy) * (k1 + 2 * k2 * r2) * (2 * (u - Cx) / f ** 2), (1 + k1 * r2 + k2 * r4) + (v - Cy) * (k1 + 2 * k2 * r2) * (2 * (v - Cy) / f ** 2)]])


def dRq_times_a_by_dq(q, a):      # dRq_times_a_by_dq.m:29-101
    q0, qx, qy, qz = q
    d0 = np.array([[2 * q0, -2 * qz, 2 * qy], [2 * qz, 2 * q0, -2 * qx], [-2 * qy, 2 * qx, 2 * q0]])
    dx = np.array([[2 * qx, 2 * qy, 2 * qz], [2 * qy, -2 * qx, -2 * q0], [2 * qz, 2 * q0, -2 * qx]])
    dy = np.array([[-2 * qy, 2 * qx, 2 * q0], [2 * qx, 2 * qy, 2 * qz], [-2 * q0, 2 * qz, -2 * qy]])
    dz = np.array([[-2 * qz, -2 * q0, 2 * qx], [2 * q0, -2 * qz, 2 * qy], [2 * qx, 2 * qy, 2 * qz]])
    return np.stack([d0 @ a, dx @ a, dy @ a, dz @ a], 1)


def Hi_landmark(t, xv, y, cam, zi):   # calculate_Hi_inverse_depth_my_version.m / calculate_Hi_cartesian_my_version.m
    f = cam[0]
    Rrw = np.linalg.inv(q2r(xv[3:7]))
    dhd_dhu = np.linalg.inv(np.linalg.inv(jacob_distor(cam, zi)))
    if t == INVDEPTH:
        theta, phi, rho = y[3], y[4], y[5]
        mi = np.array([np.cos(phi) * np.sin(theta), -np.sin(phi), np.cos(phi) * np.cos(theta)])
        a = (y[0:3] - xv[0:3]) * rho + mi
    else:
        rho = 1.0
        a = y[0:3] - xv[0:3]
    hc = Rrw @ a
    dhu_dhrl = np.array([[f / hc[2], 0, -hc[0] * f / hc[2] ** 2], [0, f / hc[2], -hc[1] * f / hc[2] ** 2]])
    dh_dhrl = dhd_dhu @ dhu_dhrl
    Hc = np.zeros((2, 7))
    Hc[:, 0:3] = dh_dhrl @ (-Rrw * rho)
    qc = np.array([xv[3], -xv[4], -xv[5], -xv[6]])
    Hc[:, 3:7] = dh_dhrl @ (dRq_times_a_by_dq(qc, a) @ np.diag([1, -1, -1, -1]))
    Hl = np.zeros((2, 6))
    if t == INVDEPTH:
        dth = Rrw @ np.array([np.cos(phi) * np.cos(theta), 0, -np.cos(phi) * np.sin(theta)])
        dph = Rrw @ np.array([-np.sin(phi) * np.sin(theta), -np.cos(phi), -np.sin(phi) * np.cos(theta)])
        A = np.column_stack([rho * Rrw, dth, dph, Rrw @ (y[0:3] - xv[0:3])])
        Hl[:, :] = dh_dhrl @ A
    else:
        Hl[:, 0:3] = dh_dhrl @ Rrw
    return Hc, Hl


def jacobian(types, off, x, cam, h, has_h):   # calculate_derivatives.m:27-60
    N = len(types)
    Hc, Hl = np.zeros((N, 2, 7)), np.zeros((N, 2, 6))
    for i in range(N):
        if has_h[i]:
            d = 6 if types[i] == INVDEPTH else 3
            Hc[i], Hl[i] = Hi_landmark(types[i], x[0:13], x[off[i]:off[i] + d], cam, h[i])
    return Hc, Hl


def _rows(n, types, off, idx, Hc, Hl):
    """stack dense H rows of landmarks idx (the sparse H the reference concatenates)"""
    H = np.zeros((2 * len(idx), n))
    for s, i in enumerate(idx):
        d = 6 if types[i] == INVDEPTH else 3
        H[2 * s:2 * s + 2, 0:7] = Hc[i]
        H[2 * s:2 * s + 2, off[i]:off[i] + d] = Hl[i][:, :d]
    return H


def _sparse_cols(types, off, idx):
    cols = [np.arange(7)]
    for i in idx:
        d = 6 if types[i] == INVDEPTH else 3
        cols.append(np.arange(off[i], off[i] + d))
    return np.unique(np.concatenate(cols))


def innovation(types, off, P, Hc, Hl, has_h):   # search_IC_matches.m:33-44
    N = len(types)
    n = P.shape[0]
    S = np.zeros((N, 2, 2))
    for i in range(N):
        if has_h[i]:
            c = _sparse_cols(types, off, [i])
            Hi = _rows(n, types, off, [i], Hc, Hl)[:, c]
            S[i] = Hi @ P[np.ix_(c, c)] @ Hi.T + np.eye(2)
    return S


# ---- a9 -----------------------------------------------------------------------------------------------------
def update(x, P, H, R, z, h):     # update.m:27-56; H dense r x n (sparse rows exploited through their column support)
    if z.shape[0] == 0:
        return x.copy(), P.copy(), 0
    c = np.nonzero(np.any(H != 0, axis=0))[0]          # MATLAB multiplies the sparse H: only these columns contribute
    Hc_ = H[:, c]
    PHt = P[:, c] @ Hc_.T
    S = Hc_ @ PHt[c, :] + R
    K = PHt @ np.linalg.inv(S)
    xo = x + K @ (z - h)
    Po = P - K @ S @ K.T
    Po = 0.5 * Po + 0.5 * Po.T
    J = normJac(xo[3:7])
    Po = jnorm_rebuild(Po, J)
    xo[3:7] = xo[3:7] / np.linalg.norm(xo[3:7])
    return xo, Po, K


def update_landmarks(types, off, sel, x, P, Hc, Hl, z, h):
    n = x.shape[0]
    H = _rows(n, types, off, sel, Hc, Hl)
    zz = z[sel].ravel() if len(sel) else np.zeros(0)
    hh = h[sel].ravel() if len(sel) else np.zeros(0)
    xo, Po, _ = update(x, P, H, np.eye(len(zz)), zz, hh)
    return xo, Po


# ---- a6-a8 --------------------------------------------------------------------------------------------------
def support(meas, types, off, xi, cam, z_meas, threshold):   # compute_hypothesis_support_fast.m:33-110
    meas = np.asarray(meas)
    tm = np.asarray(types)[meas]
    rot = q2r(xi[3:7]).T
    m = len(meas)
    mask = np.zeros(m, bool)
    res = np.zeros(m)
    total = 0
    idm = np.nonzero(tm == INVDEPTH)[0]
    if len(idm):
        o = np.asarray(off)[meas[idm]]
        ri = np.stack([xi[o], xi[o + 1], xi[o + 2]])
        mi = np.stack([np.cos(xi[o + 4]) * np.sin(xi[o + 3]), -np.sin(xi[o + 4]), np.cos(xi[o + 4]) * np.cos(xi[o + 3])])
        hc = rot @ ((ri - xi[0:3, None]) * xi[o + 5] + mi)
        h_image = cam[0] * np.stack([hc[0] / hc[2], hc[1] / hc[2]]) + np.array([[cam[1]], [cam[2]]])
        nu = z_meas[idm].T - distort(h_image, cam)
        r_ = np.sqrt(nu[0] ** 2 + nu[1] ** 2)
        res[idm] = r_
        mask[idm] = r_ < (r_.min() + threshold)
        total += mask[idm].sum()
    eum = np.nonzero(tm == CARTESIAN)[0]
    if len(eum):
        o = np.asarray(off)[meas[eum]]
        xyz = np.stack([xi[o], xi[o + 1], xi[o + 2]])
        hc = rot @ (xyz - xi[0:3, None])
        h_image = cam[0] * np.stack([hc[0] / hc[2], hc[1] / hc[2]]) + np.array([[cam[1]], [cam[2]]])
        nu = z_meas[eum].T - distort(h_image, cam)
        r_ = np.sqrt(nu[0] ** 2 + nu[1] ** 2)
        res[eum] = r_
        mask[eum] = r_ < threshold
        total += mask[eum].sum()
    return int(total), mask.astype(np.int32), res


def hypothesis_state(sel, types, off, x, P, Hc, Hl, z, h):   # ransac_hypotheses.m:51-63
    n = x.shape[0]
    c = _sparse_cols(types, off, sel)
    Hi = _rows(n, types, off, sel, Hc, Hl)[:, c]
    PHt = P[:, c] @ Hi.T
    S = Hi @ PHt[c, :] + np.eye(2 * len(sel))
    K = PHt @ np.linalg.inv(S)
    return x + K @ (z[sel].ravel() - h[sel].ravel())


def ransac(types, off, x, P, Hc, Hl, z, h, ic_list, meas, cam, hyp, threshold, early_exit=True):   # ransac_hypotheses.m:27-85
    n_draw, k = hyp.shape
    m = len(meas)
    sup = -np.ones(n_draw, np.int32)
    li = np.zeros(m, np.int32)
    n_hyp, max_support, best, iters = 1000, 0, -1, 0
    limit = min(n_draw, 1000) if early_exit else n_draw
    zm = z[meas]
    for it in range(limit):
        if early_exit and n_hyp == 0:
            break
        sel = [ic_list[p] for p in hyp[it]]
        xi = hypothesis_state(sel, types, off, x, P, Hc, Hl, z, h)
        s, mask, _ = support(meas, types, off, xi, cam, zm, threshold)
        sup[it] = s
        iters += 1
        if s > max_support:
            max_support, best, li = s, it, mask
            epsilon = 1 - s / len(ic_list)
            with np.errstate(divide="ignore"):
                n_hyp = int(np.ceil(np.log(1 - 0.99) / np.log(1 - (1 - epsilon))))
        if early_exit and n_hyp <= k:
            break
    return dict(support=sup, li_mask=li, best=best, iters=iters, n_hyp=n_hyp, max_support=max_support)


def rescue(types, off, P, Hc, Hl, h, z, ic, li, chi2=5.9915):   # @ekf_filter/rescue_hi_inliers.m:35-46
    N = len(types)
    n = P.shape[0]
    hi = np.zeros(N, np.int32)
    for i in range(N):
        if ic[i] == 1 and li[i] == 0:
            c = _sparse_cols(types, off, [i])
            Hi = _rows(n, types, off, [i], Hc, Hl)[:, c]
            Si = Hi @ P[np.ix_(c, c)] @ Hi.T
            nui = z[i] - h[i]
            try:
                hi[i] = 1 if nui @ np.linalg.inv(Si) @ nui < chi2 else 0
            except np.linalg.LinAlgError:
                # a measured landmark the projection skipped has H = 0, so Si = 0: MATLAB's inv warns and returns Inf, the quadratic form is
                # Inf or NaN and `< chi2` is false (the C oracle's inv2 and the device's 1/det behave the same way)
                hi[i] = 0
    return hi


def step(types, off, cam, x_kk, P_kk, u, meas_idx, z_meas, hyp, threshold, early_exit=True, chi2=5.9915):
    """mono_slam.m:153-187 ('1PRE')."""
    N = len(types)
    meas_idx = np.asarray(meas_idx, np.int64)
    x1, P1 = predict(x_kk, P_kk, u)
    h, has_h = project(types, off, x1, cam)
    Hc, Hl = jacobian(types, off, x1, cam, h, has_h)
    S = innovation(types, off, P1, Hc, Hl, has_h)
    z = np.zeros((N, 2))
    z[meas_idx] = z_meas
    ic = np.zeros(N, np.int32)
    ic[meas_idx] = 1
    li = np.zeros(N, np.int32)
    out = dict(x_km1=x1, P_km1=P1, S=S)
    if len(meas_idx) >= hyp.shape[1] and len(meas_idx) > 0:
        r = ransac(types, off, x1, P1, Hc, Hl, z, h, meas_idx, meas_idx, cam, hyp, threshold, early_exit)
        li[meas_idx] = r["li_mask"]
        out["ransac"] = r
    x2, P2 = update_landmarks(types, off, np.nonzero(li)[0], x1, P1, Hc, Hl, z, h)
    h2, has2 = project(types, off, x2, cam, h, has_h)
    Hc2, Hl2 = jacobian(types, off, x2, cam, h2, has2)
    hi = rescue(types, off, P2, Hc2, Hl2, h2, z, ic, li, chi2)
    x3, P3 = update_landmarks(types, off, np.nonzero(hi)[0], x2, P2, Hc2, Hl2, z, h2)
    out.update(x_kk=x3, P_kk=P3, li=li[meas_idx], hi=hi[meas_idx])
    return out


# ---- a10 / a11 ------------------------------------------------------------------------------------------------
def siftmatch(L1, L2, thresh=1.5):    # sift/siftmatch.c:83-132 (vectorised distances; exact for integer classes)
    L1, L2 = np.asarray(L1), np.asarray(L2)
    acc_t = {np.dtype(np.float64): np.float64, np.dtype(np.float32): np.float32}.get(L1.dtype, np.int64)
    a, b = L1.astype(acc_t).T, L2.astype(acc_t).T
    pairs, scores = [], []
    th = np.float32(thresh)
    for k1 in range(a.shape[0]):
        d = ((a[k1][None, :] - b) ** 2).sum(1)
        if len(d) == 0:
            continue
        order = np.argsort(d, kind="stable")
        best = d[order[0]]
        second = d[order[1]] if len(d) > 1 else (np.inf if acc_t != np.int64 else 0x7fffffff)
        if th * np.float32(best) <= np.float32(second):
            pairs.append((k1 + 1, order[0] + 1))
            scores.append(float(best))
    return np.array(pairs, float).reshape(-1, 2).T, np.array(scores)


def knn(data, query, k):              # kNearestNeighbors.m:29-39
    ids = np.zeros((query.shape[0], k))
    dist = np.zeros((query.shape[0], k))
    for i in range(query.shape[0]):
        d = ((query[i][None, :] - data) ** 2).sum(1)
        order = np.argsort(d, kind="stable")
        ids[i] = order[:k] + 1
        dist[i] = np.sqrt(d[order[:k]])
    return ids, dist


# ---- map management (independent restatement; unpinned)
def undistort(uvd, cam):                      # undistort_fm_my_version.m:27-48
    f, Cx, Cy, k1, k2 = cam[:5]
    xd, yd = (uvd[0] - Cx) / f, (uvd[1] - Cy) / f
    rd = np.sqrt(xd * xd + yd * yd)
    ru = rd / (1 + k1 * rd ** 2 + k2 * rd ** 4)
    for _ in range(10):
        ru = ru - (ru + k1 * ru ** 3 + k2 * ru ** 5 - rd) / (1 + 3 * k1 * ru ** 2 + 5 * k2 * ru ** 4)
    D = 1 + k1 * ru ** 2 + k2 * ru ** 4
    return np.array([f * xd / D + Cx, f * yd / D + Cy])


def map_delete(types, off, x, P, del_idx):    # delete_features.m:54-74 / delete_a_feature.m:47-51, from the back
    x, P, types = x.copy(), P.copy(), list(types)
    offs = list(off)
    for i in sorted(del_idx, reverse=True):
        d = 6 if types[i] == INVDEPTH else 3
        o = offs[i]
        x = np.concatenate([x[:o], x[o + d:]])
        P = np.hstack([P[:, :o], P[:, o + d:]])
        P = np.vstack([P[:o, :], P[o + d:, :]])
        del types[i]
        offs = [13 + sum(6 if t == INVDEPTH else 3 for t in types[:k]) for k in range(len(types))]
    return x, P, np.array(types, np.int32)


def map_add(x, P, cam, uvd, std_pxl, initial_rho):   # add_features_inverse_depth.m / add_a_feature_covariance_inverse_depth.m
    f, Cx, Cy = cam[0], cam[1], cam[2]
    X, PR = x.copy(), P.copy()
    Xv = x[:13]
    uvd = np.asarray(uvd, float).reshape(-1, 2)
    rho0 = np.broadcast_to(initial_rho, (len(uvd),))
    for k in range(len(uvd)):
        uv = undistort(uvd[k], cam)
        R = q2r(Xv[3:7])
        hc = np.array([-(Cx - uv[0]) / f, -(Cy - uv[1]) / f, 1.0])
        nw = R @ hc
        X = np.concatenate([X, Xv[:3], [np.arctan2(nw[0], nw[2]), np.arctan2(-nw[1], np.hypot(nw[0], nw[2])), rho0[k]]])
        std_rho = rho0[k] ** 2 * 0.01
        Xw, Yw, Zw = nw
        dth = np.array([Zw / (Xw ** 2 + Zw ** 2), 0, -Xw / (Xw ** 2 + Zw ** 2)])
        s2, sxz = Xw ** 2 + Yw ** 2 + Zw ** 2, np.sqrt(Xw ** 2 + Zw ** 2)
        dph = np.array([Xw * Yw / (s2 * sxz), -sxz / s2, Zw * Yw / (s2 * sxz)])
        dg_dq = dRq_times_a_by_dq(Xv[3:7], hc)
        dy_dxv = np.zeros((6, 13))
        dy_dxv[:3, :3] = np.eye(3)
        dy_dxv[3, 3:7] = dth @ dg_dq
        dy_dxv[4, 3:7] = dph @ dg_dq
        dyp_dgw = np.vstack([np.zeros((3, 3)), dth, dph])
        dgc_dhu = np.array([[1 / f, 0], [0, 1 / f], [0, 0]])
        dhu_dhd = np.linalg.inv(jacob_distor(cam, uvd[k]))
        dy_dhd = np.zeros((6, 3))
        dy_dhd[:5, :2] = dyp_dgw @ R @ dgc_dhu @ dhu_dhd
        dy_dhd[5, 2] = 1
        Padd = np.diag([std_pxl ** 2, std_pxl ** 2, std_rho ** 2])
        P_xv, P_yxv, P_y, P_xvy = PR[:13, :13], PR[13:, :13], PR[13:, 13:], PR[:13, 13:]
        PR = np.block([[P_xv, P_xvy, P_xv @ dy_dxv.T], [P_yxv, P_y, P_yxv @ dy_dxv.T],
                       [dy_dxv @ P_xv, dy_dxv @ P_xvy, dy_dxv @ P_xv @ dy_dxv.T + dy_dhd @ Padd @ dy_dhd.T]])
    return X, PR


def map_convert(types, x, P, threshold=0.1):  # inversedepth_2_cartesian.m:27-76
    X, PR, types = x.copy(), P.copy(), np.array(types, np.int32).copy()
    conv = np.zeros(len(types), np.int32)
    for i in range(len(types)):
        if types[i] != INVDEPTH:
            continue
        o = 13 + sum(6 if t == INVDEPTH else 3 for t in types[:i])
        rho, theta, phi = X[o + 5], X[o + 3], X[o + 4]
        std_d = np.sqrt(PR[o + 5, o + 5]) / rho ** 2
        mi = m_dir(theta, phi)
        p = X[o:o + 3] + mi / rho
        a, c2 = p - X[o:o + 3], p - X[0:3]
        li = 4 * std_d * (a @ c2) / (np.linalg.norm(a) * np.linalg.norm(c2)) / np.linalg.norm(c2)
        if li < threshold:
            n0 = X.shape[0]
            J = np.hstack([np.eye(3), (np.array([np.cos(phi) * np.cos(theta), 0, -np.cos(phi) * np.sin(theta)]) / rho)[:, None],
                           (np.array([-np.sin(phi) * np.sin(theta), -np.cos(phi), -np.sin(phi) * np.cos(theta)]) / rho)[:, None], (-mi / rho ** 2)[:, None]])
            Jall = np.zeros((n0 - 3, n0))
            Jall[:o, :o] = np.eye(o)
            Jall[o:o + 3, o:o + 6] = J
            Jall[o + 3:, o + 6:] = np.eye(n0 - o - 6)
            X = np.concatenate([X[:o], p, X[o + 6:]])
            PR = Jall @ PR @ Jall.T
            types[i] = CARTESIAN
            conv[i] = 1
    return X, PR, types, conv


# ---- VO front end (code_from_dr_ye/), LAPACK svd like MATLAB's ----------------------------------------
def vo_find_transform(pset1, pset2):              # find_transform_matrix_dr_ye.m:8-41 (3 x pnum arrays)
    pnum = pset2.shape[1]
    ct1, ct2 = pset1.sum(1) / pnum, pset2.sum(1) / pnum
    H = np.zeros((3, 3))
    for i in range(pnum):
        H = H + np.outer(pset2[:, i] - ct2, pset1[:, i] - ct1)
    U, S, Vt = np.linalg.svd(H)
    V = Vt.T
    Xq = V @ U.T
    mdet = np.linalg.det(Xq)
    if np.round(mdet) == 1:
        return Xq, ct1 - Xq @ ct2, 1
    if np.round(mdet) == -1:
        zn = np.nonzero(np.abs(S) < 0.00000000000001)[0]
        if zn.size == 1:
            V[:, zn] = -V[:, zn]
            rot = V @ U.T
            return rot, ct1 - rot @ ct2, 2
        return H, np.zeros(3), -1
    return H, np.zeros(3), 0


def vo_dist(pset2):                               # ransac_dr_ye.m:20-23
    nr = np.sqrt(pset2[2] ** 2 + pset2[1] ** 2 + pset2[0] ** 2)
    minZ = pset2[2, nr > 0.4].min()
    k = np.nonzero(pset2[2] == minZ)[0][0]
    return np.sqrt((pset2[:, k] ** 2).sum())


def vo_ransac(pset1, pset2, draws):               # vodometry_dr_ye.m:162-236 over ransac_dr_ye.m:48-72
    pnum = pset1.shape[1]
    dist = vo_dist(pset2)
    cn, maxC, nIter = [], 0, float(len(draws))
    for d in draws:
        rot, tr, _ = vo_find_transform(pset1[:, d], pset2[:, d])
        dd = ((rot @ pset2 + tr[:, None] - pset1) ** 2).sum(0)
        c = int((dd < 0.001 * dist).sum())
        cn.append(c)
        if c > maxC:
            maxC = c
            with np.errstate(divide="ignore"):
                nIter = 5 * np.ceil(np.log(0.01) / np.log(1 - (maxC / pnum) ** 4))
    cn = np.array(cn)
    best = int(np.argmax(cn))
    out = dict(cnum=cn, best=best, n_iterations=int(min(len(draws), nIter)), dist=dist)
    if cn[best] < 3:
        out.update(sta=4, n_support=int(cn[best]), inliers=np.zeros(pnum, np.int32))
        return out
    rot, tr, _ = vo_find_transform(pset1[:, draws[best]], pset2[:, draws[best]])
    inl = ((rot @ pset2 + tr[:, None] - pset1) ** 2).sum(0) < 0.001 * dist
    o1, o2 = pset1[:, inl], pset2[:, inl]
    rot, tr, sta = vo_find_transform(o1, o2)
    en = np.sqrt(((rot @ o2 + tr[:, None] - o1) ** 2).sum(0))
    out.update(sta=sta, n_support=int(inl.sum()), inliers=inl.astype(np.int32), rot=rot, trans=tr, error_mean=en.mean(),
               error_std=en.std(ddof=1) if en.size > 1 else 0.0)
    if sta >= 1:
        out["euler"] = np.array([np.arctan2(rot[2, 1], rot[2, 2]), np.arcsin(-rot[2, 0]), np.arctan2(rot[1, 0], rot[0, 0])])
    return out
