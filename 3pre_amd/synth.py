"""Deterministic synthetic EKF-SLAM sequences of a named state dimension.

This module only SYNTHESISES INPUTS (landmarks, covariance, odometry, pixels, hypothesis draws) with numpy;
it is used by bench.py and the tests.  It contains no part of the filter.

Camera = the SR4000 intrinsics of the reference's snapshot (f=250.57731, Cx=90, Cy=70, k1=-0.84656,
k2=0.53701, 176x144).  Landmarks are inverse-depth (the reference's fixture has only those).
"""
import numpy as np

CAM = np.array([250.57731, 90.0, 70.0, -0.84656, 0.53701, 144.0, 176.0])   # f Cx Cy k1 k2 nRows nCols

# The headline workload of bench.py (and of the parity test that pins it, tests/test_gpu_fullsize.py): the reference's own RANSAC threshold
# (ransac_hypotheses.m:33: threshold = std_z = mono_slam.m:78's sigma_image_noise = 1 px) on a sequence whose truth leaves the odometry by
# 2.5 times the process noise the filter assumes.  The 3-point hypothesis states are then off by a pixel or so, some true inliers miss the
# low-innovation set of the winning hypothesis and come back through rescue_hi_inliers.m:29-47 once the LI update has pinned the pose: with
# motion_noise = 0.5 (rounds 1-3) the rescue finds 0.5 rows per step at this threshold; with 1.5 / 2.0 / 2.5 it finds 12 / 21 / 31 rows per step
# over 60 steps (LI 612 / 583 / 554 rows of the 640 possible; oracle/np_twin.py, N = 500), 36 rows in steps 5..25.
HEADLINE = dict(threshold=1.0, motion_noise=2.5)


def q2r(q):
    r, x, y, z = q
    return np.array([[r * r + x * x - y * y - z * z, 2 * (x * y - r * z), 2 * (z * x + r * y)],
                     [2 * (x * y + r * z), r * r - x * x + y * y - z * z, 2 * (y * z - r * x)],
                     [2 * (z * x - r * y), 2 * (y * z + r * x), r * r - x * x - y * y + z * z]])


def qprod(q1, q2):
    a, b, c, d = q1
    w, x, y, z = q2
    return np.array([a * w - b * x - c * y - d * z, a * x + b * w + c * z - d * y, a * y - b * z + c * w + d * x, a * z + b * y - c * x + d * w])


def pixels(xv, Y, cam=CAM):
    """Distorted pixel of every inverse-depth landmark Y (N x 6) seen from pose xv = [r(3); q(4)].
    Returns (uv (N x 2), visible mask) using the reference's visibility rules."""
    f, Cx, Cy, k1, k2, nRows, nCols = cam
    R = q2r(xv[3:7])
    cphi = np.cos(Y[:, 4])
    m = np.stack([cphi * np.sin(Y[:, 3]), -np.sin(Y[:, 4]), cphi * np.cos(Y[:, 3])], 1)
    v = (Y[:, 0:3] - xv[0:3]) * Y[:, 5:6] + m
    hc = v @ R                                # rows: R' v
    ax = np.degrees(np.arctan2(hc[:, 0], hc[:, 2]))
    ay = np.degrees(np.arctan2(hc[:, 1], hc[:, 2]))
    xu, yu = hc[:, 0] / hc[:, 2], hc[:, 1] / hc[:, 2]
    r2 = xu * xu + yu * yu
    D = 1 + k1 * r2 + k2 * r2 * r2
    uv = np.stack([xu * D * f + Cx, yu * D * f + Cy], 1)
    vis = (np.abs(ax) <= 60) & (np.abs(ay) <= 60) & (uv[:, 0] > 0) & (uv[:, 0] < nCols) & (uv[:, 1] > 0) & (uv[:, 1] < nRows)
    return uv, vis


def make_map(N, seed=None):
    """Initial estimate x0 (13+6N), covariance P0, and the hidden truth drawn consistently with P0."""
    rng = np.random.default_rng(1000 + N if seed is None else seed)
    n = 13 + 6 * N
    x0 = np.zeros(n)
    x0[3] = 1.0
    # rays through the (undistorted) image plane, depth 0.5..5 m, anchors near the origin
    xu = rng.uniform(-0.30, 0.30, N)
    yu = rng.uniform(-0.24, 0.24, N)
    d = rng.uniform(0.5, 5.0, N)
    anchor = rng.normal(0, 0.05, (N, 3))
    ray = np.stack([xu, yu, np.ones(N)], 1)
    ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    theta = np.arctan2(ray[:, 0], ray[:, 2])
    phi = np.arctan2(-ray[:, 1], np.hypot(ray[:, 0], ray[:, 2]))      # hinv_my_version.m:40-51 convention
    Y = np.concatenate([anchor, theta[:, None], phi[:, None], (1.0 / d)[:, None]], 1)
    x0[13:] = Y.ravel()
    rngP = np.random.default_rng(2000 + N if seed is None else seed + 1)
    A = rngP.standard_normal((n, 32))
    # per-state scales: pose tight, landmark position 2 cm, angles 0.3 deg, inverse depth 2 %
    s = np.empty(n)
    s[0:3] = 0.01; s[3:7] = 0.002; s[7:13] = 0.01
    sl = np.array([0.02, 0.02, 0.02, 0.005, 0.005, 0.02])
    s[13:] = np.tile(sl, N)
    A = A * s[:, None] / np.sqrt(32.0)
    P0 = A @ A.T                                   # in place from here on: at N=2000 every extra n x n temporary is 1.15 GB
    P0[np.diag_indices(n)] += (0.1 * s) ** 2
    iu = np.triu_indices(n, 1, m=n) if n <= 600 else None
    if iu is not None:
        P0[iu] = P0.T[iu]                          # exact symmetry (mirror the lower triangle)
    else:
        for i0 in range(0, n, 1024):               # blockwise mirror: no n x n index arrays
            i1 = min(i0 + 1024, n)
            P0[i0:i1, i1:] = P0[i1:, i0:i1].T
            blk = P0[i0:i1, i0:i1]
            blk[np.triu_indices(i1 - i0, 1)] = blk.T[np.triu_indices(i1 - i0, 1)]
    xi = rngP.standard_normal(32)
    x_true = x0 + 0.7 * (A @ xi) + 0.07 * s * rngP.standard_normal(n)
    x_true[7:13] = 0
    x_true[3:7] /= np.linalg.norm(x_true[3:7])
    return x0, P0, x_true


def draw_hypotheses(rng, m, n_hyp, k=3):
    """RANSAC draws for m individually compatible measurements: k distinct positions per hypothesis when more than k are there, ONE
    otherwise (select_random_match.m:47-51 takes 3 landmarks if #IC > 3, else 1); with no measurement at all the table still has one
    column (of zeros: pre3_step wants n_draw >= 1, k >= 1 and then skips RANSAC because m < k).  A long synthetic sequence walks away
    from its map and ends up here -- a table with fewer than one column is what pre3_step rejects as "bad hypothesis table"."""
    kk = k if m > k else 1
    if m == 0:
        return np.zeros((n_hyp, 1), np.int32)
    return np.stack([rng.permutation(m)[:kk] for _ in range(n_hyp)]).astype(np.int32)


def make_sequence(N, steps, n_hyp, k=3, meas_frac=0.8, outlier_frac=0.2, sigma_z=0.25, seed=None, motion_noise=0.5):
    """A whole input sequence: per step the odometry u, the measured landmark list, their pixels and the
    RANSAC draws.  The truth moves by u_true = u + noise (motion_noise x the process noise the filter assumes,
    predict_state_and_covariance.m:98-102); pixels come from the truth."""
    x0, P0, x_true = make_map(N, seed)
    rng = np.random.default_rng(3000 + N if seed is None else seed + 2)
    rngh = np.random.default_rng(4000 + N if seed is None else seed + 3)
    pose = x_true[0:7].copy()
    Y_true = x_true[13:].reshape(N, 6)
    seq = []
    m_target = int(round(meas_frac * N))
    for _ in range(steps):
        dX = rng.normal(0, 0.004, 3)
        ang = rng.normal(0, np.radians(0.1), 3)
        dq = np.array([1.0, ang[0] / 2, ang[1] / 2, ang[2] / 2])
        dq /= np.linalg.norm(dq)
        u = np.concatenate([dX, dq])
        # the truth follows the odometry up to the process noise the filter assumes (0.01/3 m, 0.12 deg)
        dXt = dX + rng.normal(0, 0.01 / 3 * motion_noise, 3)
        angt = ang + rng.normal(0, np.radians(0.12) * motion_noise, 3) * np.array([1, 0.1, 1])
        dqt = np.array([1.0, angt[0] / 2, angt[1] / 2, angt[2] / 2])
        dqt /= np.linalg.norm(dqt)
        pose[0:3] = pose[0:3] + q2r(pose[3:7]) @ dXt
        pose[3:7] = qprod(pose[3:7], dqt)
        pose[3:7] /= np.linalg.norm(pose[3:7])
        uv, vis = pixels(pose, Y_true)
        cand = np.nonzero(vis)[0]
        pick = np.sort(rng.choice(cand, size=min(m_target, len(cand)), replace=False)).astype(np.int32)
        z = uv[pick] + rng.normal(0, sigma_z, (len(pick), 2))
        n_out = int(round(outlier_frac * len(pick)))
        out_pos = rng.choice(len(pick), size=n_out, replace=False)
        z[out_pos] = np.stack([rng.uniform(1, 175, n_out), rng.uniform(1, 143, n_out)], 1)
        hyp = draw_hypotheses(rngh, len(pick), n_hyp, k)
        seq.append(dict(u=u, meas_idx=pick, z=z, hyp=hyp, outliers=np.sort(out_pos)))
    return dict(N=N, n=13 + 6 * N, cam=CAM.copy(), x0=x0, P0=P0, x_true0=x_true, steps=seq)
