"""GPU: a whole frame loop through every row of SURVEY 8 -- VO RANSAC gives u (8(f)-4), prediction, IC search on SIFT-like
descriptors (8(f)-2), RANSAC, LI update, rescue, HI update (8(a)), then map management (8(f)-1) -- against the oracle doing the
same on the host, frame by frame.  fp64: the inlier sets and the measurement lists must be identical, the state within 1e-9."""
import importlib

import numpy as np
import pytest

from test_vo_oracle import rotm

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")
vo = importlib.import_module("3pre_amd.vo")


def test_frame_loop_all_rows(pre3, orc):
    N0, frames = 48, 3
    seq = synth.make_sequence(N0, frames, 30, seed=211, sigma_z=0.4)
    cam = seq["cam"]
    rng = np.random.default_rng(17)
    bank = np.abs(rng.normal(0, 1, (128, N0)))
    bank /= np.linalg.norm(bank, axis=0)
    types = np.zeros(N0, np.int32)
    f = pre3.EkfFilter(cam, types, dtype="f64", max_hyp=30, max_landmarks=N0 + 4)
    x, P = seq["x0"].copy(), seq["P0"].copy()
    f.set_x_p_k_k(x, P)
    f.set_descriptors(bank)
    ids = list(range(N0))
    for k, s in enumerate(seq["steps"]):
        # --- VO: two synthetic range frames related by the step's odometry; its u drives the prediction (fv.m:44-47)
        q = s["u"][3:7]
        R = np.array([[1 - 2 * (q[2] ** 2 + q[3] ** 2), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2])],
                      [2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] ** 2 + q[3] ** 2), 2 * (q[2] * q[3] - q[0] * q[1])],
                      [2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2)]])
        p2 = np.stack([rng.uniform(-1.5, 1.5, 80), rng.uniform(-1, 1, 80), rng.uniform(0.6, 5, 80)])
        p1 = R @ p2 + s["u"][0:3, None] + rng.normal(0, 0.001, (3, 80))
        p1[:, :16] += rng.normal(0, 0.4, (3, 16))
        match = np.stack([np.arange(1, 81), rng.permutation(80) + 1])
        draws = vo.draw_hypotheses(match, vo.vo_rst(80), rng)
        g_vo, o_vo = vo.vo_ransac(p1, p2, draws), orc.vo_ransac(p1, p2, draws)
        assert g_vo["sta"] == o_vo["sta"] == 1 and np.array_equal(g_vo["inliers"], o_vo["inliers"])
        u = np.r_[o_vo["trans"], orc.R2q(o_vo["rot"])]
        assert np.abs(g_vo["u"] - u).max() < 1e-12
        # --- prediction + IC search on the frame's SIFT set
        to, off, n = orc.landmark_table(types)
        f.ekf_prediction(u)
        x1, P1 = orc.predict(x, P, u)
        h, has_h = orc.project(to, off, x1, cam)
        desc, pos = [], []
        truth = {int(i): zz for i, zz in zip(s["meas_idx"], s["z"])}
        for j in range(len(types)):
            if has_h[j] and (ids[j] in truth or ids[j] < 0) and rng.uniform() < 0.9:
                desc.append(bank[:, j] + rng.normal(0, 0.01, 128))
                pos.append(np.r_[truth[ids[j]] if ids[j] in truth else h[j] + rng.normal(0, 1.0, 2), 2.0, 0.0])
        for _ in range(25):
            d = np.abs(rng.normal(0, 1, 128)); desc.append(d / np.linalg.norm(d)); pos.append([rng.uniform(1, 176), rng.uniform(1, 144), 2.0, 0.0])
        sd, sp = np.array(desc).T.copy(), np.array(pos).T.copy()
        f.load_scan(sd, sp)
        g_ic = f.matching_sift_based(1.5, strict_reference=True)
        o_ic = orc.ic_search(to, off, x1, P1, cam, bank, sd, sp, 1.5, True)
        assert np.array_equal(g_ic["meas_idx"], o_ic["meas_idx"]) and np.array_equal(g_ic["z"], o_ic["z"]) and len(g_ic["meas_idx"]) >= 8
        bank = o_ic["bank"]
        # --- RANSAC + updates
        m = len(o_ic["meas_idx"])
        hyp = np.stack([rng.permutation(m)[:3] for _ in range(30)]).astype(np.int32)
        f.ransac_hypotheses(hyp, threshold=1.0)
        f.ekf_update_li_inliers(); f.rescue_hi_inliers(); f.ekf_update_hi_inliers()
        ref = orc.step(to, off, cam, x, P, u, o_ic["meas_idx"], o_ic["z"], hyp, 1.0)
        li, hi = f.get_flags()
        assert np.array_equal(li, ref["li"]) and np.array_equal(hi, ref["hi"]), "frame %d" % k
        x, P = ref["x_kk"], ref["P_kk"]
        assert np.abs(f.get_x_k_k() - x).max() < 1e-9 and np.abs(f.get_p_k_k() - P).max() < 1e-9 * np.abs(P).max(), "frame %d" % k
        # --- map management: drop one landmark, convert, add one (with a descriptor)
        d = [int(rng.integers(0, len(types)))]
        f.delete_features(d)
        x, P, types = orc.map_delete(to, off, x, P, d)
        ids.pop(d[0]); bank = np.delete(bank, d[0], axis=1)
        conv = f.inversedepth_2_cartesian(0.5)
        x, P, types, c2 = orc.map_convert(types, x, P, 0.5)
        assert np.array_equal(conv, c2)
        uvd = np.array([[rng.uniform(30, 140), rng.uniform(30, 110)]])
        f.add_features_inverse_depth(uvd, 1.0, 0.5)
        x, P = orc.map_add(x, P, cam, uvd, 1.0, 0.5)
        types = np.r_[types, 0].astype(np.int32); ids.append(-1)
        nd = np.abs(rng.normal(0, 1, (128, 1))); nd /= np.linalg.norm(nd)
        f.set_descriptors(nd, first=f.N - 1); bank = np.concatenate([bank, nd], 1)
        assert np.array_equal(f.get_descriptors(), bank) and np.array_equal(f.lm_type, types)
        assert np.abs(f.get_p_k_k() - P).max() < 1e-9 * np.abs(P).max()
    f.close()
