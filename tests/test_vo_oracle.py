"""CPU: the VO RANSAC restatement.  The reference ships no VO golden data, so the oracle is pinned to
LAPACK's svd (the numpy twin calls the routine MATLAB calls) and to rigid-motion known answers."""
import importlib

import numpy as np


def rotm(a):
    th = np.linalg.norm(a)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def scene(pnum, seed, outliers=0.3, sigma=0.002):
    rng = np.random.default_rng(seed)
    R, T = rotm(rng.normal(0, 0.1, 3)), rng.normal(0, 0.05, 3)
    p2 = np.stack([rng.uniform(-1.5, 1.5, pnum), rng.uniform(-1, 1, pnum), rng.uniform(0.6, 5, pnum)])
    p1 = R @ p2 + T[:, None] + rng.normal(0, sigma, (3, pnum))
    bad = rng.choice(pnum, int(outliers * pnum), replace=False)
    p1[:, bad] += rng.normal(0, 0.5, (3, len(bad)))
    match = np.stack([rng.permutation(pnum * 2)[:pnum] + 1, rng.permutation(pnum * 2)[:pnum] + 1])
    return rng, R, T, p1, p2, match, bad


def test_find_transform_against_lapack_svd(orc):
    from oracle import np_twin as tw
    rng = np.random.default_rng(0)
    worst = 0
    for t in range(400):
        n = 4 if t % 2 == 0 else int(rng.integers(4, 40))
        R, T = rotm(rng.normal(0, 0.3, 3)), rng.normal(0, 0.1, 3)
        p2 = rng.normal(0, 1, (3, n)) + np.array([[0], [0], [2.5]])
        p1 = R @ p2 + T[:, None] + rng.normal(0, 0.003, (3, n))
        r1, t1, s1 = orc.vo_find_transform(p1, p2)
        r2, t2, s2 = tw.vo_find_transform(p1, p2)
        assert s1 == s2
        worst = max(worst, np.abs(r1 - r2).max(), np.abs(t1 - t2).max())
    assert worst < 1e-12


def test_find_transform_known_answers(orc):
    rng = np.random.default_rng(1)
    R, T = rotm(np.array([0.2, -0.1, 0.3])), np.array([0.1, -0.2, 0.05])
    p2 = rng.normal(0, 1, (3, 12)) + np.array([[0], [0], [3.0]])
    rot, tr, st = orc.vo_find_transform(R @ p2 + T[:, None], p2)             # noise-free: exact recovery
    assert st == 1 and np.abs(rot - R).max() < 1e-14 and np.abs(tr - T).max() < 1e-14
    flat = np.array([[0, 1, 0, 1.0], [0, 0, 1, 1], [2, 2, 2, 2]])             # exactly co-planar: still the rotation
    rot, tr, st = orc.vo_find_transform(R @ flat + T[:, None], flat)
    assert st in (1, 2) and np.abs(rot - R).max() < 1e-14
    refl = np.diag([1.0, 1.0, -1.0])                                          # a reflection is not a solution (state -1: rot = H, trans = 0)
    rot, tr, st = orc.vo_find_transform(refl @ p2, p2)
    assert st == -1 and not tr.any()
    q = orc.R2q(R)
    assert abs(np.linalg.norm(q) - 1) < 1e-15 and q[0] > 0
    assert np.allclose(orc.R2q(np.eye(3)), [1, 0, 0, 0])


def test_ransac_against_twin_and_truth(orc):
    from oracle import np_twin as tw
    vo = importlib.import_module("3pre_amd.vo")
    for seed, pnum in ((3, 150), (4, 40), (5, 9)):
        rng, R, T, p1, p2, match, bad = scene(pnum, seed)
        draws = vo.draw_hypotheses(match, vo.vo_rst(pnum), rng)
        a, b = orc.vo_ransac(p1, p2, draws), tw.vo_ransac(p1, p2, draws)
        assert np.array_equal(a["cnum"], b["cnum"]) and a["best"] == b["best"] and a["n_iterations"] == b["n_iterations"]
        assert a["sta"] == b["sta"] == 1 and a["n_support"] == b["n_support"] and np.array_equal(a["inliers"], b["inliers"])
        assert np.abs(a["rot"] - b["rot"]).max() < 1e-13 and np.abs(a["trans"] - b["trans"]).max() < 1e-13
        assert abs(a["error_mean"] - b["error_mean"]) < 1e-14 and abs(a["error_std"] - b["error_std"]) < 1e-14
        assert np.abs(a["euler"] - b["euler"]).max() < 1e-13
        assert np.abs(a["rot"] - R).max() < 5e-3 and np.abs(a["trans"] - T).max() < 1e-2
        assert not a["inliers"][bad].all()


def test_draw_rule_and_rst():
    vo = importlib.import_module("3pre_amd.vo")
    assert vo.vo_rst(4) == 1 and vo.vo_rst(6) == 15 and vo.vo_rst(200) == 700
    rng = np.random.default_rng(7)
    match = np.stack([rng.integers(1, 30, 60), rng.integers(1, 30, 60)])       # many shared keypoints
    d = vo.draw_hypotheses(match, 300, rng)
    assert d.min() >= 0 and d.max() < 60
    for r in d:
        assert len(set(r)) == 4
        assert len({match[0, r[0]], match[0, r[1]], match[0, r[2]]}) == 3 and len({match[1, r[0]], match[1, r[1]], match[1, r[2]]}) == 3
        assert match[0, r[3]] not in (match[0, r[0]], match[0, r[2]]) and match[1, r[3]] not in (match[1, r[1]], match[1, r[2]])


def test_gather_rounding(orc):
    rng = np.random.default_rng(9)
    x, y, z = rng.normal(0, 1, (144, 176)), rng.normal(0, 1, (144, 176)), rng.uniform(0.5, 5, (144, 176))
    frm = np.stack([rng.uniform(1, 176, 50), rng.uniform(1, 144, 50), rng.uniform(1, 3, 50), rng.uniform(-3, 3, 50)])
    frm[0, 0], frm[1, 0] = 10.5, 20.5                                          # MATLAB round(): half away from zero -> (11, 21)
    p = orc.vo_gather(x, y, z, frm, np.arange(1, 51.0))
    assert p[0, 0] == -x[20, 10] and p[1, 0] == -y[20, 10] and p[2, 0] == z[20, 10]
    for i in range(1, 50):
        c, r = int(np.floor(frm[0, i] + 0.5)), int(np.floor(frm[1, i] + 0.5))
        assert p[2, i] == z[r - 1, c - 1]
