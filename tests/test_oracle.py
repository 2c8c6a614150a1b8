"""CPU suite, part 1: the oracle against the reference's golden vectors (pins), and against its independent
numpy twin where no reference artefact exists (prediction, Cartesian landmarks, hypothesis loop, kNN)."""
import importlib
import json
import os

import numpy as np
import pytest

from util import GOLDEN

twin = importlib.import_module("oracle.np_twin")


@pytest.fixture(scope="module")
def chain(orc, sr4000):
    """oracle run over the snapshot: km1 quantities, LI update, rescue, HI update"""
    g = sr4000
    types, off, n = orc.landmark_table(np.zeros(g["N"], int))
    assert n == g["n"]
    h0, has0 = orc.project(types, off, g["x_k_km1"], g["cam"])
    Hc0, Hl0 = orc.jacobian(types, off, g["x_k_km1"], g["cam"], h0, has0)
    S = orc.innovation(types, off, g["p_k_km1"], Hc0, Hl0, has0)
    x1, P1 = orc.update_landmarks(types, off, g["li_idx"], g["x_k_km1"], g["p_k_km1"], Hc0, Hl0, g["z"], h0)
    h1, has1 = orc.project(types, off, x1, g["cam"], h0, has0)
    Hc1, Hl1 = orc.jacobian(types, off, x1, g["cam"], h1, has1)
    hi, d2 = orc.rescue(types, off, P1, Hc1, Hl1, h1, g["z"], g["individually_compatible"], g["low_innovation_inlier"])
    x2, P2 = orc.update_landmarks(types, off, np.nonzero(hi)[0], x1, P1, Hc1, Hl1, g["z"], h1)
    return dict(types=types, off=off, h0=h0, has0=has0, Hc0=Hc0, Hl0=Hl0, S=S, x1=x1, P1=P1, h1=h1, Hc1=Hc1, Hl1=Hl1, hi=hi, x2=x2, P2=P2)


def test_pin_innovation_S(chain, sr4000):
    assert chain["has0"].all()
    assert np.abs(chain["S"] - sr4000["S"]).max() < 5e-14          # 6.2e-15


def test_pin_h_and_H_after_li_update(chain, sr4000):
    assert np.abs(chain["h1"] - sr4000["h"]).max() < 5e-13         # 5.7e-14
    assert np.abs(chain["Hc1"] - sr4000["Hcam"]).max() < 2e-12     # 2.3e-13 (entries up to 501)
    assert np.abs(chain["Hl1"] - sr4000["Hlm"]).max() < 2e-12


def test_pin_rescue_set(chain, sr4000):
    assert np.array_equal(np.nonzero(chain["hi"])[0], sr4000["hi_idx"])
    assert list(sr4000["hi_idx"]) == [125]


def test_pin_state_after_both_updates(chain, sr4000):
    assert np.abs(chain["x2"] - sr4000["x_k_k"]).max() < 1e-15
    assert np.abs(chain["P2"] - sr4000["p_k_k"]).max() < 1e-17      # P scale 6.1e-4; observed 2.4e-19


def test_pin_one_point_hypotheses(orc, chain, sr4000):
    g = sr4000
    want = g["low_innovation_inlier"][g["meas_idx"]]
    same = 0
    sups = []
    for i in g["ic_idx"]:
        xi = orc.hypothesis_state([i], chain["types"], chain["off"], g["x_k_km1"], g["p_k_km1"], chain["Hc0"], chain["Hl0"], g["z"], chain["h0"])
        s, mask, _ = orc.support(g["meas_idx"], chain["types"], chain["off"], xi, g["cam"], g["z"][g["meas_idx"]], g["std_z"])
        same += bool((mask == want).all())
        sups.append(s)
    assert same == 39 and sorted(sups) == [39] * 39 + [40]


def test_pin_siftmatch_known_answers(orc):
    box = np.load(os.path.join(GOLDEN, "sift_box.npz"))["descriptors"]
    kat = json.load(open(os.path.join(GOLDEN, "siftmatch_kat.json")))
    for c in kat["cases"]:
        for dt in (np.uint8, np.float64, np.float32):
            L1 = box[c["L1"][0]:c["L1"][1]].T.astype(dt)
            L2 = box[c["L2"][0]:c["L2"][1]].T.astype(dt)
            mt, sc = orc.siftmatch(L1, L2, c["thresh"])
            M = mt.shape[1]
            assert M == c["M"]
            assert int(sum((i + 1) * (1000 * int(mt[0, i]) + int(mt[1, i])) for i in range(M))) == c["checksum"]
            assert sc.sum() == c["sum_best_d2"]
            assert list(mt[:, 0]) == c["first"] and list(mt[:, -1]) == c["last"]


def test_pin_knn_docstring_example(orc):
    k = json.load(open(os.path.join(GOLDEN, "siftmatch_kat.json")))["knn_docstring_example"]
    ids, d = orc.knn(k["data"], k["query"], k["k"])
    assert ids.tolist() == k["neighbors"]
    assert np.allclose(np.round(d, 4), k["distances"])


# ---------------------------------------------------------------- unpinned parts: C oracle vs numpy twin
def _mixed_problem(seed=3, N=24):
    rng = np.random.default_rng(seed)
    synth = importlib.import_module("3pre_amd.synth")
    x0, P0, _ = synth.make_map(N, seed=100 + seed)
    types = np.zeros(N, np.int32)
    types[rng.choice(N, N // 3, replace=False)] = 1
    # convert the chosen landmarks to Cartesian points (inversedepth_2_cartesian.m:40-45 formula)
    xs = [x0[:13]]
    for i in range(N):
        y = x0[13 + 6 * i:19 + 6 * i]
        if types[i] == 1:
            m = np.array([np.cos(y[4]) * np.sin(y[3]), -np.sin(y[4]), np.cos(y[4]) * np.cos(y[3])])
            xs.append(y[0:3] + m / y[5])
        else:
            xs.append(y)
    x = np.concatenate(xs)
    n = x.shape[0]
    A = rng.standard_normal((n, 12)) * 0.01
    P = A @ A.T + 1e-6 * np.eye(n)
    return types, x, P, synth.CAM.copy()


def test_twin_predict(orc):
    types, x, P, cam = _mixed_problem()
    u = np.array([0.01, -0.02, 0.015, 0.9999, 0.004, -0.003, 0.002])
    u[3:] /= np.linalg.norm(u[3:])
    xa, Pa = orc.predict(x, P, u)
    xb, Pb = twin.predict(x, P, u)
    assert np.abs(xa - xb).max() < 1e-15
    assert np.abs(Pa - Pb).max() < 1e-15 * max(1.0, np.abs(Pb).max() / 1e-4)
    assert np.allclose(orc.process_noise(), twin.process_noise(), rtol=0, atol=1e-20)
    # structure: only rows/cols 3..6 and the pose block change; velocities zeroed; q normalised
    chg = np.abs(Pa - P) > 0
    chg[3:7, :] = False
    chg[:, 3:7] = False
    chg[:7, :7] = False
    assert not chg.any()
    assert np.all(xa[7:13] == 0) and abs(np.linalg.norm(xa[3:7]) - 1) < 1e-15


def test_twin_mixed_landmark_chain(orc):
    types, x, P, cam = _mixed_problem()
    t, off, n = orc.landmark_table(types)
    ha, hasa = orc.project(t, off, x, cam)
    hb, hasb = twin.project(t, off, x, cam)
    assert np.array_equal(hasa, hasb) and hasa.sum() >= len(types) - 2
    assert np.abs(ha - hb).max() < 1e-11
    Hca, Hla = orc.jacobian(t, off, x, cam, ha, hasa)
    Hcb, Hlb = twin.jacobian(t, off, x, cam, hb, hasb)
    assert np.abs(Hca - Hcb).max() < 1e-9 and np.abs(Hla - Hlb).max() < 1e-9
    Sa = orc.innovation(t, off, P, Hca, Hla, hasa)
    Sb = twin.innovation(t, off, P, Hcb, Hlb, hasb)
    assert np.abs(Sa - Sb).max() < 1e-10
    # finite-difference check of H against h itself (the reference's own commented self-check,
    # calculate_Hi_inverse_depth_my_version.m:59-64): d h / d x
    eps = 1e-7
    for i in np.nonzero(hasa)[0][:8]:
        d = 6 if t[i] == 0 else 3
        for c in list(range(7)) + [off[i] + j for j in range(d)]:
            xp = x.copy()
            xp[c] += eps
            hp, _ = orc.project(t, off, xp, cam)
            col = (hp[i] - ha[i]) / eps
            want = Hca[i][:, c] if c < 7 else Hla[i][:, c - off[i]]
            # Q8: the reference evaluates the distortion Jacobian at the distorted pixel -> approximate only
            assert np.abs(col - want).max() < 0.15 * max(1.0, np.abs(want).max())
    # hypotheses + support with both landmark types
    vis = np.nonzero(hasa)[0]
    rng = np.random.default_rng(0)
    z = np.zeros((len(types), 2))
    z[vis] = ha[vis] + rng.normal(0, 0.5, (len(vis), 2))
    hyp = np.stack([rng.permutation(len(vis))[:3] for _ in range(12)]).astype(np.int32)
    ra = orc.ransac(t, off, x, P, Hca, Hla, z, ha, vis, vis, cam, hyp, 1.0, early_exit=False)
    rb = twin.ransac(t, off, x, P, Hcb, Hlb, z, hb, vis, vis, cam, hyp, 1.0, early_exit=False)
    assert np.array_equal(ra["support"], rb["support"]) and np.array_equal(ra["li_mask"], rb["li_mask"])
    for ee in (True,):
        ra = orc.ransac(t, off, x, P, Hca, Hla, z, ha, vis, vis, cam, hyp, 1.0, early_exit=ee)
        rb = twin.ransac(t, off, x, P, Hcb, Hlb, z, hb, vis, vis, cam, hyp, 1.0, early_exit=ee)
        assert (ra["best"], ra["iters"], ra["n_hyp"], ra["max_support"]) == (rb["best"], rb["iters"], rb["n_hyp"], rb["max_support"])
    # update + rescue
    sel = vis[ra["li_mask"].astype(bool)]
    xa, Pa = orc.update_landmarks(t, off, sel, x, P, Hca, Hla, z, ha)
    xb, Pb = twin.update_landmarks(t, off, sel, x, P, Hcb, Hlb, z, hb)
    assert np.abs(xa - xb).max() < 1e-11 and np.abs(Pa - Pb).max() < 1e-13


def test_twin_on_snapshot(sr4000, chain):
    """the twin reproduces the MATLAB snapshot as well (so both restatements are pinned where pins exist)"""
    g = sr4000
    t, off = chain["types"], chain["off"]
    x1, P1 = twin.update_landmarks(t, off, g["li_idx"], g["x_k_km1"], g["p_k_km1"], chain["Hc0"], chain["Hl0"], g["z"], chain["h0"])
    assert np.abs(P1 - chain["P1"]).max() < 1e-17
    hi = twin.rescue(t, off, P1, chain["Hc1"], chain["Hl1"], chain["h1"], g["z"], g["individually_compatible"], g["low_innovation_inlier"])
    assert np.array_equal(np.nonzero(hi)[0], g["hi_idx"])
    x2, P2 = twin.update_landmarks(t, off, g["hi_idx"], x1, P1, chain["Hc1"], chain["Hl1"], g["z"], chain["h1"])
    assert np.abs(P2 - g["p_k_k"]).max() < 1e-17 and np.abs(x2 - g["x_k_k"]).max() < 1e-15


def test_update_properties(orc):
    types, x, P, cam = _mixed_problem(seed=5, N=10)
    n = x.shape[0]
    # r = 0 is the identity (update.m:50-55)
    xo, Po = orc.update(x, P, np.zeros((0, n)), None, [], [])
    assert np.array_equal(xo, x) and np.array_equal(Po, P)
    # symmetric, PSD-ness preserved, quaternion normalised
    t, off, _ = orc.landmark_table(types)
    h, has = orc.project(t, off, x, cam)
    Hc, Hl = orc.jacobian(t, off, x, cam, h, has)
    sel = np.nonzero(has)[0][:5]
    z = np.zeros((len(types), 2))
    z[sel] = h[sel] + 0.3
    xo, Po = orc.update_landmarks(t, off, sel, x, P, Hc, Hl, z, h)
    assert np.abs(Po - Po.T).max() < 1e-18
    assert np.linalg.eigvalsh(Po).min() > -1e-12
    assert abs(np.linalg.norm(xo[3:7]) - 1) < 1e-15


def test_siftmatch_and_knn_twin(orc):
    rng = np.random.default_rng(11)
    for dt in (np.uint8, np.int8, np.float32, np.float64):
        if dt in (np.uint8, np.int8):
            L1 = rng.integers(0, 120, (128, 37)).astype(dt)
            L2 = rng.integers(0, 120, (128, 53)).astype(dt)
            L2[:, 5] = L1[:, 7]
            L2[:, 9] = L1[:, 7]          # exact tie -> first index, and ratio test fails (second == best)
        else:
            L1 = rng.random((128, 37)).astype(dt)
            L2 = rng.random((128, 53)).astype(dt)
            L2[:, 4] = L1[:, 3] + dt(1e-3)
        ma, sa = orc.siftmatch(L1, L2, 1.5)
        mb, sb = twin.siftmatch(L1, L2, 1.5)
        assert np.array_equal(ma, mb)
        if dt in (np.uint8, np.int8):
            assert np.array_equal(sa, sb)
    # empty / degenerate shapes (siftmatch.c handles K2 = 0 and K2 = 1)
    m0, _ = orc.siftmatch(np.zeros((128, 3)), np.zeros((128, 0)), 1.5)
    assert m0.shape == (2, 0)
    m1, _ = orc.siftmatch(np.ones((4, 2)), np.ones((4, 1)), 1.5)
    assert m1.tolist() == [[1, 2], [1, 1]]       # single candidate: second_best stays +inf -> accepted
    data, query = rng.random((40, 2)), rng.random((9, 2))
    data[7] = data[3]
    ia, da = orc.knn(data, query, 3)
    ib, db = twin.knn(data, query, 3)
    assert np.array_equal(ia, ib) and np.allclose(da, db, rtol=0, atol=1e-15)
