"""GPU: edge cases the reference's code paths imply -- ragged / empty / degenerate inputs, all hypothesis sizes,
error behaviour (status codes, never a crash), maximum configured size (BASELINE.json configs[4])."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_ransac_all_hypothesis_sizes(pre3, orc, k):
    """select_random_match.m:47-51 uses 3 landmarks when #IC > 3, else 1; the ABI accepts 1..4"""
    N = 36
    seq = synth.make_sequence(N, 1, 25, k=k, seed=40 + k)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=25)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.ekf_prediction(s["u"])
    f.search_IC_matches()
    f.set_measurements(s["meas_idx"], s["z"])
    x1, P1 = orc.predict(seq["x0"], seq["P0"], s["u"])
    h, has = orc.project(types, off, x1, seq["cam"])
    Hc, Hl = orc.jacobian(types, off, x1, seq["cam"], h, has)
    z = np.zeros((N, 2)); z[s["meas_idx"]] = s["z"]
    for ee in (False, True):
        out = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=ee)
        ref = orc.ransac(types, off, x1, P1, Hc, Hl, z, h, s["meas_idx"], s["meas_idx"], seq["cam"], s["hyp"], 1.0, early_exit=ee)
        assert np.array_equal(out["support"], ref["support"]) and np.array_equal(out["li_mask"], ref["li_mask"])
        assert (out["best"], out["iters"], out["n_hyp"], out["max_support"]) == (ref["best"], ref["iters"], ref["n_hyp"], ref["max_support"])
    f.close()


def test_fewer_measurements_than_k_and_single_measurement(pre3, orc):
    N = 12
    seq = synth.make_sequence(N, 1, 4, seed=3)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    # two measurements, k = 3 requested: the step skips RANSAC (no LI set), rescue may still pick both up
    st = f.step(s["u"], s["meas_idx"][:2], s["z"][:2], np.zeros((4, 3), np.int32))
    ref = orc.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"][:2], s["z"][:2], np.zeros((4, 3), np.int32), 1.0)
    li, hi = f.get_flags()
    assert st["n_li"] == 0 and np.array_equal(hi, ref["hi"])
    assert np.abs(f.get_p_k_k() - ref["P_kk"]).max() < 1e-11 * np.abs(ref["P_kk"]).max()
    # one measurement, 1-point hypotheses (the reference's own fallback)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    st = f.step(s["u"], s["meas_idx"][:1], s["z"][:1], np.zeros((3, 1), np.int32))
    ref = orc.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"][:1], s["z"][:1], np.zeros((3, 1), np.int32), 1.0)
    assert st["n_li"] == int(ref["li"].sum()) == 1
    assert np.abs(f.get_x_k_k() - ref["x_kk"]).max() < 1e-11
    f.close()


def test_landmarks_behind_the_camera_are_not_predicted(pre3, orc):
    N = 20
    seq = synth.make_sequence(N, 1, 4, seed=6)
    x = seq["x0"].copy()
    x[3:7] = [0.0, 0.0, 1.0, 0.0]                      # camera turned by 180 deg about y: everything is behind it
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=4)
    f.set_x_p_k_km1(x, seq["P0"])
    f.search_IC_matches()
    fld = f.landmark_fields()
    h, has = orc.project(types, off, x, seq["cam"])
    assert has.sum() == 0 and fld["has_h"].sum() == 0
    # quirk Q7: with clear_first=0 a landmark that is no longer visible keeps its previous h and gets an H at it
    f.set_x_p_k_km1(seq["x0"], seq["P0"])
    f.search_IC_matches()
    before = f.landmark_fields()
    f.set_x_p_k_k(x, seq["P0"])
    f.predict_camera_measurements(which=pre3.X_K_K, clear_first=False)
    after = f.landmark_fields()
    assert np.array_equal(after["has_h"], before["has_h"]) and np.array_equal(after["h"], before["h"])
    h0, has0 = orc.project(types, off, seq["x0"], seq["cam"])
    h1, has1 = orc.project(types, off, x, seq["cam"], h0, has0)
    Hc1, Hl1 = orc.jacobian(types, off, x, seq["cam"], h1, has1)
    v = has1.astype(bool)
    assert np.abs(after["Hc"][v] - Hc1[v]).max() < 1e-8 * max(1.0, np.abs(Hc1[v]).max())
    f.close()


def test_error_codes_not_crashes(pre3):
    N = 10
    seq = synth.make_sequence(N, 1, 4, seed=2)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=4)
    with pytest.raises(pre3.Pre3Error) as e:                      # call order: update before any state
        f.ekf_update_li_inliers()
    assert e.value.code in (-4, -1)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    with pytest.raises(pre3.Pre3Error):                           # wrong state size
        f.set_x_p_k_k(seq["x0"][:-6], seq["P0"][:-6, :-6])
    with pytest.raises(pre3.Pre3Error):                           # measurement indices must be ascending and in range
        f.set_measurements([3, 2], np.zeros((2, 2)))
    with pytest.raises(pre3.Pre3Error):
        f.set_measurements([0, N], np.zeros((2, 2)))
    f.ekf_prediction(s["u"])
    f.search_IC_matches()
    f.set_measurements(s["meas_idx"], s["z"])
    with pytest.raises(pre3.Pre3Error):                           # more draws than the context was sized for
        f.ransac_hypotheses(np.zeros((5, 3), np.int32))
    with pytest.raises(pre3.Pre3Error):                           # draw outside the IC list
        f.ransac_hypotheses(np.full((2, 3), len(s["meas_idx"]), np.int32))
    with pytest.raises(pre3.Pre3Error):                           # the covariance buffer holds p_k_km1, not p_k_k
        f.get_p_k_k()
    # an indefinite "covariance" makes S non positive definite: reported as PRE3_E_NUMERIC (-5), no crash
    bad = -np.eye(seq["n"]) * 10.0
    f.set_x_p_k_km1(seq["x0"], bad)
    f.search_IC_matches()
    f.set_measurements(s["meas_idx"], s["z"])
    f.ekf_update_all()
    with pytest.raises(pre3.Pre3Error) as e:
        f.get_p_k_k()
    assert e.value.code == -5
    f.close()
    with pytest.raises(pre3.Pre3Error):                           # H row with more than 16 non-zeros
        pre3.update(np.zeros(31), np.eye(31), np.ones((2, 31)), None, [0.0, 0.0], [0.0, 0.0])


def test_stateless_update_dense_R_and_fp32(pre3, orc):
    """update(x,P,H,R,z,h) with a general symmetric R (update.m:32 adds whatever R it is given)"""
    seq = synth.make_sequence(15, 1, 4, seed=12)
    types, off, n = orc.landmark_table(np.zeros(15, int))
    h, has = orc.project(types, off, seq["x0"], seq["cam"])
    Hc, Hl = orc.jacobian(types, off, seq["x0"], seq["cam"], h, has)
    sel = np.nonzero(has)[0][:6]
    H = np.zeros((12, n))
    for s_, i in enumerate(sel):
        H[2 * s_:2 * s_ + 2, 0:7] = Hc[i]
        H[2 * s_:2 * s_ + 2, off[i]:off[i] + 6] = Hl[i]
    rng = np.random.default_rng(0)
    A = rng.standard_normal((12, 12))
    R = A @ A.T * 0.05 + np.eye(12) * 0.7
    z = (h[sel] + 0.4).ravel()
    xr, Pr, Kr = orc.update(seq["x0"], seq["P0"], H, R, z, h[sel].ravel(), want_K=True)
    x, P, K = pre3.update(seq["x0"], seq["P0"], H, R, z, h[sel].ravel(), dtype="f64")
    assert np.abs(P - Pr).max() < 1e-11 * np.abs(Pr).max() and np.abs(K - Kr).max() < 1e-9 * np.abs(Kr).max()
    x32, P32, _ = pre3.update(seq["x0"], seq["P0"], H, R, z, h[sel].ravel(), dtype="f32", want_K=False)
    assert np.abs(P32 - Pr).max() < 5e-4 * np.abs(seq["P0"]).max() and np.abs(x32 - xr).max() < 1e-4


def test_maximum_configured_size_properties(pre3):
    """BASELINE.json configs[4]: N=2000 (n=12013), 1000 hypotheses, fp32.  One full step; size-independent checks."""
    N, n_hyp = 2000, 1000
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
    li, hi = f.get_flags()
    inl = (li | hi).astype(bool)
    # first step from P0: part of the true matches is outside the chi-square rescue gate; no gross outlier may enter
    assert inl[s["outliers"]].sum() <= 8 and inl.sum() >= 0.75 * (len(li) - len(s["outliers"]))
    assert st["n_li"] == li.sum() and st["n_hi"] == hi.sum() and st["max_support"] == st["n_li"]
    x = f.get_x_k_k()
    P = f.get_p_k_k()
    assert abs(np.linalg.norm(x[3:7]) - 1) < 1e-12 and np.isfinite(P).all()
    A = np.abs(P - P.T)
    A[3:7, :] = 0; A[:, 3:7] = 0
    assert A.max() == 0.0 and P.diagonal().min() > 0 and np.trace(P) < np.trace(seq["P0"])
    f.close()
    # the same step with the down-date (and the factorisation's pending updates) on the f32 MFMA instead of the bf16 split: 17 rounds of
    # 128x128 tiles + 64x64 left-overs against the persistent 64x64 form -- same inlier sets, P equal to f32 rounding
    g = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp)
    assert g.k9_bf16x3(False) is False
    g.set_x_p_k_k(seq["x0"], seq["P0"])
    st2 = g.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
    li2, hi2 = g.get_flags()
    P2 = g.get_p_k_k()
    g.close()
    assert np.array_equal(li, li2) and np.array_equal(hi, hi2) and st2["n_li"] == st["n_li"]
    d = np.sqrt(P.diagonal())
    assert (np.abs(P - P2) / np.outer(d, d)).max() < 2e-3


def test_ransac_slices_with_partial_products_equal_the_full_round(pre3, orc):
    """pre3_ransac_score on a slice multiplies out H*P / H*P*H' only for the measurements that slice draws; supports and inlier masks of
    the slices, put together, must be those of the full round -- and an LI update after a sliced round must not use the partial products."""
    import torch
    N, n_hyp = 90, 64
    seq = synth.make_sequence(N, 1, n_hyp, seed=123)
    s = seq["steps"][0]
    outs = {}
    for mode in ("full", "sliced"):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f64", max_hyp=n_hyp)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
        if mode == "full":
            res = f.ransac_hypotheses(s["hyp"], 1.0, early_exit=False)
        else:
            words = (len(s["meas_idx"]) + 31) // 32
            sup = torch.zeros(n_hyp, dtype=torch.int32, device="cuda")
            msk = torch.zeros(n_hyp * words, dtype=torch.int32, device="cuda")
            tot_s, tot_m = torch.zeros_like(sup), torch.zeros_like(msk)
            for lo, hi in ((0, 7), (7, 8), (8, 40), (40, 64)):                  # ragged slices, one of a single hypothesis
                f.ransac_score_shard(s["hyp"], 1.0, lo, hi)
                f.ransac_export(n_hyp, sup.data_ptr(), msk.data_ptr())
                torch.cuda.synchronize()
                tot_s += sup; tot_m += msk
            f.ransac_import(n_hyp, tot_s.data_ptr(), tot_m.data_ptr())
            res = f.ransac_select(n_hyp, 3, early_exit=False)
        f.ekf_update_li_inliers()
        outs[mode] = (res, f.get_x_k_k(), f.get_p_k_k())
        f.close()
    a, b = outs["full"], outs["sliced"]
    assert np.array_equal(a[0]["support"], b[0]["support"]) and np.array_equal(a[0]["li_mask"], b[0]["li_mask"]) and a[0]["best"] == b[0]["best"]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
