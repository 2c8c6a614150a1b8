"""GPU parity on the reference's own SR4000 snapshot (tests/golden/sr4000_step3.npz), through the C ABI.

fp64: inlier sets bit-exact; S, h, H within 1e-11 abs; x_k_k, P_k_k within 1e-12 of P's scale (the MATLAB
values are reproduced to 2e-13 by the oracle; the GPU path uses a Cholesky solve instead of inv(S)).
fp32 covariance path: same inlier sets on this data; P within 2e-5 relative to max|P|.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _filter(pre3, g, dtype):
    f = pre3.EkfFilter(g["cam"], np.zeros(g["N"], np.int32), dtype=dtype, max_hyp=64, std_z=g["std_z"])
    f.set_x_p_k_km1(g["x_k_km1"], g["p_k_km1"])
    return f


@pytest.mark.parametrize("dtype,tolS,tolP", [("f64", 1e-11, 1e-12), ("f32", 5e-5, 2e-5)])
def test_snapshot_chain(pre3, sr4000, dtype, tolS, tolP):
    g = sr4000
    f = _filter(pre3, g, dtype)
    # search_IC_matches at (x_k_km1, p_k_km1): stored S
    f.search_IC_matches()
    fld = f.landmark_fields()
    assert fld["has_h"].all()
    assert np.abs(fld["S"] - g["S"]).max() <= tolS * 10
    # measurements + the stored LI flags -> LI update
    f.set_measurements(g["meas_idx"], g["z"][g["meas_idx"]])
    f.set_flags(li=g["low_innovation_inlier"][g["meas_idx"]])
    f.ekf_update_li_inliers()
    # rescue at the post-LI state: stored h/H belong to this state
    hi = f.rescue_hi_inliers()
    fld = f.landmark_fields()
    assert np.abs(fld["h"] - g["h"]).max() < (1e-10 if dtype == "f64" else 1e-3)
    scaleH = np.abs(g["Hcam"]).max()
    assert np.abs(fld["Hc"] - g["Hcam"]).max() < (1e-11 if dtype == "f64" else 1e-4) * scaleH
    assert np.abs(fld["Hl"] - g["Hlm"]).max() < (1e-11 if dtype == "f64" else 1e-4) * scaleH
    assert np.array_equal(hi, g["high_innovation_inlier"][g["meas_idx"]])          # bit-exact HI set {125}
    f.ekf_update_hi_inliers()
    x, P = f.get_x_k_k(), f.get_p_k_k()
    pscale = np.abs(g["p_k_k"]).max()
    assert np.abs(x - g["x_k_k"]).max() < (1e-12 if dtype == "f64" else 1e-6)
    assert np.abs(P - g["p_k_k"]).max() < tolP * pscale
    assert np.abs(P - P.T).max() <= (1e-18 if dtype == "f64" else 1e-9)
    f.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_snapshot_ransac(pre3, orc, sr4000, dtype):
    """39 of the 40 one-point hypotheses give exactly the stored LI mask at thr = std_z = 2;
    supports and masks must equal the oracle's for every hypothesis."""
    g = sr4000
    f = _filter(pre3, g, dtype)
    f.search_IC_matches()
    f.set_measurements(g["meas_idx"], g["z"][g["meas_idx"]])
    m = len(g["meas_idx"])
    hyp = np.arange(m, dtype=np.int32).reshape(-1, 1)
    out = f.ransac_hypotheses(hyp, threshold=g["std_z"], early_exit=False)
    types, off, _ = orc.landmark_table(np.zeros(g["N"], int))
    h0, has0 = orc.project(types, off, g["x_k_km1"], g["cam"])
    Hc0, Hl0 = orc.jacobian(types, off, g["x_k_km1"], g["cam"], h0, has0)
    ref = orc.ransac(types, off, g["x_k_km1"], g["p_k_km1"], Hc0, Hl0, g["z"], h0, g["ic_idx"], g["meas_idx"], g["cam"], hyp,
                     g["std_z"], early_exit=False)
    assert np.array_equal(out["support"], ref["support"])
    assert out["best"] == ref["best"] and out["max_support"] == ref["max_support"]
    assert np.array_equal(out["li_mask"], ref["li_mask"])
    assert (out["support"] == 39).sum() == 39 and (out["support"] == 40).sum() == 1
    # the reference's early-exit replay (quirk Q1)
    out2 = f.ransac_hypotheses(hyp, threshold=g["std_z"], early_exit=True)
    ref2 = orc.ransac(types, off, g["x_k_km1"], g["p_k_km1"], Hc0, Hl0, g["z"], h0, g["ic_idx"], g["meas_idx"], g["cam"], hyp,
                      g["std_z"], early_exit=True)
    for k in ("best", "iters", "n_hyp", "max_support"):
        assert out2[k] == ref2[k], k
    assert np.array_equal(out2["li_mask"], ref2["li_mask"])
    assert np.array_equal(out2["support"], ref2["support"])
    # 3-point hypotheses as the reference draws them when #IC > 3 (select_random_match.m:47-48)
    rng = np.random.default_rng(7)
    hyp3 = np.stack([rng.permutation(m)[:3] for _ in range(50)]).astype(np.int32)
    out3 = f.ransac_hypotheses(hyp3, threshold=g["std_z"], early_exit=False)
    ref3 = orc.ransac(types, off, g["x_k_km1"], g["p_k_km1"], Hc0, Hl0, g["z"], h0, g["ic_idx"], g["meas_idx"], g["cam"], hyp3,
                      g["std_z"], early_exit=False)
    assert np.array_equal(out3["support"], ref3["support"])
    assert np.array_equal(out3["li_mask"], ref3["li_mask"])
    f.close()


def test_snapshot_full_1pre_flow(pre3, sr4000):
    """RANSAC -> LI update -> rescue -> HI update with the flags computed on the device (fp64)."""
    g = sr4000
    f = _filter(pre3, g, "f64")
    f.search_IC_matches()
    f.set_measurements(g["meas_idx"], g["z"][g["meas_idx"]])
    m = len(g["meas_idx"])
    # a hypothesis order whose first entry is one of the 39 "good" ones: the replay then stops with the stored LI set
    hyp = np.arange(m, dtype=np.int32).reshape(-1, 1)
    out = f.ransac_hypotheses(hyp, threshold=g["std_z"], early_exit=False)
    good = int(np.nonzero(out["support"] == 39)[0][0])
    out = f.ransac_hypotheses(hyp[good:good + 1], threshold=g["std_z"], early_exit=True)
    assert np.array_equal(out["li_mask"], g["low_innovation_inlier"][g["meas_idx"]])
    f.ekf_update_li_inliers()
    hi = f.rescue_hi_inliers()
    assert np.array_equal(hi, g["high_innovation_inlier"][g["meas_idx"]])
    f.ekf_update_hi_inliers()
    P = f.get_p_k_k()
    assert np.abs(P - g["p_k_k"]).max() < 1e-12 * np.abs(g["p_k_k"]).max()
    assert np.abs(f.get_x_k_k() - g["x_k_k"]).max() < 1e-12
    f.close()


def test_stateless_update_matches_reference_snapshot(pre3, orc, sr4000):
    """update(x,P,H,R,z,h) drop-in (update.m:27) on the snapshot's LI rows, incl. the gain K."""
    g = sr4000
    n, N = g["n"], g["N"]
    types, off, _ = orc.landmark_table(np.zeros(N, int))
    h0, has0 = orc.project(types, off, g["x_k_km1"], g["cam"])
    Hc0, Hl0 = orc.jacobian(types, off, g["x_k_km1"], g["cam"], h0, has0)
    li = g["li_idx"]
    H = np.zeros((2 * len(li), n))
    for s, i in enumerate(li):
        H[2 * s:2 * s + 2, 0:7] = Hc0[i]
        H[2 * s:2 * s + 2, off[i]:off[i] + 6] = Hl0[i]
    z = g["z"][li].ravel()
    h = h0[li].ravel()
    xr, Pr, Kr = orc.update(g["x_k_km1"], g["p_k_km1"], H, None, z, h, want_K=True)
    x, P, K = pre3.update(g["x_k_km1"], g["p_k_km1"], H, np.eye(len(z)), z, h, dtype="f64")
    assert np.abs(x - xr).max() < 1e-12
    assert np.abs(P - Pr).max() < 1e-12 * np.abs(Pr).max()
    assert np.abs(K - Kr).max() < 1e-10 * np.abs(Kr).max()
    # empty z: inputs returned, K = 0 (update.m:50-55)
    x0, P0, K0 = pre3.update(g["x_k_km1"], g["p_k_km1"], np.zeros((0, n)), None, [], [])
    assert np.array_equal(x0, g["x_k_km1"]) and np.array_equal(P0, g["p_k_km1"]) and K0 == 0
