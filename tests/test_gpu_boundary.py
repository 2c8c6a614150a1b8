"""GPU: the stateless drop-ins of SURVEY 8(b) added in round 2 -- compute_hypothesis_support_fast, predict_state_and_covariance --
through the C ABI against the oracle, plus the call-order guards of the sharded RANSAC entries and the empty-map IC search."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


def _mixed_state(N, seed):
    """A synthetic map with every third landmark converted to a cartesian point (xyz = r + m(theta,phi)/rho)."""
    x0, P0, _ = synth.make_map(N, seed)
    types = np.zeros(N, int)
    types[::3] = 1
    parts = [x0[:13]]
    for i in range(N):
        y = x0[13 + 6 * i:19 + 6 * i]
        if types[i]:
            m = np.array([np.cos(y[4]) * np.sin(y[3]), -np.sin(y[4]), np.cos(y[4]) * np.cos(y[3])])
            parts.append(y[:3] + m / y[5])
        else:
            parts.append(y)
    return types, np.concatenate(parts)


@pytest.mark.parametrize("N,seed", [(30, 1), (90, 2), (400, 3)])
def test_compute_hypothesis_support_fast_matches_the_oracle(pre3, orc, N, seed):
    """compute_hypothesis_support_fast.m:27-116 -- both landmark classes, the min+threshold rule (:70) and the plain rule (:109)"""
    rng = np.random.default_rng(seed)
    types, x = _mixed_state(N, seed)
    t, off, n = orc.landmark_table(types)
    assert n == x.shape[0]
    xi = x + rng.normal(0, 1e-3, n)                       # a hypothesis state: un-normalised quaternion on purpose (quirk Q4)
    h, has = orc.project(t, off, x, synth.CAM)
    meas = np.nonzero(has)[0][: max(3, int(0.8 * N))].astype(np.int32)
    z = h[meas] + rng.normal(0, 1.5, (len(meas), 2))
    z[::7] += 40.0                                          # gross outliers
    has_z = np.zeros(N, bool); has_z[meas] = True
    zfull = np.zeros((N, 2)); zfull[meas] = z
    pat, z_id, z_euc = pre3.generate_state_vector_pattern(types, has_z, zfull)
    for thr in (0.5, 2.0):
        sup, pid, peu = pre3.compute_hypothesis_support_fast(xi, synth.CAM, pat, z_id, z_euc, thr)
        rs, rmask, _ = orc.support(meas, t, off, xi, synth.CAM, z, thr)
        # the oracle's mask is in measurement (= landmark) order; the reference returns the two classes separately
        is_id = types[meas] == 0
        assert sup == rs
        assert np.array_equal(pid, rmask[is_id].astype(bool)) and np.array_equal(peu, rmask[~is_id].astype(bool))
        assert sup == int(pid.sum() + peu.sum())


def test_compute_hypothesis_support_fast_empty_classes_and_bad_pattern(pre3, orc):
    types, x = _mixed_state(12, 5)
    t, off, n = orc.landmark_table(types)
    h, has = orc.project(t, off, x, synth.CAM)
    # only cartesian measurements: positions_li_inliers_id = [] (:75)
    meas = np.nonzero((types == 1) & (has != 0))[0].astype(np.int32)
    has_z = np.zeros(12, bool); has_z[meas] = True
    zfull = np.zeros((12, 2)); zfull[meas] = h[meas] + 0.3
    pat, z_id, z_euc = pre3.generate_state_vector_pattern(types, has_z, zfull)
    sup, pid, peu = pre3.compute_hypothesis_support_fast(x, synth.CAM, pat, z_id, z_euc, 1.0)
    assert pid.size == 0 and sup == len(meas) and peu.all()
    # nothing measured at all: support 0 (:29)
    pat0 = np.zeros((n, 4))
    assert pre3.compute_hypothesis_support_fast(x, synth.CAM, pat0, np.zeros((2, 0)), np.zeros((2, 0)), 1.0)[0] == 0
    # a pattern that does not match the measurement count: MATLAB's reshape (:40) fails, so does the drop-in
    with pytest.raises(pre3.Pre3Error) as e:
        pre3.compute_hypothesis_support_fast(x, synth.CAM, pat, z_id, np.zeros((2, len(meas) + 1)), 1.0)
    assert e.value.code == -1


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-13), ("f32", 3e-6)])
def test_stateless_predict_state_and_covariance(pre3, orc, dtype, tol):
    """predict_state_and_covariance.m:59-143 as a host-in / host-out call (pre3_predict_dense)"""
    N = 25
    seq = synth.make_sequence(N, 1, 4, seed=9)
    u = seq["steps"][0]["u"]
    x1, P1 = pre3.predict_state_and_covariance(seq["x0"], seq["P0"], u, dtype=dtype)
    xr, Pr = orc.predict(seq["x0"], seq["P0"], u)
    assert np.abs(x1 - xr).max() < 1e-13
    assert np.abs(P1 - Pr).max() < tol * np.abs(Pr).max()
    # a mixed map (n = 13 + 6a + 3b): the prediction only touches the camera block and rows/columns 1:13
    types, x = _mixed_state(14, 4)
    n = x.shape[0]
    A = np.random.default_rng(4).normal(0, 1e-2, (n, n)); P = A @ A.T + 1e-6 * np.eye(n)
    x1, P1 = pre3.predict_state_and_covariance(x, P, u, dtype=dtype)
    xr, Pr = orc.predict(x, P, u)
    assert np.abs(x1 - xr).max() < 1e-13 and np.abs(P1 - Pr).max() < tol * np.abs(Pr).max()
    with pytest.raises(pre3.Pre3Error):
        pre3.predict_state_and_covariance(np.zeros(14), np.eye(14), u)          # n is not 13 + 6a + 3b


def test_sharded_ransac_entries_check_the_scored_round(pre3):
    """pre3_ransac_select / _export / _import read the mask buffer laid out by the scored round: a different n_draw or k is refused"""
    N = 30
    seq = synth.make_sequence(N, 1, 16, seed=12)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f64", max_hyp=32)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    f.ransac_score_shard(s["hyp"], 1.0, 0, 16)
    ok = f.ransac_select(16, 3, early_exit=False)
    assert ok["max_support"] > 0
    for bad in ((8, 3), (16, 2)):
        with pytest.raises(pre3.Pre3Error) as e:
            f.ransac_select(bad[0], bad[1], early_exit=False)
        assert e.value.code == -4
    f.close()


def test_ic_search_on_an_empty_map_returns_quietly(pre3):
    """matching_sift_based.m:115 `if isempty(des1) return`: a run starts with no landmarks"""
    f = pre3.EkfFilter(synth.CAM, np.zeros(0, np.int32), dtype="f32", max_landmarks=8)
    x = np.zeros(13); x[3] = 1.0
    f.set_x_p_k_k(x, np.eye(13) * 1e-4)
    f.ekf_prediction(np.array([0, 0, 0, 1.0, 0, 0, 0]))
    f.set_descriptors(np.zeros((128, 0)))
    rng = np.random.default_rng(0)
    f.load_scan(rng.random((128, 5)), rng.random((4, 5)) * 100)
    out = f.matching_sift_based()
    assert out["meas_idx"].size == 0 and out["match_idx"].shape[1] == 0 and f.m == 0
    f.close()
