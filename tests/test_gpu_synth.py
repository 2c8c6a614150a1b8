"""GPU parity on seeded synthetic sequences (sizes the oracle finishes in seconds) and size-independent
properties at BASELINE.json's full sizes.  Everything goes through the C ABI."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


def _run_sequence(pre3, orc, N, steps, n_hyp, dtype, tol_x, tol_P, chain_on_oracle=True, seed=None):
    seq = synth.make_sequence(N, steps, n_hyp, seed=seed)
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype=dtype, max_hyp=n_hyp, std_z=1.0)
    x, P = seq["x0"], seq["P0"]
    f.set_x_p_k_k(x, P)
    for s in seq["steps"]:
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=True)
        ref = orc.step(types, off, seq["cam"], x, P, s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0, early_exit=True)
        li, hi = f.get_flags()
        assert np.array_equal(li, ref["li"]), "LI set differs"
        assert np.array_equal(hi, ref["hi"]), "HI set differs"
        r = ref["ransac"]
        assert (st["best"], st["iters"], st["n_hyp"], st["max_support"]) == (r["best"], r["iters"], r["n_hyp"], r["max_support"])
        xg, Pg = f.get_x_k_k(), f.get_p_k_k()
        assert np.abs(xg - ref["x_kk"]).max() < tol_x
        assert np.abs(Pg - ref["P_kk"]).max() < tol_P * np.abs(ref["P_kk"]).max()
        assert np.abs(Pg - Pg.T).max() == 0.0 or np.abs(Pg - Pg.T).max() < 1e-3 * tol_P * np.abs(Pg).max()
        # chain on the oracle's state so that tolerances do not compound (fp32) / on the GPU's own state (fp64)
        x, P = ref["x_kk"], ref["P_kk"]
        if chain_on_oracle:
            f.set_x_p_k_k(x, P)
    f.close()


def test_sequence_fp64_n50(pre3, orc):
    _run_sequence(pre3, orc, 50, 4, 40, "f64", 1e-10, 1e-11, chain_on_oracle=False)


def test_sequence_fp64_n120(pre3, orc):
    _run_sequence(pre3, orc, 120, 2, 60, "f64", 1e-10, 1e-11)


def test_sequence_fp32_n50(pre3, orc):
    # fp32 covariance path: inlier index sets must still be identical on this data; P to 5e-4 of its scale
    _run_sequence(pre3, orc, 50, 4, 40, "f32", 2e-5, 5e-4)


def test_predict_parity(pre3, orc):
    seq = synth.make_sequence(30, 1, 4, seed=9)
    x, P = seq["x0"].copy(), seq["P0"]
    x[3:7] = np.array([0.98, 0.1, -0.12, 0.05]) / np.linalg.norm([0.98, 0.1, -0.12, 0.05])
    x[7:13] = 0.3
    u = seq["steps"][0]["u"]
    xr, Pr = orc.predict(x, P, u)
    for dtype, tol in (("f64", 1e-13), ("f32", 2e-6)):
        xg, Pg = pre3.predict_state_and_covariance(x, P, u, dtype=dtype)
        assert np.abs(xg - xr).max() < 1e-14
        assert np.abs(Pg - Pr).max() < tol * np.abs(Pr).max()


def test_mixed_landmark_types(pre3, orc):
    """Cartesian + inverse-depth landmarks through projection, S_i, RANSAC, updates (unpinned branch: GPU vs oracle)."""
    from test_oracle import _mixed_problem
    types, x, P, cam = _mixed_problem(seed=4, N=30)
    t, off, n = orc.landmark_table(types)
    f = pre3.EkfFilter(cam, types, dtype="f64", max_hyp=32, std_z=1.0)
    f.set_x_p_k_km1(x, P)
    f.search_IC_matches()
    fld = f.landmark_fields()
    h, has = orc.project(t, off, x, cam)
    Hc, Hl = orc.jacobian(t, off, x, cam, h, has)
    S = orc.innovation(t, off, P, Hc, Hl, has)
    assert np.array_equal(fld["has_h"], has)
    v = has.astype(bool)
    assert np.abs(fld["h"][v] - h[v]).max() < 1e-10
    assert np.abs(fld["Hc"][v] - Hc[v]).max() < 1e-8 and np.abs(fld["Hl"][v] - Hl[v]).max() < 1e-8
    assert np.abs(fld["S"][v] - S[v]).max() < 1e-9
    vis = np.nonzero(has)[0].astype(np.int32)
    rng = np.random.default_rng(1)
    z = np.zeros((len(types), 2))
    z[vis] = h[vis] + rng.normal(0, 0.4, (len(vis), 2))
    z[vis[3]] += 25
    f.set_measurements(vis, z[vis])
    hyp = np.stack([rng.permutation(len(vis))[:3] for _ in range(20)]).astype(np.int32)
    out = f.ransac_hypotheses(hyp, threshold=1.0, early_exit=False)
    ref = orc.ransac(t, off, x, P, Hc, Hl, z, h, vis, vis, cam, hyp, 1.0, early_exit=False)
    assert np.array_equal(out["support"], ref["support"]) and np.array_equal(out["li_mask"], ref["li_mask"])
    f.ekf_update_li_inliers()
    sel = vis[ref["li_mask"].astype(bool)]
    xr, Pr = orc.update_landmarks(t, off, sel, x, P, Hc, Hl, z, h)
    # this prior is rank-12 + 1e-6 I, so cond(S) ~ 1e4 and P shrinks 100x: tolerance relative to the PRIOR scale
    assert np.abs(f.get_x_k_k() - xr).max() < 1e-10 and np.abs(f.get_p_k_k() - Pr).max() < 1e-11 * np.abs(P).max()
    f.close()


def test_update_all_and_empty_measurements(pre3, orc):
    seq = synth.make_sequence(20, 1, 4, seed=21)
    types, off, n = orc.landmark_table(np.zeros(20, int))
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=4)
    f.set_x_p_k_km1(seq["x0"], seq["P0"])
    f.search_IC_matches()
    # 'PURE_EKF' branch: ekf_update_all with all IC landmarks
    f.set_measurements(s["meas_idx"], s["z"])
    f.ekf_update_all()
    h, has = orc.project(types, off, seq["x0"], seq["cam"])
    Hc, Hl = orc.jacobian(types, off, seq["x0"], seq["cam"], h, has)
    z = np.zeros((20, 2))
    z[s["meas_idx"]] = s["z"]
    xr, Pr = orc.update_landmarks(types, off, s["meas_idx"], seq["x0"], seq["P0"], Hc, Hl, z, h)
    assert np.abs(f.get_p_k_k() - Pr).max() < 1e-11 * np.abs(Pr).max()
    assert np.abs(f.get_x_k_k() - xr).max() < 1e-10
    # no measurements at all: the step degenerates to predict; updates are the identity (update.m:50-55)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    st = f.step(s["u"], np.zeros(0, np.int32), np.zeros((0, 2)), np.zeros((1, 3), np.int32))
    xr, Pr = orc.predict(seq["x0"], seq["P0"], s["u"])
    assert st["n_li"] == 0 and st["n_hi"] == 0
    assert np.abs(f.get_x_k_k() - xr).max() < 1e-14 and np.abs(f.get_p_k_k() - Pr).max() < 1e-13 * np.abs(Pr).max()
    f.close()


def test_window_gate(pre3, orc):
    seq = synth.make_sequence(40, 1, 4, seed=33)
    types, off, n = orc.landmark_table(np.zeros(40, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=4)
    f.set_x_p_k_km1(seq["x0"], seq["P0"])
    f.search_IC_matches()
    fld = f.landmark_fields()
    pred = np.nonzero(fld["has_h"])[0]
    rng = np.random.default_rng(2)
    k1 = np.sort(rng.choice(len(pred), 25, replace=False)).astype(np.int32)
    zc = fld["h"][pred[k1]] + rng.normal(0, 6, (25, 2))
    for strict in (True, False):
        f.search_IC_matches()
        acc = f.matching(k1, zc, strict_reference=strict)
        ref = orc.window_gate(pred, k1, zc, fld["h"], fld["S"], fld["has_h"], strict_reference=strict)
        assert np.array_equal(acc, ref)
        assert 0 < acc.sum() < 25
    f.close()


def test_full_size_properties_config3(pre3):
    """N=500 (n=3013), 200 hypotheses, fp32: no oracle at this size in seconds -> properties:
    exact symmetry, positive diagonal, covariance never grows in the update, unit quaternion, stable inlier
    counts, and fp32 vs fp64 agreement of the inlier sets on the first step."""
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 3, n_hyp)
    types = np.zeros(N, np.int32)
    res = {}
    for dtype in ("f64", "f32"):
        f = pre3.EkfFilter(seq["cam"], types, dtype=dtype, max_hyp=n_hyp)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        flags = []
        for s in seq["steps"]:
            st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=True)
            li, hi = f.get_flags()
            flags.append((li.copy(), hi.copy()))
            inl = (li | hi).astype(bool)
            assert inl[s["outliers"]].sum() <= 2                       # gross outliers rejected
            assert inl.sum() >= 0.9 * (len(li) - len(s["outliers"]))   # true matches kept
        P = f.get_p_k_k()
        x = f.get_x_k_k()
        # exactly symmetric except the 4 quaternion rows/cols the Jnorm rebuild touches (as in the reference: 6.6e-24 there)
        A = np.abs(P - P.T)
        assert A.max() < 1e-12 * np.abs(P).max()
        A[3:7, :] = 0
        A[:, 3:7] = 0
        A[:7, :7] = 0
        assert A.max() == 0.0
        assert P.diagonal().min() > 0
        assert abs(np.linalg.norm(x[3:7]) - 1) < 1e-12
        assert np.trace(P) < np.trace(seq["P0"])
        res[dtype] = (flags, x, P)
        f.close()
    assert np.array_equal(res["f64"][0][0][0], res["f32"][0][0][0])     # first-step LI sets identical
    assert np.abs(res["f64"][2] - res["f32"][2]).max() < 2e-3 * np.abs(res["f64"][2]).max()


def test_deferred_hi_update_gives_identical_results(pre3, orc):
    """PRE3_OPT_DEFER_HI: the HI update of step k is completed by the next call on the context; states and flags are the same."""
    N = 60
    seq = synth.make_sequence(N, 12, 30, seed=77, sigma_z=0.6)          # noisier pixels: several steps rescue HI inliers
    types, off, n = orc.landmark_table(np.zeros(N, int))
    outs = []
    for defer in (False, True):
        f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=30, std_z=1.0)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.defer_hi_update(defer)
        n_hi = []
        for s in seq["steps"]:
            st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0)
            n_hi.append(st["n_hi"])
        li, hi = f.get_flags()                                        # any call completes the pending update
        outs.append((f.get_x_k_k(), f.get_p_k_k(), li, hi, n_hi))
        f.close()
    (x0, P0, li0, hi0, nh0), (x1, P1, li1, hi1, nh1) = outs
    assert np.array_equal(x0, x1) and np.array_equal(P0, P1) and np.array_equal(li0, li1) and np.array_equal(hi0, hi1)
    assert sum(nh0) > 0, "the sequence must exercise HI updates"
    assert nh1[1:] == nh0[:-1]                                        # deferred mode reports the previous step's count


def test_k9_bf16_split_matches_f32_mfma_and_fp64(pre3, orc):
    """PRE3_OPT_K9_BF16X3: the down-date as a three-way bf16 split (six bf16 MFMA products, f32 accumulate) against the plain f32 MFMA
    and against the fp64 path, over whole steps (several panels, 128x128 and 64x64 tiles, LI and HI updates).  Tolerance: the
    split form may not be further from fp64 than 1.5x the f32 form's own distance (+ a floor), and P stays exactly symmetric."""
    N = 120                                                            # n = 733 -> ld = 768: 21 tiles of 128, a partial last block
    seq = synth.make_sequence(N, 4, 40, seed=91)
    types = np.zeros(N, np.int32)
    res = {}
    for tag, dtype, b3 in (("f64", "f64", None), ("mfma", "f32", False), ("b3", "f32", True)):
        f = pre3.EkfFilter(seq["cam"], types, dtype=dtype, max_hyp=40, std_z=1.0)
        if b3 is not None:
            assert f.k9_bf16x3(b3) == b3
        else:
            assert f.k9_bf16x3() is False and f.k9_bf16x3(True) is False        # no effect on an fp64 context
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        for s in seq["steps"]:
            st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        res[tag] = (f.get_x_k_k(), f.get_p_k_k(), st)
        f.close()
    x64, P64, s64 = res["f64"]
    scale = np.abs(P64).max()
    e_mfma, e_b3 = np.abs(res["mfma"][1] - P64).max() / scale, np.abs(res["b3"][1] - P64).max() / scale
    assert res["mfma"][2]["n_li"] == res["b3"][2]["n_li"] == s64["n_li"] and s64["n_li"] > 64          # more than one panel
    assert e_b3 <= 1.5 * e_mfma + 2e-6, (e_b3, e_mfma)
    assert e_b3 < 1e-3
    assert np.array_equal(res["b3"][1], res["b3"][1].T)
    sig = np.sqrt(np.diag(P64))
    assert (np.abs(res["b3"][0] - x64) / sig).max() < 5e-3


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_step_predicted_equals_the_call_by_call_sequence_and_the_whole_step(pre3, dtype):
    """pre3_step_predicted (RANSAC .. HI update behind a prediction and installed measurements: what a frame loop with the IC search on the device
    calls) against (a) the call-by-call sequence ransac_hypotheses / ekf_update_li_inliers / rescue_hi_inliers / ekf_update_hi_inliers and (b)
    pre3_step on the same inputs: flags, statistics, x and P bit for bit -- they are the same launches' arithmetic."""
    N, n_hyp = 120, 60
    seq = synth.make_sequence(N, 4, n_hyp, seed=91, motion_noise=2.5)
    outs = []
    for mode in ("predicted", "calls", "step"):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, std_z=1.0)
        f.step_tail(True)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        st_all = []
        for s in seq["steps"]:
            if mode == "step":
                st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
            else:
                f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
                if mode == "predicted":
                    st = f.step_predicted(s["hyp"], threshold=1.0, early_exit=False)
                else:
                    r = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)
                    f.ekf_update_li_inliers(); f.rescue_hi_inliers(); f.ekf_update_hi_inliers()
                    st = dict(best=r["best"], max_support=r["max_support"])
            li, hi = f.get_flags()
            st_all.append((st["best"], st["max_support"], li.tobytes(), hi.tobytes()))
        outs.append((st_all, f.get_x_k_k(), f.get_p_k_k()))
        f.close()
    for k, other in enumerate(outs[1:]):
        assert outs[0][0] == other[0]
        if dtype == "f32" and k == 0:
            # (round 5) inside pre3_step / pre3_step_predicted an fp32 context runs the rescue stage and the HI update in the LI update's persistent
            # launch (PRE3_OPT_STEP_TAIL): P - W'W - W~'W~ in one sweep, P_LI never rounded to fp32.  The call-by-call sequence rounds it: same
            # flags and statistics (asserted above), x / P to fp32 rounding over the four chained steps
            assert np.abs(outs[0][1] - other[1]).max() < 2e-5 and np.abs(outs[0][2] - other[2]).max() < 3e-4 * np.abs(other[2]).max()
        else:
            assert np.array_equal(outs[0][1], other[1]) and np.array_equal(outs[0][2], other[2])
    assert any(np.frombuffer(h, np.int32).sum() > 0 for _, _, _, h in outs[0][0])          # the rescue stage found work somewhere


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_step_all_equals_the_call_by_call_pure_ekf_branch(pre3, dtype):
    """pre3_step_all (mono_slam.m:153-162 + :199, EST_METHOD 'PURE_EKF': prediction, search_IC_matches' projection / Jacobians / S_i, ekf_update_all as
    one call with the projection and S_i riding in other launches) against ekf_prediction / search_IC_matches / set_measurements / ekf_update_all:
    the same arithmetic, so x, P and the landmark table must agree bit for bit; an empty measurement set included."""
    N = 90
    seq = synth.make_sequence(N, 4, 8, seed=17, outlier_frac=0.0)
    outs = []
    for mode in ("one", "calls"):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=8, std_z=1.0)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        for t, s in enumerate(seq["steps"]):
            mi, z = (s["meas_idx"], s["z"]) if t != 2 else (s["meas_idx"][:0], s["z"][:0])
            if mode == "one":
                f.step_all(s["u"], mi, z)
            else:
                f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(mi, z); f.ekf_update_all()
        fl = f.landmark_fields()
        outs.append((f.get_x_k_k(), f.get_p_k_k(), fl["h"], fl["Hc"], fl["S"]))
        f.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_a_step_of_outliers_then_a_normal_one(pre3, orc):
    """pre3_step at the headline's size (fp32, persistent launch + riders) when every measurement is an outlier: the only low-innovation inliers are the
    winning hypothesis' own draws -- an LI update of a handful of rows (one panel, cut short), the projection on the strips and the gate riding with
    the Jnorm pass as for any other update --, then a normal step.  Both against the oracle."""
    N, n_hyp = 500, 40
    seq = synth.make_sequence(N, 2, n_hyp, seed=77)
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f32", max_hyp=n_hyp, std_z=1.0)
    x, P = seq["x0"], seq["P0"]
    f.set_x_p_k_k(x, P)
    rng = np.random.default_rng(5)
    for k, s in enumerate(seq["steps"]):
        z = np.array(s["z"], float)
        if k == 0:
            z = z + rng.uniform(40.0, 90.0, z.shape) * rng.choice([-1.0, 1.0], z.shape)      # nothing agrees with anything
        st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False)
        ref = orc.step(types, off, seq["cam"], x, P, s["u"], s["meas_idx"], z, s["hyp"], 1.0, early_exit=False)
        li, hi = f.get_flags()
        assert np.array_equal(li, ref["li"]) and np.array_equal(hi, ref["hi"])
        if k == 0:
            assert st["n_li"] <= 3 and st["n_li"] == int(ref["li"].sum())
        else:
            assert st["n_li"] > 100
        xg, Pg = f.get_x_k_k(), f.get_p_k_k()
        assert np.abs(xg - ref["x_kk"]).max() < 2e-5
        assert np.abs(Pg - ref["P_kk"]).max() < 5e-4 * np.abs(ref["P_kk"]).max()
        # (the second step starts from the sequence's own state again: after an update with outliers the estimate is far from the truth, and what
        #  the fp32 covariance path makes of THAT -- nine rescue decisions differ from fp64 -- is not what this test is about)
        x, P = seq["x0"], seq["P0"]
        f.set_x_p_k_k(x, P)
    f.close()
