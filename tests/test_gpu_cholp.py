"""GPU: update.m:32-33 (S = L L', W = L^-1 [HP | nu]) as ONE persistent launch (3pre_amd/csrc/pre3_cholp.hip, fp32 contexts, <= 13 panels of
64 rows) against the launch-per-panel form on the same inputs, and against the oracle where a test says so.  The two forms factor and solve
with different operation orders (blocked inverse-of-diagonal-block solve vs lock-step substitution), so they agree to fp32 rounding of the
update, not bit for bit: tolerances are written at each assert; inlier sets must be identical."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


def _pair(pre3, seq, N, n_hyp):
    fs = []
    for on in (False, True):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
        f.chol_persist(on)
        assert f.chol_persist() == on
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        fs.append(f)
    return fs


@pytest.mark.parametrize("N,n_hyp,steps", [(50, 40, 3), (120, 60, 3), (500, 200, 3)])
def test_whole_steps_agree_between_the_two_forms(pre3, N, n_hyp, steps):
    """LI update (1 .. 10 panels), rescue, HI update (one panel) through pre3_step: same inlier sets, x / P to fp32 rounding of one update"""
    seq = synth.make_sequence(N, steps, n_hyp)
    fs = _pair(pre3, seq, N, n_hyp)
    for s in seq["steps"]:
        out = []
        for f in fs:
            st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
            li, hi = f.get_flags()
            out.append((st, li, hi, f.get_x_k_k(), f.get_p_k_k()))
        (s0, li0, hi0, x0, P0), (s1, li1, hi1, x1, P1) = out
        assert np.array_equal(li0, li1) and np.array_equal(hi0, hi1)
        assert (s0["best"], s0["n_li"], s0["n_hi"]) == (s1["best"], s1["n_li"], s1["n_hi"])
        scale = np.abs(P0).max()
        assert np.isfinite(P1).all()
        assert np.abs(P1 - P0).max() < 1e-4 * scale, np.abs(P1 - P0).max() / scale       # (the fp32-vs-fp64 tolerance of the suite is 3e-4)
        assert np.abs(x1 - x0).max() < 1e-5
        A = np.abs(P1 - P1.T); A[3:7, :] = 0; A[:, 3:7] = 0
        assert A.max() == 0.0                                                            # still exactly symmetric outside the Jnorm rows
        fs[1].set_x_p_k_k(x0, P0)                                                        # each step compares ONE update
    for f in fs:
        f.close()


@pytest.mark.parametrize("n_rows_meas", [1, 31, 32, 33, 97, 400, 416, 500])
def test_update_sizes_from_one_row_pair_to_thirteen_panels(pre3, orc, n_rows_meas):
    """ekf_update_li_inliers with a forced inlier set: r = 2 .. 832 rows (1 .. 13 panels; padding rows when r is not a multiple of 64),
    persistent form vs the C oracle's update.m restatement (fp64) and vs the launch-per-panel form"""
    N, n_hyp = 500, 8
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    m = min(n_rows_meas, len(s["meas_idx"]))
    if m < n_rows_meas:
        # more measurements than the sequence provides: measure further landmarks at their predicted pixels (innovation 0)
        extra = np.setdiff1d(np.arange(N), s["meas_idx"])[: n_rows_meas - m]
        meas = np.sort(np.concatenate([s["meas_idx"], extra])).astype(np.int32)
    else:
        meas = np.asarray(s["meas_idx"][:m], np.int32)
    res = []
    for on in (True, False):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
        f.chol_persist(on)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.ekf_prediction(s["u"])
        f.search_IC_matches()
        h = f.landmark_fields()["h"]
        zmap = {int(i): zz for i, zz in zip(s["meas_idx"], s["z"])}
        z = np.array([zmap.get(int(i), h[int(i)]) for i in meas])
        f.set_measurements(meas, z)
        f.set_flags(li=np.ones(len(meas), np.int32))
        f.ekf_update_li_inliers()
        res.append((f.get_x_k_k(), f.get_p_k_k(), f.get_x_k_km1() if False else None, meas, z))
        f.close()
    (xp, Pp, _, meas, z), (xl, Pl, _, _, _) = res
    scale = np.abs(Pl).max()
    assert np.isfinite(Pp).all()
    # (the two forms differ by fp32 rounding of the update: 1.5e-4 of P's scale up to 13 panels, 2e-4 for the 16 panels of 500 measured landmarks, outliers included.
    #  Round 6: the persistent form's chain also sums its lookahead in another order -- matrix-core fma chain instead of pair sums --, 1.12e-4 at 13 panels
    #  where it was 0.9e-4; each form's distance from the fp64 twin, the parity claim, is the 3e-4 below and in profiles/r6_form_diff.txt)
    assert np.abs(Pp - Pl).max() < (1.5e-4 if n_rows_meas <= 416 else 2e-4) * scale, (n_rows_meas, np.abs(Pp - Pl).max() / scale)
    assert np.abs(xp - xl).max() < 1e-5
    # ... and against update.m's restatement in fp64 (the numpy twin: explicit inv(S), K S K', 0.5 (P + P'), Jnorm rebuild)
    from oracle import np_twin as tw
    xr, Pr = tw.predict(seq["x0"], seq["P0"], s["u"])
    hh, has = tw.project(types, off, xr, seq["cam"])
    Hc, Hl = tw.jacobian(types, off, xr, seq["cam"], hh, has)
    zfull = np.zeros((N, 2)); zfull[meas] = z
    xo, Po = tw.update_landmarks(types, off, meas, xr, Pr, Hc, Hl, zfull, hh)
    assert np.abs(Pp - Po).max() < 3e-4 * np.abs(Po).max(), np.abs(Pp - Po).max() / np.abs(Po).max()
    assert np.abs(xp - xo).max() < 2e-5


def test_an_update_beyond_the_lds_window_streams_older_blocks(pre3):
    """r = 900 rows = 15 panels: more blocks of W than a strip keeps in LDS (a ring of 11): the older ones are re-read from the planes the strip
    wrote (round 4; rounds 1-3 took the launch-per-panel form here).  Against that form on the same input: fp32 rounding of one update."""
    N = 500
    seq = synth.make_sequence(N, 1, 8)
    s = seq["steps"][0]
    res = []
    for on in (True, False):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=8, std_z=1.0)
        f.chol_persist(on)
        assert f.chol_persist() == on
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.ekf_prediction(s["u"])
        f.search_IC_matches()
        meas = np.arange(450, dtype=np.int32)
        f.set_measurements(meas, f.landmark_fields()["h"][:450] + 0.25)
        f.set_flags(li=np.ones(450, np.int32))
        f.ekf_update_li_inliers()
        res.append((f.get_x_k_k(), f.get_p_k_k()))
        f.close()
    (xp, P), (xl, Pl) = res
    assert np.isfinite(P).all() and np.abs(P - P.T).max() < 1e-12 * np.abs(P).max() + 1e-30
    assert (np.diag(P)[7:] > 0).all()
    scale = np.abs(Pl).max()
    assert np.abs(P - Pl).max() < 2e-4 * scale, np.abs(P - Pl).max() / scale        # (fp32 rounding of a 900-row update; 1e-4 up to 13 panels)
    assert np.abs(xp - xl).max() < 1e-5


def test_not_positive_definite_is_reported_by_the_persistent_form(pre3):
    seq = synth.make_sequence(60, 1, 8)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(60, np.int32), dtype="f32", max_hyp=8, std_z=1.0)
    assert f.chol_persist()
    f.set_x_p_k_km1(seq["x0"], -np.eye(seq["n"]) * 10.0)
    f.search_IC_matches()
    f.set_measurements(s["meas_idx"], s["z"])
    f.ekf_update_all()
    with pytest.raises(pre3.Pre3Error) as e:
        f.get_p_k_k()
    assert e.value.code == -5
    f.close()


def test_more_than_two_capable_contexts_fall_back(pre3):
    """a launch of the persistent form is a set of workgroups that wait for one another; two launches fit the chip side by side, a third
    context on the device switches every context's NEXT updates to the launch-per-panel form (3pre_amd/csrc/pre3_cholp.hip, cholp_usable)"""
    seq = synth.make_sequence(30, 1, 8)
    fs = [pre3.EkfFilter(seq["cam"], np.zeros(30, np.int32), dtype="f32", max_hyp=8) for _ in range(2)]
    assert all(f.chol_persist() for f in fs)
    f3 = pre3.EkfFilter(seq["cam"], np.zeros(30, np.int32), dtype="f32", max_hyp=8)
    assert not f3.chol_persist() and not fs[0].chol_persist()
    f3.close()
    assert fs[0].chol_persist()
    for f in fs:
        f.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n_hi", [1, 5, 16, 17])
def test_small_hi_updates_against_the_twin(pre3, orc, dtype, n_hi):
    """ekf_update_hi_inliers with 1 .. 17 rescued measurements after an LI update (one panel with 2 .. 34 real rows: the lock-step kernel,
    rows built inside the H*P launch) against update.m's restatement in fp64 (numpy twin) applied twice -- LI set at x_k_km1, HI set with
    h / H re-evaluated at x_k_k (rescue_hi_inliers.m:29-47)"""
    from oracle import np_twin as tw
    N, n_hyp = 500, 8
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    meas, z = np.asarray(s["meas_idx"], np.int32), np.asarray(s["z"])
    inl = np.setdiff1d(np.arange(len(meas)), s["outliers"])
    li_pos, hi_pos = inl[:100], inl[100:100 + n_hi]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, std_z=1.0)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.ekf_prediction(s["u"])
    f.search_IC_matches()
    f.set_measurements(meas, z)
    li = np.zeros(len(meas), np.int32); li[li_pos] = 1
    f.set_flags(li=li)
    f.ekf_update_li_inliers()
    f.rescue_hi_inliers()                                            # h / H at x_k_k (its own gate result is overwritten below)
    hi = np.zeros(len(meas), np.int32); hi[hi_pos] = 1
    f.set_flags(hi=hi)
    f.ekf_update_hi_inliers()
    xg, Pg = f.get_x_k_k(), f.get_p_k_k()
    f.close()
    xr, Pr = tw.predict(seq["x0"], seq["P0"], s["u"])
    hh, has = tw.project(types, off, xr, seq["cam"])
    Hc, Hl = tw.jacobian(types, off, xr, seq["cam"], hh, has)
    zfull = np.zeros((N, 2)); zfull[meas] = z
    x1, P1 = tw.update_landmarks(types, off, meas[li_pos], xr, Pr, Hc, Hl, zfull, hh)
    h1, has1 = tw.project(types, off, x1, seq["cam"])
    Hc1, Hl1 = tw.jacobian(types, off, x1, seq["cam"], h1, has1)
    x2, P2 = tw.update_landmarks(types, off, meas[hi_pos], x1, P1, Hc1, Hl1, zfull, h1)
    tolP, tolx = (3e-4, 2e-5) if dtype == "f32" else (1e-9, 1e-9)
    assert np.isfinite(Pg).all()
    assert np.abs(Pg - P2).max() < tolP * np.abs(P2).max(), np.abs(Pg - P2).max() / np.abs(P2).max()
    assert np.abs(xg - x2).max() < tolx, np.abs(xg - x2).max()
    assert np.abs(P2 - P1).max() > 1e-6 * np.abs(P1).max()           # the HI update did something


def test_not_positive_definite_is_reported_by_a_one_panel_update(pre3):
    seq = synth.make_sequence(60, 1, 8)
    s = seq["steps"][0]
    for dtype in ("f32", "f64"):
        f = pre3.EkfFilter(seq["cam"], np.zeros(60, np.int32), dtype=dtype, max_hyp=8, std_z=1.0)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
        f.set_flags(li=np.zeros(len(s["meas_idx"]), np.int32))
        f.ekf_update_li_inliers()                                     # no inliers: x_k_k = x_k_km1
        f.rescue_hi_inliers()
        f.set_x_p_k_k(f.get_x_k_k(), -np.eye(seq["n"]) * 10.0)       # S = H P H' + R not positive definite
        hi = np.zeros(len(s["meas_idx"]), np.int32); hi[:6] = 1
        f.set_flags(hi=hi)
        f.ekf_update_hi_inliers()
        with pytest.raises(pre3.Pre3Error) as e:
            f.get_p_k_k()
        assert e.value.code == -5
        f.close()


@pytest.mark.parametrize("N,n_hyp,steps", [(30, 16, 3), (120, 60, 4), (500, 200, 4), (560, 100, 2)])
def test_downdate_consumers_inside_the_factorisation_are_bit_identical(pre3, N, n_hyp, steps):
    """update.m:37 accumulated panel by panel by consumer workgroups of k_cholp (PRE3_OPT_K9_OVERLAP, default) against the down-date as a
    launch of its own behind the factorisation: the same six bf16 products per k-stage in the same order, so x, P and the flags must agree
    bit for bit -- LI updates of 1 .. 10 panels, HI updates, deferred and not.  N = 560: more tile groups than idle CUs, the rest of the
    tiles goes through k_downdate_b3."""
    seq = synth.make_sequence(N, steps, n_hyp, seed=31 + N)
    res = []
    for on in (False, True):              # one filter at a time: with two fp32 contexts alive on the device the launches carry no consumers
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
        f.k9_overlap(on)
        f.step_tail(False)                # (the rescue stage + HI update as launches of their own in both runs: what is compared is WHERE the down-date runs)
        assert f.k9_overlap() == on and f.chol_persist()
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.defer_hi_update(bool(N % 2 == 0))
        sts = [f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0 if t % 2 == 0 else 0.5, early_exit=False) for t, s in enumerate(seq["steps"])]
        li, hi = f.get_flags()
        res.append((sts, li, hi, f.get_x_k_k(), f.get_p_k_k()))
        f.close()
    (st0, li0, hi0, x0, P0), (st1, li1, hi1, x1, P1) = res
    assert st0 == st1, (st0, st1)
    assert np.array_equal(li0, li1) and np.array_equal(hi0, hi1)
    assert np.array_equal(x0, x1)
    assert np.array_equal(P0, P1), np.abs(P0 - P1).max()
    assert np.isfinite(P1).all() and np.abs(P1).max() > 0


_TRAIL2_WORKER = r"""
import hashlib, importlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
N = 500
seq = synth.make_sequence(N, 1, 8); s = seq["steps"][0]
h = hashlib.sha256()
for m in (450, 500, 330):
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=8, std_z=1.0)
    f.chol_persist(False)
    f.set_x_p_k_k(seq["x0"], seq["P0"]); f.ekf_prediction(s["u"]); f.search_IC_matches()
    meas = np.arange(m, dtype=np.int32)
    f.set_measurements(meas, f.landmark_fields()["h"][:m] + 0.25)
    f.set_flags(li=np.ones(m, np.int32)); f.ekf_update_li_inliers()
    P = f.get_p_k_k(); assert np.isfinite(P).all()
    h.update(f.get_x_k_k().tobytes()); h.update(P.tobytes()); f.close()
print("DIGEST", h.hexdigest())
"""


def test_two_panel_trailing_sweep_is_bit_identical_to_the_one_panel_sweep():
    """launch-per-panel form with the trailing update as launches of its own (PRE3_CHOL_TRAIL_SPLIT=1 forces that at N = 500; it is N = 2000's form):
    the sweep over a group of 2 / 3 / 4 panels at a time (round 5: each tile read and written once per group, the panels applying what is pending for their own column)
    rounds and subtracts every product in the one-panel sweep's order -- updates of 15, 16 and 11 panels must give the same bits"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    # (group size, tile form of the W part: 2 = 128 x 128 super-tiles staged through LDS (default), 1 = super-tiles with fragments from memory, 0 = 64 x 64)
    for v, t in (("4", "2"), ("3", "2"), ("2", "2"), ("1", "2"), ("4", "1"), ("4", "0"), ("3", "1")):
        env = dict(os.environ, PRE3_CHOL_TRAIL_P=v, PRE3_CHOL_TRAIL_T128=t, PRE3_CHOL_TRAIL_SPLIT="1")
        r = subprocess.run([sys.executable, "-c", _TRAIL2_WORKER % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out[v, t] = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]
    assert all(d == out["1", "2"] for d in out.values()), out
