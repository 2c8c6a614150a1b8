"""Round 5: mono_slam.m:184-187 -- rescue_hi_inliers + ekf_update_hi_inliers -- inside the LI update's persistent launch (pre3_cholp.hip, CpTail):
whole steps through pre3_step against the numpy twin (fp64) with a chosen number of rescued landmarks.  Up to 32 the launch itself updates with
them (their rows are panel nrb of the same block factorisation, P is written once); 33 and more take the host's general path behind the same
in-launch gate.  Tolerances as everywhere for one step at N = 500: fp32 3e-4 of P's scale / 2e-5 on x, fp64 1e-9."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


def _with_n_rescued(tw, types, off, seq, s, n_hi):
    """the step's measurements with all but n_hi of the twin's rescued landmarks turned into gross outliers (a rescue candidate is not a row of the
    LI update, and each gate is per landmark: the others' outcome does not change)"""
    z = np.array(s["z"], float)
    ref = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], z, s["hyp"], 1.0, early_exit=False)
    hi_pos = np.nonzero(ref["hi"])[0]
    if len(hi_pos) < n_hi:
        pytest.skip("the sequence rescues only %d landmarks" % len(hi_pos))
    z[hi_pos[n_hi:]] += 300.0
    ref2 = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], z, s["hyp"], 1.0, early_exit=False)
    assert np.array_equal(ref2["li"], ref["li"]) and int(ref2["hi"].sum()) == n_hi, (int(ref2["hi"].sum()), n_hi)
    return z, ref2


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n_hi", [1, 18, 31, 32, 33, 48, 64])
def test_whole_step_with_a_chosen_number_of_rescued_landmarks_matches_the_twin(pre3, orc, dtype, n_hi):
    from oracle import np_twin as tw
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    z, ref = _with_n_rescued(tw, types, off, seq, s, n_hi)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, std_z=1.0)
    assert f.step_tail(True) == (dtype == "f32")                  # (off by default: DESIGN.md section 5d; fp64 contexts have no such form)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False)
    li, hi = f.get_flags()
    xg, Pg = f.get_x_k_k(), f.get_p_k_k()
    f.close()
    assert np.array_equal(li, ref["li"]) and np.array_equal(hi, ref["hi"])
    assert st["n_hi"] == n_hi
    tolP, tolx = (3e-4, 2e-5) if dtype == "f32" else (1e-9, 1e-9)
    assert np.isfinite(Pg).all()
    if dtype == "f32":
        assert np.array_equal(Pg, Pg.T)                          # (the consumers write each tile and its mirror image from the same registers)
    assert np.abs(Pg - ref["P_kk"]).max() < tolP * np.abs(ref["P_kk"]).max(), np.abs(Pg - ref["P_kk"]).max() / np.abs(ref["P_kk"]).max()
    assert np.abs(xg - ref["x_kk"]).max() < tolx, np.abs(xg - ref["x_kk"]).max()


def test_deferred_and_immediate_hi_completion_agree_with_the_tail(pre3):
    """PRE3_OPT_DEFER_HI only moves the host's poll of the count (and the pending rows / columns 3..6 pass into the next prediction's launch):
    with the tail inside the persistent launch the states of a sequence must still be the same bits either way"""
    N, n_hyp = 120, 60
    seq = synth.make_sequence(N, 6, n_hyp, seed=77, motion_noise=2.5)
    res = []
    for defer in (False, True):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
        f.defer_hi_update(defer)
        assert f.step_tail(True)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        fl = []
        for s in seq["steps"]:
            f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
            fl.append(tuple(a.tobytes() for a in f.get_flags()) if not defer else None)
        res.append((f.get_flags(), f.get_x_k_k(), f.get_p_k_k()))
        f.close()
    assert all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0]))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
