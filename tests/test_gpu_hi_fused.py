"""mono_slam.m:184-187 -- rescue_hi_inliers + ekf_update_hi_inliers -- in the DEFAULT launch structure of pre3_step (k_hi_fused: the collection, the
rows, S = H P H' + I and the factorisation + solve of the HI update in one device-driven launch, its down-date behind it; pre3_update.hip) against the
numpy twin (fp64), with a chosen number of rescued landmarks: 1 .. 32 are one 64-row panel, 33 .. 64 two panels inside the same launch (round 5),
65 and more fall to the host's general path.  fp64 contexts always take the general path; they run the same cases.  Tolerances as everywhere for
one step at N = 500: fp32 3e-4 of P's scale / 2e-5 on x, fp64 1e-9."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def with_n_rescued(tw, types, off, seq, s, n_hi):
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
@pytest.mark.parametrize("n_hi", [1, 17, 18, 31, 32, 33, 34, 47, 48, 63, 64, 65, 80])
def test_whole_step_with_a_chosen_number_of_rescued_landmarks_matches_the_twin(pre3, orc, dtype, n_hi):
    from oracle import np_twin as tw
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    z, ref = with_n_rescued(tw, types, off, seq, s, n_hi)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, std_z=1.0)
    assert f.step_tail(False) is False
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
        assert np.array_equal(Pg, Pg.T)
    assert np.abs(Pg - ref["P_kk"]).max() < tolP * np.abs(ref["P_kk"]).max(), np.abs(Pg - ref["P_kk"]).max() / np.abs(ref["P_kk"]).max()
    assert np.abs(xg - ref["x_kk"]).max() < tolx, np.abs(xg - ref["x_kk"]).max()


_WORKER = r"""
import importlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
import oracle as orc
from oracle import np_twin as tw
sys.path.insert(0, %(root)r + "/tests")
N, n_hyp = 500, 200
seq = synth.make_sequence(N, 1, n_hyp); s = seq["steps"][0]
types, off, n = orc.landmark_table(np.zeros(N, int))
out = {}
for n_hi in (33, 48, 64):
    z = np.array(s["z"], float)
    ref = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], z, s["hyp"], 1.0, early_exit=False)
    hi_pos = np.nonzero(ref["hi"])[0]
    z[hi_pos[n_hi:]] += 300.0
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
    f.step_tail(False)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False)
    np.save(sys.argv[1] + "_%%d.npy" %% n_hi, f.get_p_k_k()); np.save(sys.argv[1] + "_x%%d.npy" %% n_hi, f.get_x_k_k())
    out[n_hi] = st["n_hi"]
    f.close()
print(json.dumps(out))
"""


def test_two_panels_inside_the_launch_agree_with_the_general_path(pre3, tmp_path):
    """33 .. 64 rescued landmarks: k_hi_fused's two-panel path (explicit M0 = L00^-1, products on the f32 matrix cores) against the host-polled general
    path it replaces (PRE3_HI_FUSED_TWO=0: k_build_rows -> k_ell_HP -> k_ell_G -> k_cholp -> k_downdate_b3) -- two fp32 evaluations of the same update,
    a few ulps of P's scale apart"""
    if len(seq_rescued()) < 64:
        pytest.skip("the sequence rescues fewer than 64 landmarks")
    res = {}
    for two in ("1", "0"):
        env = dict(os.environ, PRE3_HI_FUSED_TWO=two)
        base = str(tmp_path / ("two" + two))
        r = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}, base], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[two] = {k: (np.load(base + "_%d.npy" % k), np.load(base + "_x%d.npy" % k)) for k in (33, 48, 64)}
    for k in (33, 48, 64):
        (P1, x1), (P0, x0) = res["1"][k], res["0"][k]
        scale = np.abs(P0).max()
        assert np.abs(P1 - P0).max() < 2e-5 * scale, (k, np.abs(P1 - P0).max() / scale)
        assert np.abs(x1 - x0).max() < 2e-6, (k, np.abs(x1 - x0).max())
        assert not np.array_equal(P1, P0)                            # (the two runs did take different paths)


def seq_rescued():
    import oracle as orc
    from oracle import np_twin as tw
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    ref = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], np.array(s["z"], float), s["hyp"], 1.0, early_exit=False)
    return np.nonzero(ref["hi"])[0]
