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


def _convert(tw, x, P, types):
    """inversedepth_2_cartesian.m:49-70 for the landmarks with types[i] == 1 (an all-inverse-depth map in): the point and its 3 x 6 Jacobian per landmark,
    P <- A P A' with the sparse A those blocks make up"""
    import scipy.sparse as sp
    N = len(types)
    xs, rows, cols, vals, o_new = [x[:13]], list(range(13)), list(range(13)), [1.0] * 13, 13
    for i in range(N):
        o = 13 + 6 * i
        if types[i] == 0:
            xs.append(x[o:o + 6])
            rows += list(range(o_new, o_new + 6)); cols += list(range(o, o + 6)); vals += [1.0] * 6
            o_new += 6
            continue
        rho, theta, phi = x[o + 5], x[o + 3], x[o + 4]
        mi = tw.m_dir(theta, phi)
        xs.append(x[o:o + 3] + mi / rho)
        J = np.hstack([np.eye(3), (np.array([np.cos(phi) * np.cos(theta), 0, -np.cos(phi) * np.sin(theta)]) / rho)[:, None],
                       (np.array([-np.sin(phi) * np.sin(theta), -np.cos(phi), -np.sin(phi) * np.cos(theta)]) / rho)[:, None], (-mi / rho ** 2)[:, None]])
        for a in range(3):
            for b in range(6):
                rows.append(o_new + a); cols.append(o + b); vals.append(J[a, b])
        o_new += 3
    A = sp.csr_matrix((vals, (rows, cols)), shape=(o_new, len(x)))
    Pn = (A @ (A @ P).T).T
    return np.concatenate(xs), 0.5 * (Pn + Pn.T)


@pytest.mark.parametrize("n_hi", [20, 40])
def test_mixed_cartesian_and_inverse_depth_landmarks_through_the_fused_hi_update(pre3, orc, n_hi):
    """a map in which part of the landmarks has been converted to Cartesian (inversedepth_2_cartesian.m:27-76 on the twin): their rows of H have three
    landmark columns, the other three ELL slots are zero entries at column 0 -- one panel (20 rescued landmarks) and two panels (40) of k_hi_fused"""
    from oracle import np_twin as tw
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types = np.zeros(N, np.int32); types[::3] = 1                 # every third landmark Cartesian
    x0, P0 = _convert(tw, seq["x0"], seq["P0"], types)
    tt, off, n = orc.landmark_table(types)
    z = np.array(s["z"], float)
    ref = tw.step(tt, off, seq["cam"], x0, P0, s["u"], s["meas_idx"], z, s["hyp"], 1.0, early_exit=False)
    hi_pos = np.nonzero(ref["hi"])[0]
    if len(hi_pos) < n_hi:
        pytest.skip("the sequence rescues only %d landmarks" % len(hi_pos))
    z[hi_pos[n_hi:]] += 300.0
    ref = tw.step(tt, off, seq["cam"], x0, P0, s["u"], s["meas_idx"], z, s["hyp"], 1.0, early_exit=False)
    assert int(ref["hi"].sum()) == n_hi and np.any(types[np.asarray(s["meas_idx"])[ref["hi"] != 0]] != 0)       # (Cartesian landmarks among the rescued)
    f = pre3.EkfFilter(seq["cam"], types, dtype="f32", max_hyp=n_hyp, std_z=1.0)
    f.step_tail(False)
    f.set_x_p_k_k(x0, P0)
    st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False)
    li, hi = f.get_flags()
    xg, Pg = f.get_x_k_k(), f.get_p_k_k()
    f.close()
    assert np.array_equal(li, ref["li"]) and np.array_equal(hi, ref["hi"]) and st["n_hi"] == n_hi
    assert np.abs(Pg - ref["P_kk"]).max() < 3e-4 * np.abs(ref["P_kk"]).max(), np.abs(Pg - ref["P_kk"]).max() / np.abs(ref["P_kk"]).max()
    assert np.abs(xg - ref["x_kk"]).max() < 2e-5, np.abs(xg - ref["x_kk"]).max()


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


def test_s_dealt_over_the_workgroups_is_bit_identical_to_s_built_by_every_one(pre3, tmp_path):
    """round 6, two panels: S = H P H' + I dealt over the launch's workgroups (hf_S_dealt: write-through (sequence, value) pairs) against every workgroup
    building all of it (PRE3_HF_DEAL=0): the same fma chains entry by entry -- the same bits of x and P"""
    if len(seq_rescued()) < 64:
        pytest.skip("the sequence rescues fewer than 64 landmarks")
    res = {}
    for deal in ("1", "0"):
        env = dict(os.environ, PRE3_HF_DEAL=deal)
        base = str(tmp_path / ("deal" + deal))
        r = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}, base], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[deal] = {k: (np.load(base + "_%d.npy" % k), np.load(base + "_x%d.npy" % k)) for k in (33, 48, 64)}
    for k in (33, 48, 64):
        assert np.array_equal(res["1"][k][0], res["0"][k][0]) and np.array_equal(res["1"][k][1], res["0"][k][1]), k


def seq_rescued():
    import oracle as orc
    from oracle import np_twin as tw
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    ref = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], np.array(s["z"], float), s["hyp"], 1.0, early_exit=False)
    return np.nonzero(ref["hi"])[0]
