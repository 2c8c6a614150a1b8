"""Round 6, PRE3_OPT_PEND_HI: the HI update's covariance down-date (update.m:37-38 as ekf_update_hi_inliers.m:58 runs it) left pending across the step
boundary -- the next step's prediction transforms W~ with P, its H*P / S_i launch subtracts (H W~')W~, the consumers of its LI update's persistent
launch take W~ as the panels in front of panel 0.  Against the default form (the down-date as its own launch behind the update; that form is what the
twin / oracle tests pin) on the same chained steps: every step's statistics and inlier flags identical, the states equal to fp32 rounding -- the only
difference in arithmetic is that P between the HI update and the next LI update is never rounded to fp32.  Tolerances as tests/test_gpu_tail.py."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


def _run(pre3, seq, N, n_hyp, pend, z0=None, peek_after=None, dtype="f32"):
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, std_z=1.0)
    f.defer_hi_update(True)                              # (the pending form lives in the deferred path: the count is polled by the next call)
    assert f.pend_hi(pend) == (pend and dtype == "f32")
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    stats, peek = [], None
    for i, s in enumerate(seq["steps"]):
        z = z0 if (i == 0 and z0 is not None) else s["z"]
        st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False)
        stats.append((st["n_li"], st["max_support"], st["best"]))
        if peek_after == i:
            peek = (f.get_x_k_k(), f.get_p_k_k())          # any other call on the context completes the pending down-date first
    out = (f.get_flags(), f.get_x_k_k(), f.get_p_k_k(), stats, peek)
    f.close()
    return out


def _close(a, b, what):
    (fa, xa, Pa, sa, _), (fb, xb, Pb, sb, _) = a, b
    assert sa == sb, (what, sa, sb)
    assert all(np.array_equal(u, v) for u, v in zip(fa, fb)), what
    assert np.isfinite(Pa).all() and np.array_equal(Pa, Pa.T), what
    sc = np.abs(Pb).max()
    assert np.abs(Pa - Pb).max() < 3e-4 * sc, (what, np.abs(Pa - Pb).max() / sc)
    assert np.abs(xa - xb).max() < 2e-5, (what, np.abs(xa - xb).max())


@pytest.mark.parametrize("n_hi", [1, 18, 32, 33, 48, 64])
def test_chained_steps_with_a_pending_hi_downdate_agree_with_the_default_form(pre3, orc, n_hi):
    from oracle import np_twin as tw
    from test_gpu_tail import _with_n_rescued
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 4, n_hyp, motion_noise=2.5)
    types, off, n = orc.landmark_table(np.zeros(N, int))
    z0, ref = _with_n_rescued(tw, types, off, seq, seq["steps"][0], n_hi)      # step 0 rescues exactly n_hi landmarks (1 .. 32: one panel pending, 33 .. 64: two)
    a = _run(pre3, seq, N, n_hyp, True, z0)
    b = _run(pre3, seq, N, n_hyp, False, z0)
    _close(a, b, n_hi)


def test_another_call_between_the_steps_completes_the_pending_downdate(pre3):
    N, n_hyp = 500, 200
    seq = synth.make_sequence(N, 3, n_hyp, motion_noise=2.5)
    a = _run(pre3, seq, N, n_hyp, True, peek_after=0)
    b = _run(pre3, seq, N, n_hyp, False, peek_after=0)
    _close(a, b, "end")
    sc = np.abs(b[4][1]).max()
    assert np.abs(a[4][1] - b[4][1]).max() < 3e-4 * sc and np.abs(a[4][0] - b[4][0]).max() < 2e-5
    assert np.array_equal(a[4][1], a[4][1].T)


def test_fp64_contexts_take_the_option_without_effect(pre3):
    N, n_hyp = 120, 60
    seq = synth.make_sequence(N, 3, n_hyp, seed=5, motion_noise=2.5)
    a = _run(pre3, seq, N, n_hyp, True, dtype="f64")
    b = _run(pre3, seq, N, n_hyp, False, dtype="f64")
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_small_maps_and_steps_without_li_rows(pre3):
    """N = 120: one row workgroup, few consumers' groups; a step whose measurements are all gross outliers has no LI rows -- the persistent launch's
    consumers must still write the pending down-date (or the flush must)"""
    N, n_hyp = 120, 60
    seq = synth.make_sequence(N, 4, n_hyp, seed=77, motion_noise=2.5)
    seq["steps"][2]["z"] = np.array(seq["steps"][2]["z"], float) + 250.0 * np.sign(np.random.default_rng(3).standard_normal(np.array(seq["steps"][2]["z"]).shape))
    a = _run(pre3, seq, N, n_hyp, True)
    b = _run(pre3, seq, N, n_hyp, False)
    _close(a, b, "no-li")


@pytest.mark.parametrize("variant", [{}, {"PRE3_K9_OVERLAP": "0"}, {"PRE3_DD_MAX": "3"}, {"PRE3_HI_FUSED": "0"}, {"PRE3_GATE_RIDE": "0"}, {"PRE3_RIDE_INNOV": "0"},
                                     {"PRE3_HP_MB": "0"}, {"PRE3_CHOL_FORM": "0"}, {"PRE3_FUSE_JN": "0"}, {"TAIL": "1"}])
def test_launch_structures_that_cannot_take_the_pending_rows_flush_them(tmp_path, variant):
    """PRE3_PEND_HI=1 under the launch-structure switches of tests/test_gpu_variants.py: wherever a launch cannot take the pending rows along (the down-date as its own
    launch, too few consumer groups in the persistent launch, the launch-per-panel form, the stand-alone S_i pass, the one-measurement H*P kernel, the in-launch tail, the
    Jnorm pass as its own launch, the general HI path) the flush must run first.  The worker's sequences (a frame without LI rows, map management behind the steps) against
    the default form: statistics and flags identical, fp64 states equal to the bit (no pending form there), fp32 states to rounding."""
    import numpy as np
    from test_gpu_variants import _run
    v = dict(variant)
    tail = v.pop("TAIL", "0")
    a, b = str(tmp_path / "pend.npz"), str(tmp_path / "ref.npz")
    ra = _run(dict(v, PRE3_PEND_HI="1"), tail=tail, dump=a)
    rb = _run({}, tail="0", dump=b)
    assert ra["FLAGS"] == rb["FLAGS"], variant
    da, db = np.load(a), np.load(b)
    for k in da.files:
        if k.endswith("f64"):
            assert np.array_equal(da[k], db[k]), (variant, k)
        else:
            tol = 3e-4 * np.abs(db[k]).max() if k.startswith("P") else 2e-5
            assert np.abs(da[k] - db[k]).max() < tol, (variant, k, np.abs(da[k] - db[k]).max())
