"""Soak of PRE3_OPT_PEND_HI (round 6): many chained steps with the HI update's down-date left pending (deferred HI completion); the state must stay finite and
symmetric, a second identical run must reproduce it bit for bit (the in-launch hand-offs of k_hi_fused's x-update and the consumers' extra panels are
deterministic), and it must stay within fp32 rounding of the default form's."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")

def run(N, steps, n_hyp, pend, noise):
    seq = synth.make_sequence(N, steps, n_hyp, seed=1000 + N, motion_noise=noise)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.defer_hi_update(True)
    assert f.pend_hi(pend) == pend
    nli = nhi = 0
    for s in seq["steps"]:
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        nli += st["n_li"]; nhi += st["n_hi"]
    x, P = f.get_x_k_k(), f.get_p_k_k()
    f.close()
    return x, P, nli, nhi

for N, steps, n_hyp, noise in ((500, 400, 200, 2.5), (63, 600, 64, 2.5), (7, 800, 5, 2.5), (200, 400, 100, 0.5)):
    t = time.perf_counter()
    a, b, c = run(N, steps, n_hyp, True, noise), run(N, steps, n_hyp, True, noise), run(N, steps, n_hyp, False, noise)
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    sc = np.abs(c[1]).max()
    dP, dx = np.abs(a[1] - c[1]).max() / sc, np.abs(a[0] - c[0]).max()
    ok = np.isfinite(a[1]).all() and np.array_equal(a[1], a[1].T) and same
    print("N=%d %d steps: %s (mean LI %.1f HI %.1f; against the default form: P %.2e of its scale, x %.2e; LI counts %s) %.1f s"
          % (N, steps, "OK" if ok else "MISMATCH", a[2] / steps, a[3] / steps, dP, dx, "equal" if a[2] == c[2] else "%d vs %d" % (a[2], c[2]), time.perf_counter() - t), flush=True)
    assert ok
