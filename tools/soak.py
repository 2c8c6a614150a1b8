"""Soak: many steps at several sizes / dtypes (the in-launch counters, riders and last-workgroup tails must never hang or drift).
Checks that the state stays finite and symmetric and that a second, identical run reproduces it bit for bit."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")

def run(N, steps, n_hyp, dtype, defer):
    seq = synth.make_sequence(N, steps, n_hyp, seed=1000 + N)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, std_z=1.0)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.defer_hi_update(defer)
    nli = 0
    for s in seq["steps"]:
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=bool(N % 2))
        nli += st["n_li"]
    x, P = f.get_x_k_k(), f.get_p_k_k()
    f.close()
    return x, P, nli

for N, steps, n_hyp, dtype in ((500, 600, 200, "f32"), (200, 800, 200, "f64"), (40, 1500, 30, "f64"), (7, 1500, 5, "f32"), (63, 800, 64, "f32"), (129, 500, 100, "f64")):
    t = time.perf_counter()
    a = run(N, steps, n_hyp, dtype, True)
    b = run(N, steps, n_hyp, dtype, False)
    ok = np.isfinite(a[0]).all() and np.isfinite(a[1]).all() and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    print("N=%d %s %d steps x2: %s (mean LI %.1f, max|P-P'| %.2e) %.1f s" % (N, dtype, steps, "OK" if ok else "MISMATCH", a[2] / steps, np.abs(a[1] - a[1].T).max(), time.perf_counter() - t), flush=True)
    assert ok
