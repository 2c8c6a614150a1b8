import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
for N, steps, reps, nh, dt in ((500, 300, 60, 200, "f32"), (40, 300, 200, 30, "f64"), (200, 300, 60, 100, "f64")):
    seq = synth.make_sequence(N, steps, nh, seed=7 + N)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dt, max_hyp=nh, std_z=1.0)
    f.defer_hi_update(True)
    t = time.perf_counter(); ref = None
    for r in range(reps):
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        for s in seq["steps"]:
            f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        x = f.get_x_k_k()
        if ref is None: ref = x
        assert np.array_equal(x, ref), "run %d differs" % r
    print("N=%d %s: %d steps, all %d repetitions bit-identical, %.1f s" % (N, dt, steps * reps, reps, time.perf_counter() - t), flush=True)
    f.close()
