"""Time the device map-management operations at the bench configuration (N=500)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for dtype in ("f32", "f64"):
    seq = synth.make_sequence(N, 1, 8, seed=1)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=8, max_landmarks=N + 16)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    uvd = np.stack([np.linspace(10, 160, 10), np.linspace(10, 130, 10)], 1)
    ts = {"add10": [], "del10": [], "conv": []}
    for it in range(6):
        f.sync(); t = time.perf_counter(); f.add_features_inverse_depth(uvd, 1.0, 0.5); f.sync(); ts["add10"].append(time.perf_counter() - t)
        t = time.perf_counter(); f.delete_features(range(N, N + 10)); f.sync(); ts["del10"].append(time.perf_counter() - t)
    t = time.perf_counter(); c = f.inversedepth_2_cartesian(1e9); f.sync(); ts["conv"].append(time.perf_counter() - t)
    print(dtype, "n=%d" % (13 + 6 * N), {k: "%.0f us" % (1e6 * min(v)) for k, v in ts.items()}, "converted", int(c.sum()), flush=True)
    f.close()
