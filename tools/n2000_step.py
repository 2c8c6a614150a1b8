"""A few whole steps at BASELINE configs[4]'s size on one GPU (N=2000, n=12013, 1000 hypotheses): for rocprofv3 --kernel-trace --stats (where the step's time goes)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
N, H, steps = 2000, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 5
seq = synth.make_sequence(N, steps + 1, H)
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"]); f.defer_hi_update(True)
s = seq["steps"][0]; f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False); f.sync()
t0 = time.perf_counter()
st = [f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False) for s in seq["steps"][1:]]
f.sync()
el = time.perf_counter() - t0
print("N=2000 full step: %.2f ms/step (%.1f steps/s), LI rows %s HI rows %s" % (1e3 * el / steps, steps / el, [2 * x["n_li"] for x in st], [2 * x["n_hi"] for x in st]), flush=True)
f.close()
