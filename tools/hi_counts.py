"""LI / HI landmark counts per step of the headline sequence (N = 500, threshold 1.0 px, motion noise 2.5): what the driver's 20-step window holds."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
N, H = 500, 200
seq = synth.make_sequence(N, 40, H, **{"motion_noise": synth.HEADLINE["motion_noise"]})
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"])
out = []
for s in seq["steps"]:
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
    out.append((st["n_li"], st["n_hi"]))
print(out)
