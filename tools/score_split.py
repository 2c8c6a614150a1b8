"""k_ransac_score at the headline size (N=500, 400 measurements, 200 hypotheses) with and without the selection stage in its last workgroup:
run under rocprofv3 --kernel-trace and compare the durations (the un-fused form also shows k_ransac_select's own time)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
N, H = 500, 200
seq = synth.make_sequence(N, 3, H, seed=900)
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"])
s = seq["steps"][0]
f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
for _ in range(30):
    f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)          # fused: scoring + selection in one launch
for _ in range(30):
    f.ransac_score_shard(s["hyp"], 1.0, 0, H)                                # scoring only (full range)
    f.ransac_select(H, 3, False)                                             # selection as its own launch
f.sync()
print("done")
