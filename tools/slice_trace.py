"""The 8-rank slice of a sharded RANSAC round at BASELINE configs[4] (N=2000, 1000 hypotheses -> 125 per rank) on one GPU, for rocprofv3
--kernel-trace: which launches the per-rank compute consists of (pre3_ransac_score on hypotheses [0, 125))."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
N, n_hyp, G = 2000, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
seq = synth.make_sequence(N, 1, n_hyp)
s = seq["steps"][0]
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp)
f.set_x_p_k_k(seq["x0"], seq["P0"])
f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
hi = n_hyp // G
for _ in range(3): f.ransac_score_shard(s["hyp"], 1.0, 0, hi)
f.sync()
t0 = time.perf_counter()
for _ in range(20): f.ransac_score_shard(s["hyp"], 1.0, 0, hi)
el = (time.perf_counter() - t0) / 20
print("G=%d: %d hypotheses per rank, %.1f us per rank and round" % (G, hi, el * 1e6), flush=True)
f.close()
