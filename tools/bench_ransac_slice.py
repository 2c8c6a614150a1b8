"""What one rank of G does in a sharded RANSAC round at BASELINE configs[4] (N=2000, 1000 hypotheses): pre3_ransac_score on 1/G of the
draw table (partial H*P / H*P*H' for the measurements that slice draws + scoring), one GPU, no collective -- the per-rank compute that the
1 -> G scaling of bench.py's ransac_shard leg is made of."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
N, n_hyp = 2000, 1000
seq = synth.make_sequence(N, 1, n_hyp)
s = seq["steps"][0]
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp)
f.set_x_p_k_k(seq["x0"], seq["P0"])
f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
base = None
for G in (1, 2, 4, 8):
    hi = n_hyp // G
    for _ in range(3): f.ransac_score_shard(s["hyp"], 1.0, 0, hi)
    t0 = time.perf_counter()
    for _ in range(20): f.ransac_score_shard(s["hyp"], 1.0, 0, hi)
    el = (time.perf_counter() - t0) / 20
    base = base or el
    print("G=%d: %4d hypotheses per rank, %.0f us per rank and round (x%.2f of G=1)" % (G, hi, el * 1e6, base / el))
f.close()
