"""Time the VO RANSAC and the fused IC-search stage (8(f)-2) on the device, with the oracle beside them."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
oracle.build()
pre3 = importlib.import_module("3pre_amd")
vo = importlib.import_module("3pre_amd.vo")
synth = importlib.import_module("3pre_amd.synth")
from test_vo_oracle import scene
from test_gpu_icsearch import _scene, _scan

for pnum in (150, 500, 2000):
    rng, R, T, p1, p2, match, bad = scene(pnum, 3)
    draws = vo.draw_hypotheses(match, 700, rng)
    ms = vo.vo_bench(p1, p2, draws, reps=50)
    t = time.perf_counter(); out = vo.vo_ransac(p1, p2, draws); wall = time.perf_counter() - t
    t = time.perf_counter(); ref = oracle.vo_ransac(p1, p2, draws); cpu = time.perf_counter() - t
    print("vo pnum=%d hyp=700: kernels %.1f us, call incl. transfers %.0f us, oracle (1 core) %.0f us, same winner %s" %
          (pnum, ms * 1e3, wall * 1e6, cpu * 1e6, out["best"] == ref["best"]), flush=True)

for N, K2 in ((500, 600),):
    rng, seq, bank = _scene(N, 23)
    s = seq["steps"][0]
    types, off, n = oracle.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f32", max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"]); f.set_descriptors(bank)
    f.ekf_prediction(s["u"])
    x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()
    h, has_h = oracle.project(types, off, x1, seq["cam"])
    sd, sp = _scan(rng, h, has_h, bank, K2 - int(0.8 * N))
    ts = []
    for it in range(5):
        f.set_x_p_k_k(seq["x0"], seq["P0"]); f.set_descriptors(bank); f.ekf_prediction(s["u"]); f.sync()
        t = time.perf_counter(); f.load_scan(sd, sp); t1 = time.perf_counter(); out = f.matching_sift_based(); f.sync(); ts.append((t1 - t, time.perf_counter() - t1))
    t = time.perf_counter(); ref = oracle.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp); cpu = time.perf_counter() - t
    print("ic_search N=%d K2=%d: scan upload %.0f us, search %.0f us (m=%d), oracle (1 core) %.1f ms" %
          (N, sd.shape[1], 1e6 * min(a for a, b in ts), 1e6 * min(b for a, b in ts), len(out["meas_idx"]), cpu * 1e3), flush=True)
    f.close()
