"""Several independent filters on ONE GPU (one context + stream + host thread each): the step is a latency chain that leaves most of the
chip idle, so concurrent filters overlap.  Reports aggregate steps/s; NOT the headline metric (that is one filter)."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
N, K, W, H = 500, 150, 10, 200
seqs = {}
for R in (1, 2, 3, 4, 6):
    fs = []
    for r in range(R):
        if r not in seqs:
            seqs[r] = synth.make_sequence(N, K + W, H, seed=900 + r)
        seq = seqs[r]
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
        f.set_x_p_k_k(seq["x0"], seq["P0"]); f.defer_hi_update(True)
        for s in seq["steps"][:W]:
            f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        f.sync(); fs.append((f, seq))
    bar = threading.Barrier(R + 1)
    def work(f, seq):
        bar.wait()
        for s in seq["steps"][W:W + K]:
            f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        f.sync()
    th = [threading.Thread(target=work, args=fs[r]) for r in range(R)]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("%d filters on one GPU: %.0f steps/s aggregate (%.0f per filter)" % (R, R * K / el, K / el), flush=True)
    for f, _ in fs: f.close()
