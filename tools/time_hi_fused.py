"""k_hi_fused and the launches around it for a chosen number of rescued landmarks (run under rocprofv3 --kernel-trace --stats):
python3 tools/time_hi_fused.py 20 40 64  -> each count: 30 whole steps from the same state."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
import oracle as orc
from oracle import np_twin as tw
N, n_hyp = 500, 200
seq = synth.make_sequence(N, 1, n_hyp); s = seq["steps"][0]
types, off, n = orc.landmark_table(np.zeros(N, int))
z0 = np.array(s["z"], float)
ref = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], z0, s["hyp"], 1.0, early_exit=False)
hi_pos = np.nonzero(ref["hi"])[0]
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
f.step_tail(False)
for n_hi in [int(a) for a in sys.argv[1:]]:
    z = z0.copy(); z[hi_pos[n_hi:]] += 300.0
    ts = []
    for it in range(30):
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.get_x_k_k()
        t0 = time.perf_counter()
        st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False)
        f.get_flags()
        ts.append(time.perf_counter() - t0)
    print("n_hi %d (device says %d): median step wall %.1f us" % (n_hi, st["n_hi"], 1e6 * float(np.median(ts))), flush=True)
f.close()
