"""Does down-date work running beside the factorisation slow its chain?  One filter steps (thread A) while a second context on the same GPU
runs rank-r down-dates of its own P back to back (thread B, r = argv[1], 0 = nothing).  Prints steps/s of A; run under
rocprofv3 --kernel-trace to compare the k_chol_step launch durations with and without the neighbour."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
r_nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64          # > 0: down-date neighbour of that rank; < 0: streaming neighbour (tools/probe_neighbour.hip)
lds_kb, wgs, mf = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0, 0)
N, K, W, H = 500, 150, 10, 200
seq = synth.make_sequence(N, K + W, H, seed=900)
mk = lambda: pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f, g = mk(), mk()
for x in (f, g):
    x.set_x_p_k_k(seq["x0"], seq["P0"]); x.defer_hi_update(True)
for s in seq["steps"][:W]:
    f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
    g.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
f.sync(); g.sync()
stop = False
import ctypes
nb = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnb.so")) if r_nb < 0 else None
def neighbour():
    while not stop:
        if r_nb > 0: g.bench_downdate(r_nb, 20)
        else: nb.nb_run(20, lds_kb, wgs, mf)
th = threading.Thread(target=neighbour)
if r_nb != 0: th.start()
time.sleep(0.05)
t0 = time.perf_counter()
for s in seq["steps"][W:W + K]:
    f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
f.sync()
el = time.perf_counter() - t0
stop = True
if r_nb != 0: th.join()
print("neighbour %s: %.0f steps/s" % (" ".join(sys.argv[1:]), K / el), flush=True)
