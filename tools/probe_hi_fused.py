"""Phases of ONE k_hi_fused launch (workgroup 5) from its device-side wall-clock stamps (probe build):
PRE3_LIB=3pre_amd/lib/libpre3_probe.so python tools/probe_hi_fused.py 8 20 32"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PRE3_LIB", "3pre_amd/lib/libpre3_probe.so")
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
lib = importlib.import_module("3pre_amd._lib").lib
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
fn = lib.pre3_debug_hf; fn.restype = C.c_int
names = ["collection (flags, rows, list)", "padding rows, ELL export", "T", "own block of H*P", "S entries", "panel body (loads, chain, stores)"]
for n_hi in [int(a) for a in sys.argv[1:]] or [8, 20, 32]:
    z = z0.copy(); z[hi_pos[n_hi:]] += 300.0
    for it in range(4):
        f.set_x_p_k_k(seq["x0"], seq["P0"]); f.get_x_k_k()
        st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=False); f.get_flags()
    buf = np.zeros(16, np.uint64)
    assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.astype(np.float64) / 100.0
    print("n_hi %d: total %.2f us | " % (st["n_hi"], t[6] - t[0]) + ", ".join("%s %.2f" % (names[k], t[k + 1] - t[k]) for k in range(6)), flush=True)
f.close()
