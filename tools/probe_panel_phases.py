"""Phase clock of one panel workgroup (blockIdx 5) of the last k_chol_step launch, from a -DPRE3_PROBE build:
make -C 3pre_amd/csrc probe && PRE3_LIB=3pre_amd/lib/libpre3_probe.so python tools/probe_panel_phases.py"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
lib = pre3._lib.lib
N = 500
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
seq = synth.make_sequence(N, 4, 200, seed=None)
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dt, max_hyp=200, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"])
for s in seq["steps"][:3]:
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
f.sync()
buf = (C.c_ulonglong * 16)()
lib.pre3_debug_probe(buf)
a = np.array(buf[:], dtype=np.int64)
print("n_li", st["n_li"], "stamps", a[:5])
print("loads -> LDS            %6d ticks" % (a[4] - a[0]))
print("prologue (MFMA + RMW)   %6d ticks" % (a[1] - a[4]))
print("64-column chain         %6d ticks" % (a[2] - a[1]))
print("store                   %6d ticks" % (a[3] - a[2]))
print("total                   %6d ticks" % (a[3] - a[0]))
buf2 = (C.c_ulonglong * (64 * 8 * 4))()
lib.pre3_debug_k9_stamps(buf2)
b = np.array(buf2[:128], dtype=np.int64).reshape(16, 8)
print("step k: work ticks of factor | z | worker wave 2, step length (factor wave start -> next start)")
for k in range(-1, 9):
    r = b[k + 1]
    nxt = b[k + 2][0] if k < 8 else a[2]
    print("  k=%2d  %5d | %5d | %5d   step %5d" % (k, r[1] - r[0], r[3] - r[2], r[5] - r[4], nxt - r[0]))
