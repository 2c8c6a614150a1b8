"""Timeline of the tail of ONE persistent launch of a pre3_step (rescue stage + HI update inside k_cholp, round 5) from device-side wall-clock stamps.
usage (GPU box, probe build: make -C 3pre_amd/csrc probe): PRE3_LIB=3pre_amd/lib/libpre3_probe.so python tools/probe_tail.py [warm steps]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("PRE3_LIB", "3pre_amd/lib/libpre3_probe.so")
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
lib = importlib.import_module("3pre_amd._lib").lib
N, H = 500, 200
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seq = synth.make_sequence(N, warm + 1, H, motion_noise=synth.HEADLINE["motion_noise"])
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"])
for s in seq["steps"]:
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
f.sync()
buf = np.zeros(24 * 16 * 8, np.uint64)
fn = lib.pre3_debug_cholp; fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(24, 16, 8).astype(np.float64) / 100.0      # us
n_li, n_hi = st["n_li"], st["n_hi"]; nrb = (2 * n_li + 63) // 64
t0 = t[0, 0, 0]
def rel(x): return "%7.2f" % (x - t0) if x > 0 else "      -"
print("last step: n_li %d (%d panels), n_hi %d; us after crit's first chain start" % (n_li, nrb, n_hi))
print("crit: last LI chain end %s | HI chain start %s end %s" % (rel(t[0, nrb - 1, 1]), rel(t[0, nrb, 0]) if nrb < 16 else "-", rel(t[0, nrb, 1]) if nrb < 16 else "-"))
print("strip 1: last panel (J=%d) M seen %s, stored+planes %s | x-update start %s, x flag %s" % (nrb - 1, rel(t[16, nrb - 1, 1]), rel(t[16, nrb - 1, 2]), rel(t[23, 2, 0]), rel(t[23, 2, 1])))
print("strip 1 tail: enter %s | gate: flags seen %s projected %s y done %s drained %s | gate flag %s | list seen %s | C summed %s | M_hi seen %s | W~ flag %s | x done %s" %
      (rel(t[23, 0, 0]), rel(t[23, 1, 0]), rel(t[23, 1, 1]), rel(t[23, 1, 2]), rel(t[23, 1, 3]), rel(t[23, 0, 1]), rel(t[23, 0, 2]), rel(t[23, 0, 3]), rel(t[23, 0, 4]), rel(t[23, 0, 5]), rel(t[23, 0, 6])))
print("crit tail: enter %s | all gate flags %s | list %s | list flag out %s | T done %s | Y Y' done %s | D built %s" % tuple(rel(t[22, 0, k]) for k in range(7)))
print("consumers (first / last group): last LI panel flags seen %s, MFMAs done %s | %s %s" % (rel(t[20, nrb - 1, 0]), rel(t[20, nrb - 1, 1]), rel(t[21, nrb - 1, 0]), rel(t[21, nrb - 1, 1])))
if nrb < 16:
    print("  HI panel: flags seen %s MFMAs done %s | %s %s" % (rel(t[20, nrb, 0]), rel(t[20, nrb, 1]), rel(t[21, nrb, 0]), rel(t[21, nrb, 1])))
print("  epilogue start / end: %s %s | %s %s" % (rel(t[20, 15, 0]), rel(t[20, 15, 1]), rel(t[21, 15, 0]), rel(t[21, 15, 1])))
