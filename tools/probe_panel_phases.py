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
for n_, s in enumerate(seq["steps"][:3]):
    if n_ == 2:
        f.sync(); lib.pre3_debug_rt(1)
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
    if n_ == 2:
        f.sync(); lib.pre3_debug_rt(0)
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
pj = np.array(buf2[256:256 + 80], dtype=np.int64).reshape(10, 8)
print("per panel J (workgroup %d): loads | prologue | chain | store ticks" % 5)
for J in range(10):
    r = pj[J]
    print("  J=%d  %6d | %6d | %6d | %6d" % (J, r[4] - r[0], r[1] - r[4], r[2] - r[1], r[3] - r[2]))
print("step k: work ticks of factor | z | worker wave 2, step length (factor wave start -> next start)")
for k in range(-1, 9):
    r = b[k + 1]
    nxt = b[k + 2][0] if k < 8 else a[2]
    print("  k=%2d  %5d | %5d | %5d   step %5d" % (k, r[1] - r[0], r[3] - r[2], r[5] - r[4], nxt - r[0]))
f3 = np.array(buf2[100:105], dtype=np.int64)
print("factor wave, step k=3: lookahead (readlane) %d | Pn read + subtract %d | 8 columns %d | writes %d   (each incl. one stamp)" % tuple(np.diff(f3)))
rt = (C.c_ulonglong * (2048 * 4))()
lib.pre3_debug_k9rt(rt)
r = np.array(rt[:4096], dtype=np.int64).reshape(1024, 4)
ok = r[:, 1] > r[:, 0]
if ok.any():
    t0 = r[ok, 0].min()
    nP = 1 + 7 + 49
    print("launch of panel 2 (wall clock, us from the first workgroup's start): %d workgroups stamped, launch span %.2f" % (ok.sum(), (r[ok, 1].max() - t0) / 100.0))
    for name, sel in (("panel workgroups", np.arange(1024) < nP), ("trailing tiles", np.arange(1024) >= nP)):
        m = ok & sel
        if m.any():
            print("  %-18s n=%4d  start %.2f..%.2f  end %.2f..%.2f  duration median %.2f max %.2f" % (name, m.sum(), (r[m, 0].min() - t0) / 100.0, (r[m, 0].max() - t0) / 100.0,
                  (r[m, 1].min() - t0) / 100.0, (r[m, 1].max() - t0) / 100.0, np.median(r[m, 1] - r[m, 0]) / 100.0, (r[m, 1] - r[m, 0]).max() / 100.0))
    pe = (r[:nP, 1] - t0) / 100.0
    print("  panel workgroup ends: b=0 %.2f, S blocks %s, W strips min %.2f median %.2f max %.2f (b=%d)" % (pe[0], np.round(pe[1:8], 2), pe[8:].min(), np.median(pe[8:]), pe[8:].max(), 8 + int(pe[8:].argmax())))
    d = r[:, 3] - r[:, 2]
    print("  shader-clock durations (cycles): panel workgroups median %d max %d; trailing tiles median %d max %d; realtime ticks: panel median %d, tiles median %d" % (
        np.median(d[:nP][ok[:nP]]), d[:nP][ok[:nP]].max(), np.median(d[nP:][ok[nP:]]), d[nP:][ok[nP:]].max(), np.median((r[:, 1] - r[:, 0])[:nP][ok[:nP]]), np.median((r[:, 1] - r[:, 0])[nP:][ok[nP:]])))
