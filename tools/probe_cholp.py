"""Timeline of ONE persistent factorisation + solve launch (pre3_cholp.hip) from its device-side wall-clock stamps (probe build).
usage (GPU box): PRE3_LIB=3pre_amd/lib/libpre3_probe.so python tools/probe_cholp.py [N] [warm steps]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("PRE3_LIB", "3pre_amd/lib/libpre3_probe.so")
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
lib = importlib.import_module("3pre_amd._lib").lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = 200 if N <= 500 else 1000
seq = synth.make_sequence(N, warm + 1, H)
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"])
for s in seq["steps"][:warm]:
    f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
s = seq["steps"][warm]
for rep in range(3):           # the same LI update three times (state restored): the last one is read
    xs, Ps = f.get_x_k_k(), f.get_p_k_k()
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    r = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)
    f.ekf_update_li_inliers(); f.sync()
    if rep < 2: f.set_x_p_k_k(xs, Ps)
buf = np.zeros(24 * 16 * 8, np.uint64)
fn = lib.pre3_debug_cholp; fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
raw = buf.reshape(24, 16, 8).astype(np.int64)
t = buf.reshape(24, 16, 8).astype(np.float64) / 100.0      # us
n_li = int(r["li_mask"].sum()); nrb = (2 * n_li + 63) // 64
t0 = t[0, 0, 0]
def rel(x): return "%7.2f" % (x - t0) if x > 0 else "      -"
print("n_li %d -> %d panels; all times in us after crit's first chain start" % (n_li, nrb))
print("crit main   J: chain start, chain end, b0 passed, b3 passed | side: T1 issued, T1 landed, T2 issued, T2 landed, M_J flag, rowL[J] flag")
for J in range(min(nrb, 16)):
    print("  J=%2d  %s %s %s %s | %s %s %s %s %s %s   chain %.2f  products %.2f" % ((J,) + tuple(rel(t[0, J, k]) for k in range(4)) + tuple(rel(t[1, J, k]) for k in range(6)) +
          (t[0, J, 1] - t[0, J, 0], (t[0, J, 3] - t[0, J, 1]) if t[0, J, 3] > 0 else 0.0)) + "  chain cycles %d (%.2f GHz)" % (raw[19, J, 1] - raw[19, J, 0], (raw[19, J, 1] - raw[19, J, 0]) / max(t[0, J, 1] - t[0, J, 0], 1e-9) / 1e3))
print("products, per J (us after the chain's end): side waves at b0 (w10, w11) | b0 | B1 done | w11 L_JJ stored, w10 M flag | b1 | b2 | B2 done | w10 planes out, w11 f32 out | b3")
for J in range(min(nrb, 16) - 1):
    e = t[0, J, 1]
    r2 = lambda x: "%5.2f" % (x - e) if x > 0 else "    -"
    print("  J=%2d  %s %s | %s | %s | %s %s | %s | %s | %s | %s %s | %s" % (J, r2(t[1, J, 7]), r2(t[18, J, 2]), r2(t[0, J, 2]), r2(t[0, J, 7]), r2(t[18, J, 0]), r2(t[1, J, 4]), r2(t[0, J, 4]), r2(t[0, J, 5]), r2(t[0, J, 6]), r2(t[1, J, 6]), r2(t[18, J, 1]), r2(t[0, J, 3])))
print("rows  (i, J): M_J seen, L(i,J) flagged, L(J+1,J) seen, T1 flagged, T2 flagged, panel done")
for i in range(2, min(nrb, 16)):
    for J in range(0, i - 1):
        print("  i=%2d J=%2d  %s" % (i, J, " ".join(rel(t[i, J, k]) for k in range(6))))
print("row 9, wave 5, panel 1 bulk: enter %s | wave 4: rows seen %s acquire done %s | wave 5 released %s" % (rel(t[12,1,0]), rel(t[12,1,1]), rel(t[12,1,2]), rel(t[12,1,3])))
for u in range(4):
    print("   item %d: start %s  after k-step 0..3 %s  stored %s" % (u, rel(t[12+u,0,0]), " ".join(rel(t[12+u,0,1+q]) for q in range(4)), rel(t[12+u,0,5])))
print("strip 0 / last strip, per J: (1) rhs planes done, (2) acquire done, (2) sum done, M_J seen, product reduced, stored + planes done, L(J+1,J) seen, its term done")
for J in range(min(nrb, 16)):
    order = (3, 0, 4, 1, 5, 2, 6, 7)
    print("  J=%2d  %s | %s" % (J, " ".join(rel(t[16, J, k]) for k in order), " ".join(rel(t[17, J, k]) for k in order)))
print("down-date consumers (first / last group), per J: flags seen, panel's MFMAs done")
for J in range(min(nrb, 16)):
    print("  J=%2d  %s %s | %s %s" % (J, rel(t[20, J, 0]), rel(t[20, J, 1]), rel(t[21, J, 0]), rel(t[21, J, 1])))
print("  epilogue start / end: %s %s | %s %s" % (rel(t[20, 15, 0]), rel(t[20, 15, 1]), rel(t[21, 15, 0]), rel(t[21, 15, 1])))

# round 6: per-role stamps of crit's last chain (flag-driven form)
if hasattr(lib, "pre3_debug_cha"):
    g = (C.c_ulonglong * (8 * 12 * 4))()
    lib.pre3_debug_cha(g)
    a = np.array(g[:], dtype=np.int64).reshape(8, 12, 4)
    t0 = a[0, 1, 0] if a[0, 1, 0] > 0 else a[a > 0].min()
    Jl = nrb - 1
    print("  last panel: chain start (all waves enter) -> factor wave's first stamp %d clocks -> its last sub-panel written %d -> z wave's last sub-panel written %d -> chain end (behind its last barrier) %d clocks" % (t0 - int(raw[19, Jl, 0]), int(a[0, 8, 3]) - t0, int(a[1, 8, 3]) - int(a[0, 8, 3]), int(raw[19, Jl, 1]) - int(a[1, 8, 3])))
    print("flag-driven chain, crit's last panel (shader clocks after the factor wave's first stamp): per step k = -1 .. 8 the four slots")
    for r, name in enumerate(["F", "z", "D0", "D2", "D3", "X0", "w10", "X2"]):
        row = []
        for k in range(10):
            row.append(" ".join("%6d" % (v - t0) if v > 0 and abs(v - t0) < 200000 else "     -" for v in a[r, k]))
        print("  %-3s | %s" % (name, " | ".join(row)))
