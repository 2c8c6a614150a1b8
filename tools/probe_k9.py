"""Per-tile phase timing of K9 from a PROBE=1 build (make -B -C 3pre_amd/csrc PROBE=1): cycles between
ticket / first stage staged / k-loop done / epilogue done, for the first 64 workgroups."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
lib = pre3._lib.lib
N, r = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 640
n = 13 + 6 * N
f = pre3.EkfFilter([250.0, 90, 70, 0, 0, 144, 176], np.zeros(N, np.int32), dtype="f32", max_hyp=4)
f.set_x_p_k_k(np.zeros(n), np.eye(n))
f.bench_downdate(r, 3)
lib.pre3_debug_k9_clear()
ms = f.bench_downdate(r, 1)          # warm launch + 1 timed launch: stamps hold the last launch
buf = (C.c_ulonglong * (64 * 8 * 4))()
lib.pre3_debug_k9_stamps(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 8, 4).astype(np.int64)
t0 = a[a > 0].min()
print("launch %.1f us; stamps relative to the earliest (cycles @2.4GHz -> us)" % (ms * 1e3))
ok = a[:, :, 3] > 0
pro = (a[:, :, 1] - a[:, :, 0])[ok]; loop = (a[:, :, 2] - a[:, :, 1])[ok]; epi = (a[:, :, 3] - a[:, :, 2])[ok]
print("tiles recorded %d (per WG %.2f)" % (ok.sum(), ok.sum() / 64.0))
for name, v in (("prologue (ticket seen -> stage 0 in LDS)", pro), ("k-loop", loop), ("epilogue", epi)):
    print("  %-42s mean %7.0f  min %7.0f  max %7.0f cycles  (%.2f us mean)" % (name, v.mean(), v.min(), v.max(), v.mean() / 2400))
gap = []
for w in range(64):
    for t in range(1, 8):
        if ok[w, t]:
            gap.append(a[w, t, 0] - a[w, t - 1, 3])
if gap:
    gap = np.array(gap); print("  %-42s mean %7.0f  max %7.0f cycles" % ("gap epilogue end -> next ticket seen", gap.mean(), gap.max()))
first = (a[:, 0, 0] - t0); last = (a[:, :, 3].max(axis=1) - t0)
print("  first tile starts at %.1f..%.1f us, WGs finish at %.1f..%.1f us" % (first.min() / 2400, first.max() / 2400, last.min() / 2400, last.max() / 2400))
g = (C.c_ulonglong * 16)()
lib.pre3_debug_probe(g)
cyc, rt = g[10] - g[8], g[11] - g[9]
print("  WG0 lifetime: %d shader cycles in %.2f us (s_memrealtime, 100 MHz) -> %.0f MHz under this load" % (cyc, rt / 100.0, cyc / (rt / 100.0)))
f.close()
