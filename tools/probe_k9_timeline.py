"""Chip-wide timeline of the one-tile K9 kernel from a PROBE=1 build (s_memrealtime stamps per workgroup)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PRE3_K9_FORM"] = "1"
pre3 = importlib.import_module("3pre_amd")
lib = pre3._lib.lib
N, r = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 640
n = 13 + 6 * N
f = pre3.EkfFilter([250.0, 90, 70, 0, 0, 144, 176], np.zeros(N, np.int32), dtype="f32", max_hyp=4)
f.set_x_p_k_k(np.zeros(n), np.eye(n))
f.bench_downdate(r, 3)
ms = f.bench_downdate(r, 1)
buf = (C.c_ulonglong * (2048 * 4))()
lib.pre3_debug_k9rt(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(2048, 4).astype(np.int64)
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
a = (a - t0) / 100.0       # us
print("r=%d: event time %.1f us; %d workgroups; kernel span %.1f us" % (r, ms * 1e3, len(a), a[:, 3].max()))
for g in range(5):
    sl = a[g * 256:(g + 1) * 256]
    if len(sl) == 0: break
    print("  generation %d (%3d WGs): start %.1f..%.1f  loop begins %.1f..%.1f  loop ends %.1f..%.1f  done %.1f..%.1f" % (
        g, len(sl), sl[:, 0].min(), sl[:, 0].max(), sl[:, 1].min(), sl[:, 1].max(), sl[:, 2].min(), sl[:, 2].max(), sl[:, 3].min(), sl[:, 3].max()))
print("  mean phases: prologue %.1f  loop %.1f  epilogue %.1f us" % ((a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(), (a[:, 3] - a[:, 2]).mean()))
f.close()
