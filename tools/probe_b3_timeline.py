"""Chip-wide timeline and placement of k_downdate_b3 from a PROBE build (make -C 3pre_amd/csrc probe):
PRE3_LIB=3pre_amd/lib/libpre3_probe.so python tools/probe_b3_timeline.py [r]"""
import collections, ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
lib = pre3._lib.lib
N, r = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 640
n = 13 + 6 * N
f = pre3.EkfFilter([250.0, 90, 70, 0, 0, 144, 176], np.zeros(N, np.int32), dtype="f32", max_hyp=4)
f.set_x_p_k_k(np.zeros(n), np.eye(n))
f.bench_downdate(r, 3)
ms = f.bench_downdate(r, 1)
buf = (C.c_ulonglong * (2048 * 4))(); hw = (C.c_uint * 2048)()
lib.pre3_debug_k9rt(buf); lib.pre3_debug_k9hw(hw)
a = np.array(buf[:], dtype=np.uint64).reshape(2048, 4).astype(np.int64)
h = np.array(hw[:], dtype=np.uint32)
ok = a[:, 3] > 0
a, h = a[ok], h[ok]
t0 = a[:, 0].min()
a = (a - t0) / 100.0
big = (h >> 24) & 1
cu = ((h >> 16) & 0xf) * 1000 + ((h >> 13) & 0x7) * 100 + ((h >> 8) & 0xf)          # xcc, se_id, cu_id
print("r=%d: event time (2 launches incl. split) %.1f us; %d workgroups (%d big, %d small); span %.1f us" % (r, ms * 1e3, len(a), big.sum(), (1 - big).sum(), a[:, 3].max()))
for name, m in (("big", big == 1), ("small", big == 0)):
    s = a[m]
    if len(s) == 0: continue
    print("  %-5s start %.1f..%.1f | prologue %.1f (max %.1f) | loop %.1f (min %.1f max %.1f) | epilogue %.1f (max %.1f) | done %.1f..%.1f" % (
        name, s[:, 0].min(), s[:, 0].max(), (s[:, 1] - s[:, 0]).mean(), (s[:, 1] - s[:, 0]).max(), (s[:, 2] - s[:, 1]).mean(), (s[:, 2] - s[:, 1]).min(),
        (s[:, 2] - s[:, 1]).max(), (s[:, 3] - s[:, 2]).mean(), (s[:, 3] - s[:, 2]).max(), s[:, 3].min(), s[:, 3].max()))
load = collections.Counter()
for c, b in zip(cu, big): load[c] += 4 if b else 1
hist = collections.Counter(load.values())
print("  distinct CUs used: %d; load (quarter-tile units) histogram: %s" % (len(load), dict(sorted(hist.items()))))
pb = (C.c_ulonglong * 16)(); lib.pre3_debug_probe(pb)
if pb[3] > pb[2]:
    print("  workgroup 7: loop %d shader cycles in %.2f us -> %.2f GHz" % (pb[1] - pb[0], (pb[3] - pb[2]) / 100.0, (pb[1] - pb[0]) / ((pb[3] - pb[2]) * 10.0)))
f.close()
