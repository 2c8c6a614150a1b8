"""int8 matcher: the query-per-lane kernel (default) against the row-scan form (PRE3_MATCH_I8_FORM=0) -- identical results and kernel time,
4096 x 4096 x 128 uint8 resident in HBM (pre3_match_bench_*), plus ragged shapes through the stateless entry against the oracle."""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    pre3 = importlib.import_module("3pre_amd")
    lib = pre3._lib.lib
    rng = np.random.default_rng(5000)
    K = 4096
    L1 = np.minimum(np.round(np.abs(rng.standard_normal((K, 128))) * 40), 255).astype(np.uint8)
    L2 = np.clip(L1[rng.permutation(K)].astype(int) + rng.integers(-2, 3, (K, 128)), 0, 255).astype(np.uint8)
    h = lib.pre3_match_bench_create(0, 128, K, L1.ctypes.data_as(C.c_void_p), K, L2.ctypes.data_as(C.c_void_p))
    ms = C.c_double(0)
    for _ in range(3):
        lib.pre3_match_bench_run(C.c_void_p(h), 50, C.byref(ms))
    b, s, a = np.zeros(K), np.zeros(K), np.zeros(K, np.int32)
    lib.pre3_match_bench_fetch(C.c_void_p(h), b.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
    lib.pre3_match_bench_destroy(C.c_void_p(h))
    print("%.2f us per match, %.0f int8 TOPS; checksum %d %d %d" % (ms.value * 1e3, 2.0 * K * K * 128 / (ms.value * 1e-3) / 1e12, int(b.sum()), int(s.sum()), int(a.sum())))
    import oracle as orc
    bad = 0
    for (k1, k2) in ((1, 1), (5, 1), (33, 2), (70, 90), (129, 257), (300, 31), (1000, 1500)):
        A = rng.integers(0, 256, (128, k1)).astype(np.uint8); B = rng.integers(0, 256, (128, k2)).astype(np.uint8)
        if k2 > 3: B[:, 3] = A[:, 0]
        m, d = pre3.siftmatch(A, B, 1.5, return_scores=True)
        mr, dr = orc.siftmatch(A, B, 1.5)
        ok = np.array_equal(m, mr) and np.array_equal(d, dr)
        bad += not ok
        print("   %4d x %4d: %s (%d matches)" % (k1, k2, "identical to the oracle" if ok else "DIFFERENT", m.shape[1]))
    sys.exit(1 if bad else 0)
for form in ("1", "0"):
    print("PRE3_MATCH_I8_FORM=%s:" % form, flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, PRE3_MATCH_I8_FORM=form))
