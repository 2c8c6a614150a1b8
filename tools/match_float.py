"""Float / double class matcher at 4096 x 4096 x 128, inputs resident in HBM: the ranked MFMA path (bf16 distance GEMM + exact re-evaluation,
or the int8 route for integer-valued descriptors) timed with HIP events (pre3_match_bench_*), and the exact VALU kernels through the
stateless entry for comparison (that figure includes the host copies).  Usage: python tools/match_float.py"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
lib = pre3._lib.lib
rng = np.random.default_rng(4)
K = 4096
real1 = np.abs(rng.standard_normal((128, K))) * 40
real2 = real1[:, rng.permutation(K)] + rng.uniform(-2, 2, (128, K))
cases = [("double, real-valued", real1, real2), ("float, real-valued", real1.astype(np.float32), real2.astype(np.float32)),
         ("double, integer-valued 0..255 (vl_sift)", np.minimum(np.round(real1), 255), np.minimum(np.round(np.abs(real2)), 255))]
for name, L1, L2 in cases:
    cls = 0 if L1.dtype == np.float64 else 1
    a, b = np.asfortranarray(L1), np.asfortranarray(L2)
    h = lib.pre3_match_bench_create_cls(0, cls, 128, K, a.ctypes.data_as(C.c_void_p), K, b.ctypes.data_as(C.c_void_p))
    ms = C.c_double(0)
    for _ in range(3):
        lib.pre3_match_bench_run(C.c_void_p(h), 30, C.byref(ms))
    info = (C.c_int32 * 3)()
    lib.pre3_match_bench_info(C.c_void_p(h), info)
    got = [np.zeros(K), np.zeros(K), np.zeros(K, np.int32)]
    lib.pre3_match_bench_fetch(C.c_void_p(h), *[g.ctypes.data_as(C.c_void_p) for g in got])
    lib.pre3_match_bench_destroy(C.c_void_p(h))
    os.environ["PRE3_MATCH_FLOAT_FORM"] = "0"
    pre3.siftmatch_partial(L1, L2, 0)
    t0 = time.perf_counter(); ref = pre3.siftmatch_partial(L1, L2, 0); t1 = time.perf_counter()
    del os.environ["PRE3_MATCH_FLOAT_FORM"]
    same = all(np.array_equal(g, r) for g, r in zip(got, ref))
    print("%-42s route %d: %7.2f us per match (resident); %d queries scanned in full, %.2f candidates per query; exact kernels incl. copies %.0f us; %s"
          % (name, info[0], ms.value * 1e3, info[1], info[2] / K, (t1 - t0) * 1e6, "identical" if same else "DIFFERENT"), flush=True)
