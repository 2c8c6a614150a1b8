"""matcher timing incl. H2D/D2H (the C ABI takes host arrays): 4096 x 4096 x 128 for each class"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
rng = np.random.default_rng(5000)
K = 4096
base = np.minimum(np.round(np.abs(rng.standard_normal((128, K))) * 40), 255)
L2 = np.clip(base[:, rng.permutation(K)] + rng.integers(-2, 3, (128, K)), 0, 255)
for dt in (np.uint8, np.float32, np.float64):
    a, b = base.astype(dt), L2.astype(dt)
    if dt != np.uint8:
        a = a / np.linalg.norm(a, axis=0, keepdims=True); b = b / np.linalg.norm(b, axis=0, keepdims=True)
        a, b = a.astype(dt), b.astype(dt)
    a, b = np.asfortranarray(a), np.asfortranarray(b)          # MATLAB's own layout (one descriptor per column, contiguous): no per-call transpose
    pre3.siftmatch(a, b)
    t0 = time.perf_counter()
    for _ in range(3): m = pre3.siftmatch(a, b)
    el = (time.perf_counter() - t0) / 3
    print("%s: %.2f ms per 4096x4096 match incl. transfers (%d matches)" % (dt.__name__, el * 1e3, m.shape[1]))
