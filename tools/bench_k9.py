"""K9 roofline probe: P <- P - W'W with a synthetic W of r rows (no filter around it)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
for dtype, N, rs in (("f32", 500, (64, 128, 320, 640, 800)), ("f64", 200, (64, 320)), ("f32", 2000, (640, 3200))):
    n = 13 + 6 * N
    f = pre3.EkfFilter([250.0, 90, 70, 0, 0, 144, 176], np.zeros(N, np.int32), dtype=dtype, max_hyp=4)
    f.set_x_p_k_k(np.zeros(n), np.eye(n))
    for r in rs:
        ms = f.bench_downdate(r, 20)
        syrk = n * (n + 1.0) * r
        print("%s N=%d n=%d r=%d: %.1f us  SYRK %.1f TF  (2n^2r-equivalent %.1f TF)  P traffic %.0f GB/s" % (
            dtype, N, n, r, ms * 1e3, syrk / ms / 1e9, 2.0 * n * n * r / ms / 1e9, 1.5 * n * n * (4 if dtype == "f32" else 8) / ms / 1e6))
    f.close()
