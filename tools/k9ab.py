import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
N=500; n=13+6*N
f = pre3.EkfFilter([250.0, 90, 70, 0, 0, 144, 176], np.zeros(N, np.int32), dtype="f32", max_hyp=4)
f.set_x_p_k_k(np.zeros(n), np.eye(n))
for r in ([int(os.environ["K9_ROWS"])] if os.environ.get("K9_ROWS") else [320, 640]):
    ms = min(f.bench_downdate(r, 30) for _ in range(3))
    print(os.environ.get("PRE3_LIB","current").split("/")[-1], "r=%d: %.1f us %.1f TF" % (r, ms*1e3, n*(n+1.0)*r/ms/1e9))
