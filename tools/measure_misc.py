"""Numbers quoted in DESIGN.md that bench.py's headline line does not carry: PCIe-inclusive stateless update,
config 2 (N=200, fp64 predict+update), K9 at the fp64 peak."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
import oracle as orc

# (a) stateless update(x,P,H,R,z,h): host arrays in, host arrays out (what the update.mex gateway does)
for N, dtype in ((500, "f32"), (500, "f64"), (200, "f64")):
    seq = synth.make_sequence(N, 1, 4)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    h, has = orc.project(types, off, seq["x0"], seq["cam"])
    Hc, Hl = orc.jacobian(types, off, seq["x0"], seq["cam"], h, has)
    sel = s["meas_idx"][:int(0.8 * len(s["meas_idx"]))]
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    for k, i in enumerate(sel):
        for c in range(2):
            for j in range(7):
                rows.append(2 * k + c); cols.append(j); vals.append(Hc[i, c, j])
            for j in range(6):
                rows.append(2 * k + c); cols.append(off[i] + j); vals.append(Hl[i, c, j])
    H = sp.csr_matrix((vals, (rows, cols)), shape=(2 * len(sel), n))
    z = (h[sel] + 0.3).ravel(); hh = h[sel].ravel()
    pre3.update(seq["x0"], seq["P0"], H, None, z, hh, dtype=dtype, want_K=False)
    t0 = time.perf_counter()
    for _ in range(3):
        pre3.update(seq["x0"], seq["P0"], H, None, z, hh, dtype=dtype, want_K=False)
    el = (time.perf_counter() - t0) / 3
    print("stateless update N=%d n=%d r=%d %s: %.1f ms per call (PCIe + marshalling inclusive; P in+out = %.0f MB)" % (N, n, H.shape[0], dtype, el * 1e3, 2 * n * n * 8 / 1e6))

# (b) config 2: N=200 (n=1213), fp64, predict + one update with r=320 rows, state resident
N = 200
seq = synth.make_sequence(N, 30, 4)
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f64", max_hyp=4)
f.set_x_p_k_k(seq["x0"], seq["P0"])
def one(s):
    f.ekf_prediction(s["u"])
    f.search_IC_matches()
    f.set_measurements(s["meas_idx"], s["z"])
    f.ekf_update_all()
for s in seq["steps"][:5]: one(s)
f.sync(); f.kernel_timing(True)
t0 = time.perf_counter()
for s in seq["steps"][5:]: one(s)
f.sync(); el = time.perf_counter() - t0
kt = f.kernel_timing_read()
print("config 2 (N=200, n=1213, fp64, predict + project + update_all r=%d): %.1f us/step, %.0f steps/s; K9 %.1f us/launch, %.1f TF SYRK-count (%.1f%% of 78.6 TF fp64 MFMA peak)" % (
    2 * len(seq["steps"][5]["meas_idx"]), el / 25 * 1e6, 25 / el, kt["total_ms"] / kt["launches"] * 1e3, kt["flops"] / kt["total_ms"] / 1e9, kt["flops"] / kt["total_ms"] / 1e9 / 78.6 * 100))
f.close()
