"""How far apart are the two forms of update.m:32-38 in fp32 -- the persistent launch (explicit M_J = L_JJ^-1 per panel, products on the bf16 matrix cores)
and the launch-per-panel form (the chain solves W directly) -- and how far is each from update.m in fp64 (numpy twin), as the number of 64-row panels
grows?  (round-4 verdict: the 16-panel case of tests/test_gpu_cholp.py differs by 1.2e-4 of P's scale between the forms, 13 panels stay below 1e-4.)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
import oracle as orc
from oracle import np_twin as tw
N = 500
seq = synth.make_sequence(N, 1, 8); s = seq["steps"][0]
types, off, n = orc.landmark_table(np.zeros(N, int))
xr, Pr = tw.predict(seq["x0"], seq["P0"], s["u"])
hh, has = tw.project(types, off, xr, seq["cam"])
Hc, Hl = tw.jacobian(types, off, xr, seq["cam"], hh, has)
print("landmarks rows panels | persistent vs per-panel | persistent vs fp64 | per-panel vs fp64   (max |dP| / max |P|)")
for m in (32, 64, 128, 192, 256, 320, 384, 416, 448, 480, 500):
    meas = np.arange(m, dtype=np.int32)
    res = []
    for on in (True, False):
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=8, std_z=1.0)
        f.chol_persist(on)
        f.set_x_p_k_k(seq["x0"], seq["P0"]); f.ekf_prediction(s["u"]); f.search_IC_matches()
        z = f.landmark_fields()["h"][:m] + 0.25
        f.set_measurements(meas, z); f.set_flags(li=np.ones(m, np.int32)); f.ekf_update_li_inliers()
        res.append(f.get_p_k_k()); f.close()
    zfull = np.zeros((N, 2)); zfull[meas] = z
    xo, Po = tw.update_landmarks(types, off, meas, xr, Pr, Hc, Hl, zfull, hh)
    sc = np.abs(Po).max()
    print("%4d %5d %3d | %.2e | %.2e | %.2e" % (m, 2 * m, (2 * m + 63) // 64, np.abs(res[0] - res[1]).max() / sc, np.abs(res[0] - Po).max() / sc, np.abs(res[1] - Po).max() / sc), flush=True)
