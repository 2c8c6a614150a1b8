"""fp32 covariance path against the fp64 path after whole filter steps at config-3 size (N=500): how far the K9 form in use
(PRE3_K9_B3=1: bf16 x 3 split on the bf16 MFMA, =0: f32 MFMA) moves P and x away from the fp64 result, and whether P stays
exactly symmetric.  usage: PRE3_K9_B3=0|1 python tools/k9_accuracy.py [N] [steps]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seq = synth.make_sequence(N, steps, 200)
out = {}
for dt in ("f64", "f32"):
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dt, max_hyp=200, std_z=1.0)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    for s in seq["steps"]:
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0, early_exit=False)
    x, P = f.get_x_k_k(), f.get_p_k_k()
    out[dt] = (x.copy(), P.copy(), st)
    f.close()
x64, P64, s64 = out["f64"]; x32, P32, s32 = out["f32"]
d = np.sqrt(np.diag(P64))
print("form:", "bf16x3" if os.environ.get("PRE3_K9_B3", "1") != "0" else "f32 mfma", " N=%d steps=%d" % (N, steps), " LI/HI f64", s64["n_li"], s64["n_hi"], "f32", s32["n_li"], s32["n_hi"])
print("  max |P32-P64| / max|P64|          = %.3e" % (np.abs(P32 - P64).max() / np.abs(P64).max()))
print("  max |P32-P64|_ij / (s_i s_j)       = %.3e   (correlation units)" % (np.abs(P32 - P64) / np.outer(d, d)).max())
print("  ||P32-P64||_F / ||P64||_F          = %.3e" % (np.linalg.norm(P32 - P64) / np.linalg.norm(P64)))
print("  max |x32-x64| / sigma              = %.3e" % (np.abs(x32 - x64) / d).max())
print("  P32 exactly symmetric: %s, min eig(P32) / max eig = %.3e" % (np.array_equal(P32, P32.T), np.linalg.eigvalsh(P32)[0] / np.linalg.eigvalsh(P32)[-1]))
