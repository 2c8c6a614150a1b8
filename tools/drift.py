"""fp32 covariance path over a long horizon: the same N=500 sequence through an f32 and an f64 filter (both on the GPU).
Prints, per checkpoint: whether the LI / HI sets still agree, ||P32 - P64||_F / ||P64||_F, max |x32 - x64| and min eig(P32)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
N, steps, n_hyp = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 300, 200
seq = synth.make_sequence(N, steps, n_hyp)
f = {d: pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=d, max_hyp=n_hyp, std_z=1.0) for d in ("f32", "f64")}
for d in f: f[d].set_x_p_k_k(seq["x0"], seq["P0"])
first_diff, n_diff = None, 0
for t, s in enumerate(seq["steps"]):
    fl = {}
    for d in f:
        f[d].step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        fl[d] = f[d].get_flags()
    same = np.array_equal(fl["f32"][0], fl["f64"][0]) and np.array_equal(fl["f32"][1], fl["f64"][1])
    if not same:
        n_diff += 1
        if first_diff is None: first_diff = t
    if (t + 1) in (1, 3, 10, 30, 100, 200, 300, steps):
        P32, P64 = f["f32"].get_p_k_k(), f["f64"].get_p_k_k()
        x32, x64 = f["f32"].get_x_k_k(), f["f64"].get_x_k_k()
        sig = np.sqrt(np.diag(P64))
        ev = np.linalg.eigvalsh(0.5 * (P32 + P32.T)).min()
        print("step %4d: sets equal so far %s (steps with a difference: %d, first at %s)  ||dP||_F/||P||_F %.2e  max|dx|/sigma %.2e  min eig(P32) %.3e  min eig(P64) %.3e  trace ratio %.6f"
              % (t + 1, n_diff == 0, n_diff, first_diff, np.linalg.norm(P32 - P64) / np.linalg.norm(P64), np.abs((x32 - x64) / sig).max(), ev, np.linalg.eigvalsh(0.5 * (P64 + P64.T)).min(), np.trace(P32) / np.trace(P64)), flush=True)
