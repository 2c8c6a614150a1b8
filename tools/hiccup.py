"""Where do the runtime's multi-millisecond stalls fall?  Per-call wall time of `steps` headline steps (and, with `frames`, of frame-leg style map calls);
prints every call above 1 ms with its index.  Finding (round 4): the 40-ms calls are the Python garbage collector's generation-2 passes (HICCUP_NOGC=1: none left), not the library or the runtime."""
import gc, importlib, os, sys, time
if os.environ.get("HICCUP_NOGC"): gc.disable()
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N, H = 500, 200
seq = synth.make_sequence(N, 200, H, motion_noise=synth.HEADLINE["motion_noise"])
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, max_landmarks=N + 2, std_z=1.0)
f.set_x_p_k_k(seq["x0"], seq["P0"]); f.defer_hi_update(True)
t_all = []
for i in range(steps):
    s = seq["steps"][i % 200]
    if i % 200 == 0: f.set_x_p_k_k(seq["x0"], seq["P0"])
    t0 = time.perf_counter()
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    if mode in ("map", "maponly"):
        ta = time.perf_counter(); f.delete_features([f.N - 1]); tb = time.perf_counter(); f.add_features_inverse_depth(np.array([[60.0, 60.0]]), 1.0, 0.5); tc = time.perf_counter()
        if tc - ta > 1e-3: print("   call %d: delete %.0f us, add %.0f us" % (i, 1e6 * (tb - ta), 1e6 * (tc - tb)), flush=True)
    if mode == "maponly":
        t_all.append(time.perf_counter() - t0)
        if t_all[-1] > 1e-3: print("   call %d: %.0f us" % (i, 1e6 * t_all[-1]), flush=True)
        continue
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
    t_all.append(time.perf_counter() - t0)
    if t_all[-1] > 1e-3 or i in (158, 160, 235, 237): print("   call %d: %.0f us, m=%d, stats %s" % (i, 1e6 * t_all[-1], len(s["meas_idx"]), st), flush=True)
t = np.array(t_all) * 1e6
print("median %.0f us, p99 %.0f us, max %.0f us" % (np.median(t), np.percentile(t, 99), t.max()))
for i in np.nonzero(t > 1000)[0]: print("  call %d: %.0f us" % (i, t[i]))
f.close()
