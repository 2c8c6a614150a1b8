"""GPU: the persistent factorisation + solve (pre3_cholp.hip) against the launch-per-panel form on the same steps.
usage: python tools/cholp_check.py [N] [steps]   -- prints, per step, max |dP| / scale, max |dx| between the two forms (same inputs)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_hyp = 200
seq = synth.make_sequence(N, steps, n_hyp)
fs = []
for on in (False, True):
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
    f.chol_persist(on)
    print("persist requested", on, "in effect", f.chol_persist(), flush=True)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    fs.append(f)
ok = True
for si, s in enumerate(seq["steps"]):
    out = []
    for f in fs:
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        li, hi = f.get_flags()
        out.append((st, li, hi, f.get_x_k_k(), f.get_p_k_k()))
    (s0, li0, hi0, x0, P0), (s1, li1, hi1, x1, P1) = out
    scale = np.abs(P0).max()
    dP, dx = np.abs(P1 - P0).max() / scale, np.abs(x1 - x0).max()
    same = np.array_equal(li0, li1) and np.array_equal(hi0, hi1)
    print("step %d n_li %d/%d n_hi %d/%d  dP/scale %.3e dx %.3e flags equal %s finite %s" % (si, s0["n_li"], s1["n_li"], s0["n_hi"], s1["n_hi"], dP, dx, same, np.isfinite(P1).all()), flush=True)
    ok &= same and dP < 1e-4 and dx < 1e-5
    # keep the two filters on the same trajectory: each step compares ONE update
    fs[1].set_x_p_k_k(x0, P0)
print("OK" if ok else "MISMATCH")
# timing, whole steps (synchronous HI update), both forms
for f, name in zip(fs, ("per-panel", "persistent")):
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    reps = 0
    t0 = None
    for rep in range(40):
        for s in seq["steps"]:
            if rep == 5 and t0 is None: f.sync(); t0 = time.perf_counter()
            f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
            if t0 is not None: reps += 1
    f.sync()
    print("%s: %.1f us/step" % (name, (time.perf_counter() - t0) / reps * 1e6), flush=True)
sys.exit(0 if ok else 1)
