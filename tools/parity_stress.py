"""Randomised parity stress (not part of the test suite): many seeds / sizes, two steps each, fp64 and fp32, against the oracle.
fp64: inlier sets and RANSAC statistics exact, state to 1e-9 of its scale; fp32: inlier sets compared, mismatches counted (a
borderline residual may legitimately flip at fp32 rounding)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
oracle.build()
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
bad64 = bad32 = tot = 0
for seed in range(40):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(8, 90)); nh = int(rng.integers(5, 40))
    seq = synth.make_sequence(N, 2, nh, seed=500 + seed, sigma_z=float(rng.choice([0.25, 0.5, 1.0])))
    types, off, n = oracle.landmark_table(np.zeros(N, int))
    for dtype in ("f64", "f32"):
        f = pre3.EkfFilter(seq["cam"], types, dtype=dtype, max_hyp=nh, std_z=1.0)
        x, P = seq["x0"], seq["P0"]
        f.set_x_p_k_k(x, P)
        for s in seq["steps"]:
            ee = bool(seed % 2)
            st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=ee)
            ref = oracle.step(types, off, seq["cam"], x, P, s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0, early_exit=ee)
            li, hi = f.get_flags()
            same = np.array_equal(li, ref["li"]) and np.array_equal(hi, ref["hi"])
            tot += 1
            if dtype == "f64":
                ok = same and np.abs(f.get_p_k_k() - ref["P_kk"]).max() < 1e-9 * np.abs(P).max() and np.abs(f.get_x_k_k() - ref["x_kk"]).max() < 1e-9
                if "ransac" in ref:
                    r = ref["ransac"]
                    ok = ok and (st["best"], st["iters"], st["n_hyp"], st["max_support"]) == (r["best"], r["iters"], r["n_hyp"], r["max_support"])
                if not ok:
                    bad64 += 1; print("f64 MISMATCH seed", seed, "N", N, flush=True)
                x, P = ref["x_kk"], ref["P_kk"]
            else:
                if not same:
                    bad32 += 1
                if dtype == "f32":
                    x, P = ref["x_kk"], ref["P_kk"]
                    f.set_x_p_k_k(x, P)          # keep the fp32 run on the oracle's trajectory
        f.close()
print("steps checked %d: fp64 mismatches %d, fp32 inlier-set differences %d" % (tot, bad64, bad32))
assert bad64 == 0
