"""Soak of the round-4 launch structure on the headline workload (N=500, threshold 1.0, motion noise 2.5: persistent factorisation with the down-date
consumers and the x-update inside, k_hi_fused, speculative down-date): `reps` repetitions of the same `steps`-step sequence must end in bit-identical
states (every hand-off of the launch is flag-driven: a rare race would show as a difference), and the state must stay finite and symmetric."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
steps, reps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100)
N, H = 500, 200
seq = synth.make_sequence(N, steps, H, motion_noise=synth.HEADLINE["motion_noise"])
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
f.defer_hi_update(True)
t = time.perf_counter(); ref = None; n_hi = 0
for r in range(reps):
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    for s in seq["steps"]:
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=synth.HEADLINE["threshold"], early_exit=False)
        n_hi += st["n_hi"]
    x, P = f.get_x_k_k(), f.get_p_k_k()
    if ref is None: ref = (x, P)
    assert np.array_equal(x, ref[0]) and np.array_equal(P, ref[1]), "repetition %d differs" % r
    assert np.isfinite(P).all()
    if r % 10 == 9: print("  %d repetitions, %.0f s" % (r + 1, time.perf_counter() - t), flush=True)
A = np.abs(P - P.T); A[3:7, :] = 0; A[:, 3:7] = 0
print("headline workload: %d steps x %d repetitions bit-identical (mean HI landmarks per step %.1f), P symmetric outside the Jnorm rows: %s, %.1f s"
      % (steps, reps, n_hi / (steps * reps), A.max() == 0.0, time.perf_counter() - t), flush=True)
f.close()
