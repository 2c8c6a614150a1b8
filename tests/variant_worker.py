"""Worker of tests/test_gpu_variants.py: a short filter sequence in the launch structure the environment selects; prints a digest of the
final state and of every step's statistics."""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    pre3 = importlib.import_module("3pre_amd")
    synth = importlib.import_module("3pre_amd.synth")
    h = hashlib.sha256()
    hf = hashlib.sha256()                                 # statistics and flags only
    dump = {}
    for N, n_hyp, dtype, steps in ((120, 60, "f32", 12), (120, 60, "f64", 6), (9, 8, "f32", 6)):
        seq = synth.make_sequence(N, steps, n_hyp, seed=4242 + N)
        f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_hyp, max_landmarks=N + 4, std_z=1.0)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.defer_hi_update(True)
        for t, s in enumerate(seq["steps"]):
            z = s["z"]
            if t == 3:                                    # a frame whose measurements are all gross outliers: RANSAC runs, (almost) nothing is selected
                z = z + 300.0
            st = f.step(s["u"], s["meas_idx"], z, s["hyp"], threshold=1.0, early_exit=bool(t % 2))
            h.update(repr(sorted(st.items())).encode())
            hf.update(repr(sorted(st.items())).encode())
        li, hi = f.get_flags()
        h.update(li.tobytes()); h.update(hi.tobytes())
        hf.update(li.tobytes()); hf.update(hi.tobytes())
        h.update(f.get_x_k_k().tobytes()); h.update(f.get_p_k_k().tobytes())
        dump["x_%d_%s" % (N, dtype)] = f.get_x_k_k(); dump["P_%d_%s" % (N, dtype)] = f.get_p_k_k()
        # map management behind the steps (map_management.m:27-79): delete, add, convert -- the congruence in either of its forms
        f.sync()
        rng = np.random.default_rng(7 + N)
        f.delete_features([1, N // 2])
        uvd = np.stack([rng.uniform(5, 170, 3), rng.uniform(5, 140, 3)], 1)
        f.add_features_inverse_depth(uvd, 1.0, rng.uniform(0.1, 1.0, 3))
        conv = f.inversedepth_2_cartesian(1e6)              # (a threshold every inverse-depth landmark passes)
        h.update(np.asarray(conv).tobytes())
        h.update(f.get_x_k_k().tobytes()); h.update(f.get_p_k_k().tobytes())
        f.close()
    print("DIGEST", h.hexdigest())
    print("FLAGS", hf.hexdigest())
    if os.environ.get("VARIANT_DUMP"):
        np.savez(os.environ["VARIANT_DUMP"], **dump)


if __name__ == "__main__":
    main()
