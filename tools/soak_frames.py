"""Soak of the frame path's round-5 launches (map management, asynchronous scan upload, the fused two-launch IC search with the matcher riding in the projection's
launch, step_predicted with k_hi_fused up to 64 landmarks): `reps` repetitions of the same `frames`-frame sequence must end in bit-identical states, match lists
and descriptor banks."""
import importlib, os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
frames, reps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30, 40)
N, H, K2 = 500, 200, 600
seq = synth.make_sequence(N, frames, H, motion_noise=synth.HEADLINE["motion_noise"])
rng0 = np.random.default_rng(5)
bank0 = np.abs(rng0.normal(0, 1, (128, N))); bank0 /= np.linalg.norm(bank0, axis=0)
def make_scan(k, h, has_h, bank):
    """the frame's SIFT set: noisy copies of 400 predicted landmarks' descriptors at noisy pixels, clutter for the rest (deterministic in k and h)"""
    rng = np.random.default_rng(9000 + k)
    sd = np.abs(rng.normal(0, 1, (128, K2))); sd /= np.linalg.norm(sd, axis=0)
    sp = np.stack([rng.uniform(1, 176, K2), rng.uniform(1, 144, K2), np.full(K2, 2.0), np.zeros(K2)])
    seen = np.nonzero(has_h)[0]
    seen = seen[rng.permutation(len(seen))[:400]]
    cols = rng.permutation(K2)[:len(seen)]
    sd[:, cols] = bank[:, seen] + rng.normal(0, 0.01, (128, len(seen)))
    sp[0:2, cols] = h[seen].T + rng.normal(0, 1.0, (2, len(seen)))
    return sd, sp


hyp = [synth.draw_hypotheses(np.random.default_rng(100 + k), 400, H) for k in range(frames)]
t = time.perf_counter(); ref = None; routes = set()
for r in range(reps):
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, max_landmarks=N + 2, std_z=1.0)
    f.set_x_p_k_k(seq["x0"], seq["P0"]); f.set_descriptors(bank0); f.defer_hi_update(True)
    h = hashlib.sha256()
    for k in range(frames):
        s = seq["steps"][k]
        f.map_management([f.N - 1], np.array([[60.0 + k, 70.0]]), 1.0, 0.5)
        f.set_descriptors(bank0[:, k:k + 1], first=f.N - 1)
        f.ekf_prediction(s["u"])
        f.predict_camera_measurements()
        lf = f.landmark_fields()
        f.load_scan(*make_scan(k, lf["h"], lf["has_h"], f.get_descriptors()))
        ic = f.matching_sift_based(1.5, strict_reference=True)
        routes.add(f.ic_search_route())
        h.update(ic["match_idx"].tobytes()); h.update(ic["meas_idx"].tobytes())
        m = len(ic["meas_idx"])
        assert m >= 100, "frame %d: only %d accepted matches" % (k, m)
        hk = hyp[k] % m
        hk[:, 1] = np.where(hk[:, 1] == hk[:, 0], (hk[:, 1] + 1) % m, hk[:, 1])
        hk[:, 2] = np.where((hk[:, 2] == hk[:, 0]) | (hk[:, 2] == hk[:, 1]), (hk[:, 2] + 2) % m, hk[:, 2])
        hk[:, 2] = np.where((hk[:, 2] == hk[:, 0]) | (hk[:, 2] == hk[:, 1]), (hk[:, 2] + 1) % m, hk[:, 2])
        f.step_predicted(hk.astype(np.int32), threshold=1.0, early_exit=False)
    h.update(f.get_x_k_k().tobytes()); h.update(f.get_p_k_k().tobytes()); h.update(f.get_descriptors().tobytes())
    f.close()
    d = h.hexdigest()
    if ref is None: ref = d
    assert d == ref, "repetition %d differs" % r
    if r % 10 == 9: print("  %d repetitions, %.0f s" % (r + 1, time.perf_counter() - t), flush=True)
print("frame path: %d frames x %d repetitions bit-identical, IC routes seen %s, %.1f s" % (frames, reps, sorted(routes), time.perf_counter() - t), flush=True)
