"""GPU: the IC-search stage on the device against the oracle's composition of
search_IC_matches.m:31-44 and matching_sift_based.m:104-149, plus the SIFT_result%04d.mat wire format."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("3pre_amd.synth")


def _scene(N, seed, n_clutter=60, far=0):
    """A predicted map with SIFT-like descriptors, and a scan holding noisy copies of most of them at noisy pixels
    (some far outside the gate, some missing, some clutter)."""
    rng = np.random.default_rng(seed)
    seq = synth.make_sequence(N, 1, 8, seed=seed)
    bank = np.abs(rng.normal(0, 1, (128, N)))
    bank /= np.linalg.norm(bank, axis=0)
    return rng, seq, bank


def _scan(rng, h, has_h, bank, n_clutter, frac_seen=0.8, px_sigma=1.5, n_far=4):
    N = bank.shape[1]
    seen = [i for i in range(N) if has_h[i] and rng.uniform() < frac_seen]
    desc, pos = [], []
    for j, i in enumerate(seen):
        d = bank[:, i] + rng.normal(0, 0.01, 128)
        p = h[i] + rng.normal(0, px_sigma, 2)
        if j < n_far:
            p = p + np.array([60.0, -45.0])                 # right descriptor, wrong place: the gate must reject it
        desc.append(d); pos.append([p[0], p[1], rng.uniform(1, 4), rng.uniform(-3, 3)])
    for _ in range(n_clutter):
        d = np.abs(rng.normal(0, 1, 128)); d /= np.linalg.norm(d)
        desc.append(d); pos.append([rng.uniform(1, 176), rng.uniform(1, 144), 2.0, 0.0])
    perm = rng.permutation(len(desc))
    return np.array(desc).T[:, perm].copy(), np.array(pos).T[:, perm].copy()


@pytest.mark.parametrize("rank_env", ["1", "0", "fused"])
def test_ic_search_on_the_matrix_cores_matches_oracle(pre3, orc, rank_env, monkeypatch):
    """N = 300 landmarks against ~300 keypoints (N * K2 >= 65536): matching_sift_based.m:118's siftmatch runs as the bf16 distance GEMM +
    exact re-evaluation inside pre3_ic_search (PRE3_IC_RANK=0: the exact VALU kernel); match list, scores-dependent ratio test, gate,
    measurement list, z and the refreshed bank are the oracle's bit for bit on both routes"""
    # ("fused": the default at these sizes since round 5 -- the exact matcher on 32 x 32 tiles + one gate workgroup, two launches; "1" / "0": the
    #  ranked and the exact 64 x 64 tiled routes it stands in front of, PRE3_IC_FUSED=0)
    monkeypatch.setenv("PRE3_IC_FUSED", "1" if rank_env == "fused" else "0")
    monkeypatch.setenv("PRE3_IC_RANK", "1" if rank_env == "fused" else rank_env)
    N = 300
    rng, seq, bank = _scene(N, 31)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.set_descriptors(bank)
    for frame in range(2):                                     # second frame: the bank holds refreshed (scan) descriptors
        f.ekf_prediction(s["u"])
        x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()
        h, has_h = orc.project(types, off, x1, seq["cam"])
        sd, sp = _scan(rng, h, has_h, bank, 90, px_sigma=4.0)
        sd[:, 5] = sd[:, 17]                                   # two identical keypoints: a tie for best / second best
        f.load_scan(sd, sp)
        bank_before = f.get_descriptors()
        out = f.matching_sift_based(1.5, strict_reference=True)
        assert f.ic_search_was_ranked() == (rank_env == "1") and N * sd.shape[1] >= 65536
        assert f.ic_search_route() == {"fused": 2, "1": 1, "0": 0}[rank_env]
        ref = orc.ic_search(types, off, x1, P1, seq["cam"], bank_before, sd, sp, 1.5, True)
        assert ref["match_idx"].shape[1] > 100
        assert np.array_equal(out["match_idx"], ref["match_idx"]) and np.array_equal(out["accepted"], ref["accepted"])
        assert np.array_equal(out["meas_idx"], ref["meas_idx"]) and np.array_equal(out["z"], ref["z"])
        assert np.array_equal(f.get_descriptors(), ref["bank"])
        f.set_x_p_k_k(x1, P1)
        bank = ref["bank"]
    f.close()


@pytest.mark.parametrize("bad", [1e-20, 2.0 ** 61])
def test_a_scan_outside_the_ranked_routes_bounds_takes_the_exact_kernel(pre3, orc, bad, monkeypatch):
    """pre3_set_scan checks the descriptors on their way into the staging block (k_rank_pack's own test: finite, |x| <= 2^60, no non-zero
    |x| < 2^-40; AVX2 on the host since round 5): one offending value and the search must leave the matrix-core route -- with the same results"""
    monkeypatch.setenv("PRE3_IC_FUSED", "0")
    N = 300
    rng, seq, bank = _scene(N, 37)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.set_descriptors(bank)
    f.ekf_prediction(s["u"])
    x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()
    h, has_h = orc.project(types, off, x1, seq["cam"])
    sd, sp = _scan(rng, h, has_h, bank, 90, px_sigma=4.0)
    f.load_scan(sd, sp)
    f.matching_sift_based(1.5, strict_reference=True)
    assert f.ic_search_route() == 1                              # in bounds: the ranked route
    f.set_x_p_k_k(seq["x0"], seq["P0"]); f.set_descriptors(bank); f.ekf_prediction(s["u"])
    sd2 = sd.copy(); sd2[77, 41] = bad                            # (an odd position: the check's vector loop and its tail both see ordinary values around it)
    f.load_scan(sd2, sp)
    out = f.matching_sift_based(1.5, strict_reference=True)
    assert f.ic_search_route() == 0
    ref = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd2, sp, 1.5, True)
    assert np.array_equal(out["match_idx"], ref["match_idx"]) and np.array_equal(out["meas_idx"], ref["meas_idx"])
    f.close()


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("dtype,strict", [("f64", True), ("f64", False), ("f32", True)])
def test_ic_search_matches_oracle(pre3, orc, dtype, strict, fused, monkeypatch):
    monkeypatch.setenv("PRE3_IC_FUSED", fused)
    N = 60
    rng, seq, bank = _scene(N, 23)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype=dtype, max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.set_descriptors(bank)
    f.ekf_prediction(s["u"])
    x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()              # the oracle runs from the device's own prediction
    h, has_h = orc.project(types, off, x1, seq["cam"])
    sd, sp = _scan(rng, h, has_h, bank, 60, px_sigma=5.0)      # wide enough that quirk Q5 changes the accepted set
    f.load_scan(sd, sp)
    out = f.matching_sift_based(1.5, strict_reference=strict)
    assert f.ic_search_route() == (2 if fused == "1" else 0)
    ref = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp, 1.5, strict)
    other = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp, 1.5, not strict)
    assert not np.array_equal(ref["accepted"], other["accepted"])
    assert ref["match_idx"].shape[1] > 20 and 0 < ref["accepted"].sum() < ref["match_idx"].shape[1]
    assert np.array_equal(out["match_idx"], ref["match_idx"]) and np.array_equal(out["accepted"], ref["accepted"])
    assert np.array_equal(out["meas_idx"], ref["meas_idx"]) and np.array_equal(out["z"], ref["z"])
    assert np.array_equal(f.get_descriptors(), ref["bank"])                       # refreshed exactly where accepted
    # the measurements are installed: RANSAC + updates run straight on
    m = len(out["meas_idx"])
    hyp = np.stack([rng.permutation(m)[:3] for _ in range(8)]).astype(np.int32)
    r = f.ransac_hypotheses(hyp, threshold=1.0)
    assert r["li_mask"].sum() >= 3
    f.close()


def test_ic_search_edges(pre3, orc):
    N = 20
    rng, seq, bank = _scene(N, 29)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    with pytest.raises(pre3.Pre3Error):
        f.matching_sift_based()                              # no descriptors yet
    f.set_descriptors(bank)
    with pytest.raises(pre3.Pre3Error):
        f.matching_sift_based()                              # no prediction yet
    f.ekf_prediction(s["u"])
    f.load_scan(np.zeros((128, 0)), np.zeros((4, 0)))        # empty scan: no matches, no measurements
    out = f.matching_sift_based()
    assert out["match_idx"].shape == (2, 0) and len(out["meas_idx"]) == 0 and f.m == 0
    f.load_scan(bank[:, :1].copy(), np.array([[10.0], [10.0], [1.0], [0.0]]))   # one keypoint: second-best is +inf -> matches
    out = f.matching_sift_based()
    x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()
    ref = orc.ic_search(types, off, x1, P1, seq["cam"], bank, bank[:, :1].copy(), np.array([[10.0], [10.0], [1.0], [0.0]]))
    assert np.array_equal(out["match_idx"], ref["match_idx"]) and np.array_equal(out["meas_idx"], ref["meas_idx"])
    f.close()


@pytest.mark.parametrize("N,K2", [(37, 45), (65, 33), (96, 640)])
def test_fused_route_on_ragged_sizes(pre3, orc, N, K2):
    """the two-launch route's 32 x 32 pair tiles on sizes that are not multiples of 32 (ragged last tiles both ways; 96 x 640: twenty column tiles,
    two batches of the gate's merge) against the oracle's composition, bit for bit"""
    rng, seq, bank = _scene(N, 41 + N)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8)
    f.set_x_p_k_k(seq["x0"], seq["P0"]); f.set_descriptors(bank); f.ekf_prediction(s["u"])
    x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()
    h, has_h = orc.project(types, off, x1, seq["cam"])
    n_seen = int(has_h.sum())
    sd, sp = _scan(rng, h, has_h, bank, max(0, K2 - n_seen), frac_seen=1.0, px_sigma=3.0)
    sd, sp = sd[:, :K2].copy(), sp[:, :K2].copy()
    f.load_scan(sd, sp)
    out = f.matching_sift_based(1.5, strict_reference=True)
    assert f.ic_search_route() == 2
    ref = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp, 1.5, True)
    assert ref["match_idx"].shape[1] > 5
    assert np.array_equal(out["match_idx"], ref["match_idx"]) and np.array_equal(out["accepted"], ref["accepted"])
    assert np.array_equal(out["meas_idx"], ref["meas_idx"]) and np.array_equal(out["z"], ref["z"])
    assert np.array_equal(f.get_descriptors(), ref["bank"])
    f.close()


def test_fused_route_with_no_predicted_landmark(pre3, orc):
    """the camera looks away from the map (a half turn about the vertical axis): index_in_info is empty, no match, no measurement -- and the next
    search, with the camera back, is unaffected"""
    N = 40
    rng, seq, bank = _scene(N, 43)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8)
    x_away = seq["x0"].copy(); x_away[3:7] = [0.0, 0.0, 1.0, 0.0]
    f.set_x_p_k_k(x_away, seq["P0"]); f.set_descriptors(bank); f.ekf_prediction(np.array([0, 0, 0, 1.0, 0, 0, 0]))
    h, has_h = orc.project(types, off, f.get_x_k_km1(), seq["cam"])
    assert has_h.sum() == 0
    sd = np.abs(rng.normal(0, 1, (128, 50))); sd /= np.linalg.norm(sd, axis=0)
    sp = np.stack([rng.uniform(1, 176, 50), rng.uniform(1, 144, 50), np.full(50, 2.0), np.zeros(50)])
    f.load_scan(sd, sp)
    out = f.matching_sift_based(1.5)
    assert f.ic_search_route() == 2 and out["match_idx"].shape == (2, 0) and len(out["meas_idx"]) == 0 and f.m == 0
    assert np.array_equal(f.get_descriptors(), bank)
    f.set_x_p_k_k(seq["x0"], seq["P0"]); f.ekf_prediction(s["u"])
    x1, P1 = f.get_x_k_km1(), f.get_p_k_km1()
    h, has_h = orc.project(types, off, x1, seq["cam"])
    sd, sp = _scan(rng, h, has_h, bank, 20, px_sigma=3.0)
    f.load_scan(sd, sp)
    out = f.matching_sift_based(1.5)
    ref = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp, 1.5, True)
    assert ref["match_idx"].shape[1] > 5 and np.array_equal(out["match_idx"], ref["match_idx"]) and np.array_equal(out["meas_idx"], ref["meas_idx"])
    f.close()


def test_descriptor_bank_follows_the_map(pre3, orc):
    N = 16
    rng, seq, bank = _scene(N, 31)
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], types, dtype="f64", max_hyp=8, max_landmarks=N + 2)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.set_descriptors(bank)
    f.delete_features([1, 7])
    keep = [i for i in range(N) if i not in (1, 7)]
    assert np.array_equal(f.get_descriptors(), bank[:, keep])
    f.add_features_inverse_depth(np.array([[40.0, 50.0]]), 1.0, 0.5)
    d = f.get_descriptors()
    assert np.array_equal(d[:, :-1], bank[:, keep]) and not d[:, -1].any()
    new = np.full((128, 1), 0.25)
    f.set_descriptors(new, first=f.N - 1)
    f.inversedepth_2_cartesian(1e9)                          # convert everything: the bank is untouched
    assert np.array_equal(f.get_descriptors(), np.concatenate([bank[:, keep], new], 1))
    f.close()
