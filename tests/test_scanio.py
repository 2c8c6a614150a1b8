"""CPU: the SIFT_result%04d.mat wire format (SIFT_extract_save.m:44-106, matching_sift_based.m:55,78) and the oracle's
IC-search composition (quirk Q5 on/off, descriptor refresh only where accepted)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_sift_result_wire_format_round_trip(tmp_path):
    scanio = importlib.import_module("3pre_amd.scanio")
    rng = np.random.default_rng(3)
    scan = dict(idxScan=12, Image=rng.integers(0, 255, (144, 176)).astype(np.uint8), Descriptor_RAW=rng.uniform(0, 0.3, (128, 37)),
                SCALE_ORIENT_POS_RAW=rng.uniform(1, 140, (4, 37)), Descriptor=rng.uniform(0, 0.3, (128, 30)),
                SCALE_ORIENT_POS=rng.uniform(1, 140, (4, 30)), XYZ_DATA=rng.normal(0, 1, (3, 30)))
    p = scanio.sift_result_path(str(tmp_path) + "/", 12)
    assert p.endswith("FeatureExtractionMatching/SIFT_result0012.mat")
    scanio.save_sift_result(p, scan)
    back = scanio.load_sift_result(p)
    for k, v in scan.items():
        assert np.array_equal(np.asarray(back[k]), np.asarray(v)), k


def test_oracle_ic_search_quirk_and_refresh(orc):
    import test_gpu_icsearch as t
    N = 60
    rng, seq, bank = t._scene(N, 23)
    types, off, n = orc.landmark_table(np.zeros(N, int))
    x1, P1 = orc.predict(seq["x0"], seq["P0"], seq["steps"][0]["u"])
    h, has = orc.project(types, off, x1, seq["cam"])
    sd, sp = t._scan(rng, h, has, bank, 60, px_sigma=5.0)
    a = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp, 1.5, True)
    b = orc.ic_search(types, off, x1, P1, seq["cam"], bank, sd, sp, 1.5, False)
    assert np.array_equal(a["match_idx"], b["match_idx"]) and not np.array_equal(a["accepted"], b["accepted"])
    for r in (a, b):
        k1, k2 = r["match_idx"]
        lm = r["pred"][k1]
        changed = np.nonzero((r["bank"] != bank).any(0))[0]
        assert np.array_equal(changed, np.sort(lm[r["accepted"] > 0])) and np.array_equal(r["meas_idx"], changed)
        for c in np.nonzero(r["accepted"])[0]:
            assert np.array_equal(r["bank"][:, lm[c]], sd[:, k2[c]])
        # corrected gate: every accepted pixel lies inside ITS OWN landmark's window
        if r is b:
            S = orc.innovation(types, off, P1, *orc.jacobian(types, off, x1, seq["cam"], h, has), has)
            for c in np.nonzero(r["accepted"])[0]:
                assert np.linalg.norm(sp[0:2, k2[c]] - h[lm[c]]) <= np.ceil(3 * np.sqrt(S[lm[c], 0, 0]))
