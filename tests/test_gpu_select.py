"""GPU: the RANSAC selection stage (ransac_hypotheses.m:40-80 replayed on the supports) on crafted support vectors.
Round 4 turned the walk over the improvements into two reductions (select_find_best, pre3_geom.hip: the first index whose support passes the
exit test, then the first maximum up to there): ties, an exit in the middle, no inlier at all, more than 1000 / 1024 draws, with and without
the early exit, against the host replay of the reference's loop (tests/dist_worker.py::replay)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_selection_on_crafted_supports():
    import torch
    pre3 = importlib.import_module("3pre_amd")
    pd = importlib.import_module("3pre_amd.dist")
    synth = importlib.import_module("3pre_amd.synth")
    from dist_worker import replay
    N, n_draw, k = 60, 1500, 3
    seq = synth.make_sequence(N, 1, n_draw, seed=21)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_draw)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    m = len(s["meas_idx"])
    sup_ptr, msk_ptr, words = f.ransac_score_shard(s["hyp"], 1.0, 0, n_draw)
    gap = (msk_ptr - sup_ptr) // 4
    both = pd.dev_tensor(sup_ptr, gap + n_draw * words)
    masks = both[gap:gap + n_draw * words].cpu().numpy().view(np.uint32).copy()
    rng = np.random.default_rng(5)
    cases = []
    cases.append(("scored", both[:n_draw].cpu().numpy().copy()))
    cases.append(("random", rng.integers(0, m // 2, n_draw).astype(np.int32)))
    z = np.zeros(n_draw, np.int32); cases.append(("no inlier at all", z))
    t = rng.integers(0, 10, n_draw).astype(np.int32); t[[300, 900, 1400]] = 25; cases.append(("ties of the maximum", t))
    e = rng.integers(0, m // 3, n_draw).astype(np.int32); e[700] = m; e[40] = m - 1; cases.append(("exit in the middle", e))
    e2 = rng.integers(0, m // 3, n_draw).astype(np.int32); e2[1200] = m; cases.append(("exit candidate beyond the 1000-draw limit", e2))
    inc = np.minimum(np.arange(n_draw) // 40, m - 2).astype(np.int32); cases.append(("many improvements", inc))
    for name, sup in cases:
        for ee in (False, True):
            both[:n_draw] = torch.from_numpy(sup).to(both.device)
            torch.cuda.synchronize()
            got = f.ransac_select(n_draw, k, early_exit=ee)
            if sup.max() == 0:
                assert got["best"] == -1 and got["max_support"] == 0 and got["li_mask"].sum() == 0, (name, ee, got)
                continue
            ref = replay(sup, masks, words, m, k, ee)
            for key in ("best", "iters", "n_hyp", "max_support"):
                assert got[key] == ref[key], (name, ee, key, got[key], ref[key])
            assert np.array_equal(got["li_mask"], ref["li_mask"]), (name, ee)
    f.close()
