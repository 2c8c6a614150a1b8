"""CPU: the synthetic sequence's hypothesis tables stay valid when the truth has walked away from the map (a 30 000-step run of
bench.py once ended in "pre3_step: bad hypothesis table": fewer than k measurements gave a table with fewer than k -- finally 0 -- columns)."""
import importlib

import numpy as np

synth = importlib.import_module("3pre_amd.synth")


def test_draws_follow_select_random_match_for_small_sets():
    rng = np.random.default_rng(0)
    for m in (0, 1, 2, 3):
        h = synth.draw_hypotheses(rng, m, 7, k=3)
        assert h.shape == (7, 1) and h.dtype == np.int32                   # select_random_match.m:47-51: one landmark unless #IC > 3
        assert (h >= 0).all() and (h < max(m, 1)).all()
    h = synth.draw_hypotheses(rng, 4, 7, k=3)
    assert h.shape == (7, 3) and all(len(set(r)) == 3 and max(r) < 4 for r in h.tolist())


def test_a_sequence_with_hardly_any_visible_landmark_still_has_valid_tables():
    seq = synth.make_sequence(6, 4, 5, meas_frac=0.2)                          # 6 landmarks, about one measured per frame
    for s in seq["steps"]:
        m, (n_draw, k) = len(s["meas_idx"]), s["hyp"].shape
        assert n_draw == 5 and k >= 1 and (k == 3 if m > 3 else k == 1)
