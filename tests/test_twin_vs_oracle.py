"""CPU: the two restatements agree where no reference artefact pins them -- the numpy twin (the checker of the full-size GPU tests)
against the C oracle (the checker everywhere else) on a whole step at N=120, and the oracle's OpenMP loops are bit-reproducible."""
import importlib

import numpy as np

synth = importlib.import_module("3pre_amd.synth")


def test_twin_equals_c_oracle_on_a_full_step_at_N120(orc):
    from oracle import np_twin as tw
    N, n_hyp = 120, 60
    seq = synth.make_sequence(N, 1, n_hyp)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    for ee in (False, True):
        a = orc.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0, early_exit=ee)
        b = tw.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0, early_exit=ee)
        assert np.array_equal(a["li"], b["li"]) and np.array_equal(a["hi"], b["hi"])
        for key in ("best", "iters", "n_hyp", "max_support"):
            assert a["ransac"][key] == b["ransac"][key]
        assert np.array_equal(a["ransac"]["support"], b["ransac"]["support"])
        assert np.abs(a["x_kk"] - b["x_kk"]).max() < 1e-12
        assert np.abs(a["P_kk"] - b["P_kk"]).max() < 1e-12 * np.abs(a["P_kk"]).max()


def test_c_oracle_is_bit_identical_for_any_thread_count(orc):
    N = 60
    seq = synth.make_sequence(N, 1, 20, seed=8)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    outs = []
    for t in (1, 4):
        orc.lib().orc_set_threads(t)
        outs.append(orc.step(types, off, seq["cam"], seq["x0"], seq["P0"], s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0))
    orc.lib().orc_set_threads(0)
    assert np.array_equal(outs[0]["P_kk"], outs[1]["P_kk"]) and np.array_equal(outs[0]["x_kk"], outs[1]["x_kk"])
