"""GPU: libpre3's own RCCL communicator (include/pre3.h "the RCCL communicator", 3pre_amd/comm.py) with ONE rank -- all a one-GPU box can
hold (RCCL refuses two ranks on one device; tests/test_gpu_dist.py runs the 2-rank form where there are two GPUs).  The collectives really
run (ncclAllReduce / ncclAllGather over one rank, enqueued by the library on its own stream), so everything but the wire is exercised:
binding librccl at run time, communicator creation, the on-stream round and its single wait."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

synth = importlib.import_module("3pre_amd.synth")


@pytest.fixture(scope="module")
def comm(pre3):
    cm = importlib.import_module("3pre_amd.comm")
    c = cm.Comm(0, cm.unique_id(), 0, 1)
    yield c
    c.close()


def test_communicator_binds_rccl_at_run_time(comm):
    info = comm.info()
    assert info["rank"] == 0 and info["world"] == 1
    assert info["rccl_version"] >= 20000 and "rccl" in info["library"]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_on_stream_sharded_round_equals_the_unsharded_round(pre3, comm, dtype):
    N, n_draw = 120, 70
    seq = synth.make_sequence(N, 2, n_draw, seed=21)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=dtype, max_hyp=n_draw)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    with pytest.raises(pre3.Pre3Error) as e:                      # no communicator yet
        f.ekf_prediction(seq["steps"][0]["u"]); f.search_IC_matches(); f.set_measurements(seq["steps"][0]["meas_idx"], seq["steps"][0]["z"])
        f.ransac_sharded_stream(seq["steps"][0]["hyp"], 1.0)
    assert e.value.code == -4
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.set_comm(comm)
    for s in seq["steps"]:
        for early in (True, False):
            f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
            ref = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=early)
            x_ref = None
            got = f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=early)
            for key in ("best", "iters", "n_hyp", "max_support"):
                assert got[key] == ref[key], key
            assert np.array_equal(got["support"], ref["support"]) and np.array_equal(got["li_mask"], ref["li_mask"])
            st = f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=early, fetch=False)       # statistics through the mailbox only
            assert st["best"] == ref["best"] and st["max_support"] == ref["max_support"]
            # the update that follows takes the winner left on the device
            f.ekf_update_li_inliers()
            assert np.isfinite(f.get_p_k_k()).all()
            f.set_x_p_k_k(f.get_x_k_k(), f.get_p_k_k())
    f.set_comm(None)
    f.close()


def test_a_rank_local_failure_still_enters_the_collective(pre3, comm):
    """pre3_ransac_sharded with a draw table this rank rejects (a position outside the IC list): the call fails with the rank-local error AFTER
    the all-reduce has been enqueued -- its peers, who are in ncclAllReduce by then, are not left waiting -- and the word that counts missing slices
    goes round with the sums; the context and the communicator stay usable, the next round is correct."""
    N, n_draw = 60, 24
    seq = synth.make_sequence(N, 1, n_draw, seed=33)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_draw)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.set_comm(comm)
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    bad = s["hyp"].copy(); bad[3, 1] = len(s["meas_idx"]) + 5
    with pytest.raises(pre3.Pre3Error) as e:
        f.ransac_sharded_stream(bad, 1.0)
    assert e.value.code == -1
    assert comm.info()["world"] == 1                               # the communicator was not aborted
    ref = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)
    got = f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=False)
    assert got["best"] == ref["best"] and np.array_equal(got["support"], ref["support"]) and np.array_equal(got["li_mask"], ref["li_mask"])
    f.set_comm(None)
    f.close()


def test_on_stream_shard_match_equals_the_oracle(pre3, orc, comm):
    mt = importlib.import_module("3pre_amd.matcher")
    rng = np.random.default_rng(5)
    L1 = rng.integers(0, 255, (128, 333)).astype(np.uint8)
    L2 = rng.integers(0, 255, (128, 700)).astype(np.uint8)
    L2[:, 50:250] = L1[:, :200]
    L2[:, 600] = L1[:, 3]
    real1, real2 = L1 + rng.random(L1.shape), L2 + rng.random(L2.shape)
    for A, B in ((L1, L2), (real1, real2)):
        sh = mt.MatchShard(A, B, 0)
        for with_comm in (False, True):
            sh.set_comm(comm if with_comm else None)
            for thr in (1.5, 1.1, 1.5):
                m, d = sh.match(thr, return_scores=True)
                mr, dr = orc.siftmatch(A, B, thr)
                assert np.array_equal(m, mr) and np.array_equal(d, dr), (A.dtype, with_comm, thr)
        sh.close()


@pytest.fixture(autouse=True)
def _test_hooks_on(monkeypatch):
    """the stall hooks are inert unless the environment enables them (3pre_amd/csrc/pre3_test_hooks.h)"""
    monkeypatch.setenv("PRE3_TEST_HOOKS", "1")


def test_the_stall_hooks_are_inert_without_the_environment_switch(pre3, monkeypatch):
    monkeypatch.delenv("PRE3_TEST_HOOKS", raising=False)
    N = 20
    seq = synth.make_sequence(N, 1, 8, seed=3)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=8)
    with pytest.raises(pre3.Pre3Error) as e:
        f.test_stall(0)
    assert e.value.code == -4                                                  # PRE3_E_STATE: nothing was parked
    f.sync()
    f.close()


@pytest.mark.timeout(120)
def test_every_drain_behind_a_parked_stream_has_the_deadline(pre3):
    """Round-5 advisor: stage_wait (pre3_set_scan, pre3_set_descriptors, map management), pre3_set_state and pre3_set_comm used to end in a bare
    hipStreamSynchronize.  With a communicator on the context every drain of the stream is bounded by its deadline (stream_drain): behind a parked
    kernel they return PRE3_E_COMM at the deadline, the communicator is aborted, and once the stream has drained the context works again."""
    import time
    cm = importlib.import_module("3pre_amd.comm")
    N, n_draw = 60, 24
    seq = synth.make_sequence(N, 1, n_draw, seed=35)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_draw)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    c1 = cm.Comm(0, cm.unique_id(), 0, 1)
    c1.set_timeout(300)
    f.set_comm(c1)
    f.test_stall(0)
    t0 = time.time()
    with pytest.raises(pre3.Pre3Error) as e:
        f.set_x_p_k_k(seq["x0"], seq["P0"])                                    # pre3_set_state drains the stream first
    dt = time.time() - t0
    assert e.value.code == -7 and 0.25 < dt < 5.0, (e.value.code, dt)
    t0 = time.time()
    with pytest.raises(pre3.Pre3Error) as e2:                                  # ... and so does every later drain while the stream is still parked
        f.sync()
    assert e2.value.code == -7 and time.time() - t0 < 5.0
    f.test_stall(1)
    time.sleep(0.05)
    c2 = cm.Comm(0, cm.unique_id(), 0, 1)
    f.set_comm(c2)                                                             # drains (now possible), joins the abort, installs the fresh handle
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    ref = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)
    got = f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=False)
    assert got["best"] == ref["best"] and np.array_equal(got["support"], ref["support"])
    f.set_comm(None)
    f.close(); c1.close(); c2.close()


@pytest.mark.timeout(120)
def test_a_stalled_collective_ends_in_the_deadline_not_in_a_hang(pre3):
    """A peer that stalls inside (or never enters) a collective: the library's host wait has a wall-clock deadline (pre3_comm_set_timeout).  With one
    rank the stall is a kernel parked in front of the round on the context's stream (test hook); the sharded round must return PRE3_E_COMM
    within the deadline with the communicator aborted -- never sit in a stream synchronisation -- and the context must work again with a fresh
    communicator once the stream has drained."""
    import time
    cm = importlib.import_module("3pre_amd.comm")
    N, n_draw = 60, 24
    seq = synth.make_sequence(N, 1, n_draw, seed=34)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_draw)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    c1 = cm.Comm(0, cm.unique_id(), 0, 1)
    c1.set_timeout(300)
    f.set_comm(c1)
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    ref = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)
    f.test_stall(0)
    t0 = time.time()
    with pytest.raises(pre3.Pre3Error) as e:
        f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=False)
    dt = time.time() - t0
    assert e.value.code == -7 and 0.25 < dt < 5.0, (e.value.code, dt)          # PRE3_E_COMM, at the deadline
    with pytest.raises(pre3.Pre3Error) as e2:                                  # the aborted communicator is unusable ...
        f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=False)
    assert e2.value.code == -7
    f.test_stall(1)                                                            # ... the "peer" comes back: the stream drains
    f.sync()
    c2 = cm.Comm(0, cm.unique_id(), 0, 1)
    f.set_comm(c2)                                                             # a fresh communicator: the context works again
    got = f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=False)
    assert got["best"] == ref["best"] and np.array_equal(got["support"], ref["support"]) and np.array_equal(got["li_mask"], ref["li_mask"])
    f.set_comm(None)
    f.close(); c1.close(); c2.close()


@pytest.mark.timeout(120)
def test_the_sharded_match_has_the_same_deadline_and_a_failing_rank_still_enters_the_all_gather(pre3, orc):
    import time
    cm = importlib.import_module("3pre_amd.comm")
    mt = importlib.import_module("3pre_amd.matcher")
    rng = np.random.default_rng(6)
    L1 = rng.integers(0, 255, (128, 300)).astype(np.uint8)
    L2 = rng.integers(0, 255, (128, 640)).astype(np.uint8)
    L2[:, 40:200] = L1[:, :160]
    sh = mt.MatchShard(L1, L2, 0)
    c1 = cm.Comm(0, cm.unique_id(), 0, 1)
    sh.set_comm(c1)
    mr = orc.siftmatch(L1, L2, 1.5)[0]
    assert np.array_equal(sh.match(1.5), mr)
    # (i) this rank's distance kernels fail: it still enters the all-gather, its slice marked missing -> the match is void (PRE3_E_HIP here, the
    #     rank-local error; its peers would see PRE3_E_COMM from the merge), and the NEXT match is correct again
    sh.test_stall(2)
    with pytest.raises(pre3.Pre3Error) as e:
        sh.match(1.5)
    assert e.value.code == -3
    assert np.array_equal(sh.match(1.5), mr)
    # (ii) a stall in front of the collective: PRE3_E_COMM at the deadline, communicator aborted, usable again with a fresh one
    c1.set_timeout(300)
    sh.test_stall(0)
    t0 = time.time()
    with pytest.raises(pre3.Pre3Error) as e:
        sh.match(1.5)
    assert e.value.code == -7 and 0.25 < time.time() - t0 < 5.0
    sh.test_stall(1)
    c2 = cm.Comm(0, cm.unique_id(), 0, 1)
    sh.set_comm(c2)
    assert np.array_equal(sh.match(1.5), mr)
    sh.set_comm(None)
    sh.close(); c1.close(); c2.close()
