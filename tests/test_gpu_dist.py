"""GPU: the sharded RANSAC and matcher with 2 ranks sharing the one GPU of the test box (gloo rendezvous; the
nccl/RCCL branch of 3pre_amd/dist.py needs >= 2 GPUs and is exercised by bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_two_ranks(backend):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1", PRE3_TEST_BACKEND=backend,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               PRE3_CHOL_FORM="0")      # two processes on ONE device: no persistent launches (they could interleave)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == 2


def test_sharded_ransac_and_matcher_two_ranks_one_gpu():
    _run_two_ranks("gloo")


def test_sharded_ransac_and_matcher_two_ranks_rccl():
    """the nccl (= RCCL) branch itself: the in-place all-reduce on the zero-copy view, the device all-gather of the matcher partials, and
    bit-identical replicas after the update -- needs two GPUs (RCCL refuses two ranks on one device), so it runs on multi-GPU nodes only"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    _run_two_ranks("nccl")


def test_zero_copy_device_view_round_trip():
    """The nccl branch of dist.ransac_sharded all-reduces IN PLACE on a torch view of the context's own support / mask buffer
    (__cuda_array_interface__).  With one GPU the collective itself cannot run, but everything around it can: the view must alias
    libpre3's memory in both directions, so that on an 8-GPU node the only untested line is the all-reduce."""
    import importlib
    import numpy as np
    import torch
    pre3 = importlib.import_module("3pre_amd")
    pd = importlib.import_module("3pre_amd.dist")
    synth = importlib.import_module("3pre_amd.synth")
    N, n_draw = 60, 40
    seq = synth.make_sequence(N, 1, n_draw, seed=8)
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_draw)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    f.ekf_prediction(s["u"]); f.search_IC_matches(); f.set_measurements(s["meas_idx"], s["z"])
    ref = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=False)
    sup_ptr, msk_ptr, words = f.ransac_score_shard(s["hyp"], 1.0, 0, n_draw)
    gap = (msk_ptr - sup_ptr) // 4
    assert n_draw <= gap <= n_draw + 4                         # one allocation: supports, padding, masks
    both = pd.dev_tensor(sup_ptr, gap + n_draw * words)
    assert both.is_cuda and both.dtype == torch.int32 and both.data_ptr() == sup_ptr
    assert np.array_equal(both[:n_draw].cpu().numpy(), ref["support"])          # the view reads libpre3's results
    both[:n_draw] += 0                                         # an in-place torch op on libpre3's memory (what all_reduce does)
    torch.cuda.synchronize()
    got = f.ransac_select(n_draw, 3, early_exit=False)
    assert got["best"] == ref["best"] and np.array_equal(got["li_mask"], ref["li_mask"])
    # and a write through the view is what the selection then sees
    both[:n_draw] = 0
    both[7] = 999
    torch.cuda.synchronize()
    got = f.ransac_select(n_draw, 3, early_exit=False)
    assert got["best"] == 7 and got["max_support"] == 999
    f.close()


def test_resident_matcher_shards_equal_the_unsharded_match(orc=None):
    """G = 1, 2, 3, 5 database shards on one GPU: partials concatenated on the device, merged on the device -> the oracle's match list"""
    import importlib
    import numpy as np
    import torch
    import oracle
    oracle.build()
    pd = importlib.import_module("3pre_amd.dist")
    mt = importlib.import_module("3pre_amd.matcher")
    rng = np.random.default_rng(11)
    L1 = rng.integers(0, 255, (128, 333)).astype(np.uint8)
    L2 = rng.integers(0, 255, (128, 700)).astype(np.uint8)
    L2[:, 50:250] = L1[:, :200]
    L2[:, 600] = L1[:, 3]                                       # a duplicate of a matched column: ties -> lowest index, ratio test fails
    # uint8; the same descriptors as doubles (what Lowe-format SIFT files such as box.sift hold (matching_sift_based.m:104-118 itself passes unit-norm real-valued doubles): integer-valued -> the int8 route inside the shard);
    # real-valued doubles and singles (bf16 rank + exact re-evaluation inside the shard)
    real1, real2 = L1 + rng.random(L1.shape), L2 + rng.random(L2.shape)
    real2[:, 600] = real1[:, 3]; real2[:, 53] = real1[:, 3]
    for A, B, Gs in ((L1, L2, (1, 2, 3, 5)), (L1.astype(np.float64), L2.astype(np.float64), (1, 3)), (real1, real2, (1, 2, 3)),
                     (real1.astype(np.float32), real2.astype(np.float32), (2,))):
        mr, dr = oracle.siftmatch(A, B, 1.5)
        for G in Gs:
            parts, shards = [], []
            for g in range(G):
                lo, hi = pd.shard_range(B.shape[1], g, G)
                sh = mt.MatchShard(A, B[:, lo:hi], lo)
                ptr, n = sh.run()
                parts.append(pd.dev_tensor(ptr, n, "<f8").clone())
                shards.append(sh)
            allp = torch.cat(parts)
            torch.cuda.synchronize()
            m, d = shards[0].merge(G, allp.data_ptr(), 1.5, return_scores=True)
            assert np.array_equal(m, mr) and np.array_equal(d, dr), (A.dtype, G)
            for sh in shards:
                sh.close()
    # a slice too small for the matrix-core paths of the float classes is refused (the host-array form serves it)
    import pytest
    with pytest.raises(Exception):
        mt.MatchShard(real1[:, :10], real2[:, :20], 0)
