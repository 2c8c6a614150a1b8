"""GPU worker of tests/test_gpu_dist.py: 2 ranks (gloo rendezvous, both on cuda:0) run the real sharded RANSAC
(3pre_amd/dist.ransac_sharded: GPU scoring of a hypothesis slice, all-reduce of supports + masks, replay on the
GPU) and the real sharded matcher, and compare with the unsharded GPU result and the oracle."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend = os.environ.get("PRE3_TEST_BACKEND", "gloo")          # "nccl" (= RCCL): one GPU per rank, needs >= world GPUs
    dev = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    import oracle as orc
    pre3 = importlib.import_module("3pre_amd")
    pd = importlib.import_module("3pre_amd.dist")
    synth = importlib.import_module("3pre_amd.synth")
    # libpre3's own RCCL communicator (3pre_amd/comm.py): one GPU per rank only -- RCCL refuses two ranks on one device
    comm = None
    if backend == "nccl":
        comm = importlib.import_module("3pre_amd.comm").Comm.from_torch_distributed(dev)
        assert comm.info()["world"] == world
    N, n_draw = 60, 45
    seq = synth.make_sequence(N, 1, n_draw, seed=8)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    for dtype in ("f64", "f32"):
        f = pre3.EkfFilter(seq["cam"], types, dtype=dtype, max_hyp=n_draw, device=dev)
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.ekf_prediction(s["u"])
        f.search_IC_matches()
        f.set_measurements(s["meas_idx"], s["z"])
        ref = f.ransac_hypotheses(s["hyp"], threshold=1.0, early_exit=True)          # unsharded, same GPU
        got = pd.ransac_sharded(f, s["hyp"], 1.0, early_exit=True)
        for key in ("best", "iters", "n_hyp", "max_support"):
            assert got[key] == ref[key], (key, got[key], ref[key])
        assert np.array_equal(got["li_mask"], ref["li_mask"]) and np.array_equal(got["support"], ref["support"])
        if comm is not None:
            # the same round with the collective enqueued by libpre3 itself on the filter's stream (pre3_ransac_sharded)
            f.set_comm(comm)
            got2 = f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=True)
            for key in ("best", "iters", "n_hyp", "max_support"):
                assert got2[key] == ref[key], (key, got2[key], ref[key])
            assert np.array_equal(got2["li_mask"], ref["li_mask"]) and np.array_equal(got2["support"], ref["support"])
        # and the LI update that follows uses the reduced winner on every rank
        f.ekf_update_li_inliers()
        P = f.get_p_k_k()
        t = torch.from_numpy(P.copy())
        if backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert np.array_equal(t.cpu().numpy(), P), "ranks diverged after the replicated update"
        f.close()
    rng = np.random.default_rng(3)
    L1 = rng.integers(0, 255, (128, 130)).astype(np.uint8)
    L2 = rng.integers(0, 255, (128, 517)).astype(np.uint8)
    L2[:, 100:200] = L1[:, :100]
    for dt in (np.uint8, np.float32):
        m, d = pd.siftmatch_sharded(L1.astype(dt), L2.astype(dt), 1.5, return_scores=True)
        mr, dr = orc.siftmatch(L1.astype(dt), L2.astype(dt), 1.5)
        assert np.array_equal(m, mr) and np.array_equal(d, dr)
    # the device-resident sharded matcher (matcher.MatchShard): slice packed in HBM, partials gathered as tensors, merge on the device
    mt = importlib.import_module("3pre_amd.matcher")
    lo, hi = pd.shard_range(L2.shape[1], rank, world)
    sh = mt.MatchShard(L1, L2[:, lo:hi], lo, device=dev)
    for thr in (1.5, 1.1):
        m, d = pd.siftmatch_sharded_resident(sh, thr, return_scores=True)
        mr, dr = orc.siftmatch(L1, L2, thr)
        assert np.array_equal(m, mr) and np.array_equal(d, dr), "resident sharded matcher differs from the oracle"
    if comm is not None:
        sh.set_comm(comm)
        for thr in (1.5, 1.1):
            m, d = sh.match(thr, return_scores=True)               # run + ncclAllGather + merge on the shard's stream
            mr, dr = orc.siftmatch(L1, L2, thr)
            assert np.array_equal(m, mr) and np.array_equal(d, dr), "on-stream sharded matcher differs from the oracle"
    sh.close()
    if comm is not None:
        comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d/%d OK" % (rank, world))


if __name__ == "__main__":
    main()
