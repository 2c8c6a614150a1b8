"""Worker of tests/test_dist_cpu.py: run under torch.distributed.run with the gloo backend (CPU).
Drives the SAME sharding / collective / merge code as the GPU path (3pre_amd/dist.py) with the oracle standing
in for the kernels, and checks the result against the unsharded oracle on every rank."""
import importlib
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def replay(support, masks, words, m, k, early_exit):
    """ransac_hypotheses.m:40-80 termination rule on a support vector (host twin of k_ransac_select)"""
    n_hyp, best, mx, iters = 1000, -1, 0, 0
    n_draw = len(support)
    limit = min(n_draw, 1000) if early_exit else n_draw
    for it in range(limit):
        if early_exit and n_hyp == 0:
            break
        iters += 1
        if support[it] > mx:
            mx, best = int(support[it]), it
            eps = 1 - mx / m
            with np.errstate(divide="ignore"):
                n_hyp = int(np.ceil(np.log(1 - 0.99) / np.log(1 - (1 - eps))))
        if early_exit and n_hyp <= k:
            break
    row = masks.reshape(n_draw, words)[best].astype(np.uint32)
    li = np.array([(row[j >> 5] >> (j & 31)) & 1 for j in range(m)], np.int32)
    return dict(best=best, iters=iters, n_hyp=n_hyp, max_support=mx, li_mask=li)


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import oracle as orc
    pd = importlib.import_module("3pre_amd.dist")
    synth = importlib.import_module("3pre_amd.synth")
    pre3 = importlib.import_module("3pre_amd")

    # ---- shard_range covers [0, n) exactly once
    for n in (0, 1, 7, 200, 1001):
        cover = []
        for r in range(world):
            lo, hi = pd.shard_range(n, r, world)
            cover += list(range(lo, hi))
        assert cover == list(range(n))

    # ---- C1: sharded RANSAC == unsharded oracle (identical on every rank)
    N, n_draw, k = 40, 37, 3
    seq = synth.make_sequence(N, 1, n_draw, seed=5)
    s = seq["steps"][0]
    types, off, n = orc.landmark_table(np.zeros(N, int))
    x1, P1 = orc.predict(seq["x0"], seq["P0"], s["u"])
    h, has = orc.project(types, off, x1, seq["cam"])
    Hc, Hl = orc.jacobian(types, off, x1, seq["cam"], h, has)
    meas = s["meas_idx"]
    m = len(meas)
    z = np.zeros((N, 2))
    z[meas] = s["z"]
    words = (m + 31) // 32

    def score_slice(lo, hi):
        sup = np.zeros(n_draw, np.int32)
        msk = np.zeros(n_draw * words, np.uint32)
        for it in range(lo, hi):
            sel = [meas[p] for p in s["hyp"][it]]
            xi = orc.hypothesis_state(sel, types, off, x1, P1, Hc, Hl, z, h)
            cnt, mask, _ = orc.support(meas, types, off, xi, seq["cam"], z[meas], 1.0)
            sup[it] = cnt
            for j in np.nonzero(mask)[0]:
                msk[it * words + (j >> 5)] |= np.uint32(1 << (j & 31))
        return sup, msk.view(np.int32)

    for ee in (False, True):
        got = pd.ransac_sharded_generic(score_slice, lambda su, ma: replay(su, ma.view(np.uint32), words, m, k, ee), n_draw, words)
        ref = orc.ransac(types, off, x1, P1, Hc, Hl, z, h, meas, meas, seq["cam"], s["hyp"], 1.0, early_exit=ee)
        for key in ("best", "iters", "n_hyp", "max_support"):
            assert got[key] == ref[key], (key, got[key], ref[key])
        assert np.array_equal(got["li_mask"], ref["li_mask"])

    # ---- C2: sharded matcher == unsharded oracle; the merge is the product's own (host-side) pre3_siftmatch_merge
    rng = np.random.default_rng(11)
    K1, K2 = 90, 301
    L1 = rng.integers(0, 200, (128, K1)).astype(np.uint8)
    L2 = rng.integers(0, 200, (128, K2)).astype(np.uint8)
    L2[:, 50:120] = L1[:, :70]
    L2[:, 300] = L1[:, 0]                  # cross-shard tie: the lowest global index must win

    def partial(A, B, offset):
        a, b = A.astype(np.int64).T, B.astype(np.int64).T
        best = np.full(len(a), 2147483647.0)
        second = best.copy()
        arg = np.full(len(a), -1, np.int32)
        if len(b):
            d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(2)
            order = np.argsort(d, axis=1, kind="stable")
            best = d[np.arange(len(a)), order[:, 0]].astype(float)
            arg = (order[:, 0] + offset).astype(np.int32)
            if d.shape[1] > 1:
                second = d[np.arange(len(a)), order[:, 1]].astype(float)
        return best, second, arg

    merge = lambda dt, B, S, A, th: pre3.siftmatch_merge(dt, B, S, A, th, return_scores=True)
    for th in (1.5, 1.0):
        mt, sc = pd.siftmatch_sharded(L1, L2, th, partial=partial, merge=merge, return_scores=True)
        mr, sr = orc.siftmatch(L1, L2, th)
        assert np.array_equal(mt, mr) and np.array_equal(sc, sr)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d/%d OK" % (rank, world))


if __name__ == "__main__":
    main()
