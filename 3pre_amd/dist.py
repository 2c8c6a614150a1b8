"""Multi-GPU sharding of the two stages of the path that shard (DESIGN.md "multi-GPU"); one process per GPU,
torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

  * RANSAC (C1): every rank holds the same filter state, scores hypotheses [lo, hi) of the shared draw table
    on its GPU, and the int32 supports + inlier bitmasks (zero outside a rank's slice) are summed with ONE
    all-reduce each (<= 8 KB + n_draw*ceil(m/32)*4 B); every rank then replays the reference's termination
    rule on the reduced buffers and reaches the same winner.  The EKF update is not sharded.
  * matcher (C2): the database L2 is split by columns; every rank computes (best, second, arg) of all
    queries against its slice, one all-gather of 3*K1 numbers per rank, then the order-independent merge +
    ratio test (pre3_siftmatch_merge) -- bit-identical to the unsharded match for any number of ranks.

The compute callables are injected, so that the CPU test-suite can drive the SAME sharding / collective /
merge code with the oracle standing in for the GPU kernels (tests/test_dist_cpu.py).
"""
import numpy as np
import torch
import torch.distributed as dist


class _DevView:
    """device memory owned by libpre3 (a context's support / mask buffer, a matcher shard's partials), for torch.as_tensor (no copy)"""
    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def dev_tensor(ptr, n, typestr="<i4", device=None):
    """zero-copy torch view of `n` elements of libpre3 device memory at `ptr` (int32 by default, "<f8" for doubles)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    return torch.as_tensor(_DevView(ptr, n, typestr), device=dev)


def shard_range(n, rank, world):
    """contiguous slice [lo, hi) of n items for `rank` (sizes differ by at most one)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# ---------------------------------------------------------------------------------------------------
# C1: sharded RANSAC scoring
# ---------------------------------------------------------------------------------------------------
_ZERO_COPY = {}         # (backend, world, n_draw, words) -> bool: the path every rank agreed on for rounds of THAT shape (decided collectively)


def _agree_zero_copy(ok_here, n_draw, words):
    """All ranks must issue the same collectives: the zero-copy in-place all-reduce is used only if it works on EVERY rank (one MIN
    all-reduce per round shape; a rank-local choice could leave ranks in collectives of different count and size).  The decision is
    cached per (backend, world, n_draw, words): whether the view can be built depends on the buffer layout of that shape (the same on
    every rank -- state and draws are replicated), so a cached True cannot meet a local failure that the other ranks do not share; should
    it happen anyway, the cached decision is dropped and agreed again by every rank's next call, and this call raises on this rank alone --
    after the agreement, not inside a collective the others never entered."""
    key = (dist.get_backend(), dist.get_world_size(), int(n_draw), int(words))
    if key in _ZERO_COPY and _ZERO_COPY[key] and not ok_here:
        del _ZERO_COPY[key]
    if key not in _ZERO_COPY:
        t = torch.tensor([1 if ok_here else 0], dtype=torch.int32, device="cuda" if key[0] == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        _ZERO_COPY[key] = bool(t.item())
    return _ZERO_COPY[key]


def ransac_sharded(f, hyp, threshold, early_exit=True, device=None, timing=None, fetch=True):
    """f: EkfFilter with projection + measurements installed (identical on every rank).
    Scores this rank's slice on the GPU, all-reduces supports and masks, selects.  Returns the same dict as
    EkfFilter.ransac_hypotheses on every rank.  timing: optional dict that receives the round's split (compute / collective / select, s).
    fetch=False: only the statistics are returned (no D2H of the supports and the winner's mask: they stay on the device for the update)."""
    import time
    rank, world = _world()
    hyp = np.ascontiguousarray(hyp, np.int32)
    n_draw, k = hyp.shape
    lo, hi = shard_range(n_draw, rank, world)
    t0 = time.perf_counter()
    sup_ptr, msk_ptr, words = f.ransac_score_shard(hyp, threshold, lo, hi)          # synchronous on return
    t1 = time.perf_counter()
    if world > 1:
        backend = dist.get_backend()
        if backend == "nccl":
            # all-reduce in place on the context's own support / mask buffers (zero-copy view): no export / import copies, no extra syncs
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            gap = (msk_ptr - sup_ptr) // 4                  # the masks follow the supports in one allocation (pre3_ransac_score)
            both = None
            if n_draw <= gap <= n_draw + 4:
                try:
                    both = dev_tensor(sup_ptr, gap + n_draw * max(words, 1), "<i4", dev)
                except Exception:                           # a torch build that cannot import __cuda_array_interface__ objects
                    both = None
            if _agree_zero_copy(both is not None, n_draw, words):
                dist.all_reduce(both, op=dist.ReduceOp.SUM)  # ONE collective, in place; slices are disjoint: integer sum == bitwise or
                torch.cuda.synchronize()
            else:                                           # stage through copies (every rank takes this branch together)
                sup = torch.empty(n_draw, dtype=torch.int32, device=dev)
                msk = torch.empty(n_draw * max(words, 1), dtype=torch.int32, device=dev)
                f.ransac_export(n_draw, sup.data_ptr(), msk.data_ptr())
                dist.all_reduce(sup, op=dist.ReduceOp.SUM)
                dist.all_reduce(msk, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
                f.ransac_import(n_draw, sup.data_ptr(), msk.data_ptr())
        else:                                               # gloo rehearsal: stage through the host
            sup = torch.empty(n_draw, dtype=torch.int32, device="cuda")
            msk = torch.empty(n_draw * max(words, 1), dtype=torch.int32, device="cuda")
            f.ransac_export(n_draw, sup.data_ptr(), msk.data_ptr())
            sup_h, msk_h = sup.cpu(), msk.cpu()
            dist.all_reduce(sup_h, op=dist.ReduceOp.SUM)
            dist.all_reduce(msk_h, op=dist.ReduceOp.SUM)
            sup.copy_(sup_h); msk.copy_(msk_h)
            torch.cuda.synchronize()
            f.ransac_import(n_draw, sup.data_ptr(), msk.data_ptr())
    t2 = time.perf_counter()
    out = f.ransac_select(n_draw, k, early_exit, fetch=fetch)
    if timing is not None:
        t3 = time.perf_counter()
        timing["compute"] = timing.get("compute", 0.0) + (t1 - t0)
        timing["collective"] = timing.get("collective", 0.0) + (t2 - t1)
        timing["select"] = timing.get("select", 0.0) + (t3 - t2)
    return out


def ransac_sharded_generic(score_slice, select, n_draw, words):
    """The collective skeleton of ransac_sharded with injected compute (CPU tests):
    score_slice(lo, hi) -> (support int32[n_draw] zero outside the slice, masks int32[n_draw*words]);
    select(support, masks) -> result."""
    rank, world = _world()
    lo, hi = shard_range(n_draw, rank, world)
    sup, msk = score_slice(lo, hi)
    sup = torch.from_numpy(np.ascontiguousarray(sup, np.int32))
    msk = torch.from_numpy(np.ascontiguousarray(msk, np.int32))
    if world > 1:
        dist.all_reduce(sup, op=dist.ReduceOp.SUM)
        dist.all_reduce(msk, op=dist.ReduceOp.SUM)
    return select(sup.numpy(), msk.numpy())


# ---------------------------------------------------------------------------------------------------
# C2: sharded matcher
# ---------------------------------------------------------------------------------------------------
def siftmatch_sharded(L1, L2, thresh=1.5, partial=None, merge=None, return_scores=False):
    """L1 (ND x K1) replicated, L2 (ND x K2) known to every rank; rank g matches against columns
    shard_range(K2, g, G).  partial(L1, L2_slice, offset) -> (best, second, arg); merge(dtype, B, S, A, thresh).
    Defaults are the GPU kernels of this package."""
    if partial is None or merge is None:
        from . import matcher
        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0     # one process per GPU: the rank's own device
        partial = partial or (lambda a, b, off: matcher.siftmatch_partial(a, b, off, device=dev))
        merge = merge or (lambda dt, B, S, A, th: matcher.siftmatch_merge(dt, B, S, A, th, return_scores=True))
    rank, world = _world()
    L1, L2 = np.asarray(L1), np.asarray(L2)
    K1 = L1.shape[1]
    lo, hi = shard_range(L2.shape[1], rank, world)
    b, s, a = partial(L1, L2[:, lo:hi], lo)
    pack = torch.from_numpy(np.concatenate([np.asarray(b, np.float64), np.asarray(s, np.float64), np.asarray(a, np.float64)]))
    if world > 1:
        use_cuda = dist.get_backend() == "nccl"
        if use_cuda:
            pack = pack.cuda()
        out = [torch.empty_like(pack) for _ in range(world)]
        dist.all_gather(out, pack)
        allp = torch.stack(out).cpu().numpy()
    else:
        allp = pack.numpy()[None, :]
    B, S, A = allp[:, :K1], allp[:, K1:2 * K1], allp[:, 2 * K1:].astype(np.int32)
    m, d = merge(L1.dtype, B, S, A, thresh)
    return (m, d) if return_scores else m


def siftmatch_sharded_resident(shard, thresh=1.5, return_scores=False):
    """The sharded matcher with everything resident on the GPUs (matcher.MatchShard: queries replicated, this rank's database slice packed
    in HBM): distance kernel on the slice, ONE all-gather of the per-query partials as device tensors (RCCL; gloo stages through the
    host), merge + ratio test + compaction on the device.  Only the match list crosses PCIe.  Identical to the unsharded match."""
    rank, world = _world()
    ptr, n = shard.run()
    part = dev_tensor(ptr, n, "<f8")
    if world > 1:
        if dist.get_backend() == "nccl":
            allp = torch.empty(world * n, dtype=torch.float64, device=part.device)
            dist.all_gather_into_tensor(allp, part)
        else:
            ph = part.cpu()
            outs = [torch.empty_like(ph) for _ in range(world)]
            dist.all_gather(outs, ph)
            allp = torch.cat(outs).to(part.device)
        torch.cuda.synchronize()
    else:
        allp = part
    return shard.merge(world, allp.data_ptr(), thresh, return_scores)
