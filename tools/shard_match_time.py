"""Where a resident sharded match spends its time on one GPU: MatchShard.match() with and without libpre3's communicator (world = 1), against the
torch.distributed form (dist.siftmatch_sharded_resident) and the kernel alone.  usage: python tools/shard_match_time.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd")
mt = importlib.import_module("3pre_amd.matcher")
cm = importlib.import_module("3pre_amd.comm")
pd = importlib.import_module("3pre_amd.dist")
rng = np.random.default_rng(5000)
K = 4096
L1 = np.minimum(np.round(np.abs(rng.standard_normal((K, 128))) * 40), 255).astype(np.uint8)
L2 = np.clip(L1[rng.permutation(K)].astype(int) + rng.integers(-2, 3, (K, 128)), 0, 255).astype(np.uint8)
sh = mt.MatchShard(np.asfortranarray(L1.T), np.asfortranarray(L2.T), 0)
comm = cm.Comm(0, cm.unique_id(), 0, 1)


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e6 * (time.perf_counter() - t0) / reps


print("match(), no communicator      %.1f us" % timeit(lambda: sh.match(1.5)))
sh.set_comm(comm)
print("match(), communicator world 1 %.1f us" % timeit(lambda: sh.match(1.5)))
sh.set_comm(None)
print("run + torch + merge           %.1f us" % timeit(lambda: pd.siftmatch_sharded_resident(sh, 1.5)))
print("matches", sh.match(1.5).shape)
sh.close()
comm.close()
