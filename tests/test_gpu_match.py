"""GPU parity of the matcher (sift/siftmatch.c) and kNearestNeighbors.m through the C ABI: the reference's
own known answers on box.sift, the oracle on seeded sets, edge cases, and shard-merge invariance."""
import json
import os

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu


def test_box_sift_known_answers(pre3):
    box = np.load(os.path.join(GOLDEN, "sift_box.npz"))["descriptors"]
    kat = json.load(open(os.path.join(GOLDEN, "siftmatch_kat.json")))
    for c in kat["cases"]:
        for dt in (np.uint8, np.float64, np.float32):
            L1 = box[c["L1"][0]:c["L1"][1]].T.astype(dt)
            L2 = box[c["L2"][0]:c["L2"][1]].T.astype(dt)
            mt, sc = pre3.siftmatch(L1, L2, c["thresh"], return_scores=True)
            M = mt.shape[1]
            assert M == c["M"]
            assert int(sum((i + 1) * (1000 * int(mt[0, i]) + int(mt[1, i])) for i in range(M))) == c["checksum"]
            assert sc.sum() == c["sum_best_d2"]
            assert list(mt[:, 0]) == c["first"] and list(mt[:, -1]) == c["last"]


@pytest.mark.parametrize("dt", [np.uint8, np.int8, np.float32, np.float64])
def test_seeded_sets_bit_exact(pre3, orc, dt):
    rng = np.random.default_rng(123)
    K1, K2 = 300, 517
    if dt in (np.uint8, np.int8):
        lo, hi = (0, 256) if dt == np.uint8 else (-128, 128)
        L1 = rng.integers(lo, hi, (128, K1)).astype(dt)
        L2 = rng.integers(lo, hi, (128, K2)).astype(dt)
        L2[:, 100:250] = np.clip(L1[:, :150].astype(int) + rng.integers(-3, 4, (128, 150)), lo, hi - 1).astype(dt)
        L2[:, 400] = L2[:, 100]                      # duplicate: ties keep the first index, ratio test fails
    else:
        L1 = rng.random((128, K1)).astype(dt)
        L1 /= np.linalg.norm(L1, axis=0, keepdims=True)
        L2 = rng.random((128, K2)).astype(dt)
        L2 /= np.linalg.norm(L2, axis=0, keepdims=True)
        L2[:, 100:250] = L1[:, :150] + (rng.random((128, 150)) * 0.02).astype(dt)
        L2[:, 400] = L2[:, 100]
    for th in (1.5, 1.0, 3.0):
        m, d = pre3.siftmatch(L1, L2, th, return_scores=True)
        mr, dr = orc.siftmatch(L1, L2, th)
        assert np.array_equal(m, mr)
        assert np.array_equal(d, dr)                 # bit-exact scores in every class


def test_edge_shapes(pre3, orc):
    z = pre3.siftmatch(np.zeros((128, 3)), np.zeros((128, 0)))
    assert z.shape == (2, 0)
    z = pre3.siftmatch(np.zeros((128, 0)), np.zeros((128, 3)))
    assert z.shape == (2, 0)
    m = pre3.siftmatch(np.ones((4, 2)), np.ones((4, 1)))
    assert m.tolist() == [[1, 2], [1, 1]]
    # non-multiple-of-tile sizes and ND != 128 on the int8 MFMA path
    rng = np.random.default_rng(9)
    for ND, K1, K2 in ((128, 1, 1), (128, 129, 257), (36, 50, 70), (130, 33, 65)):
        L1 = rng.integers(0, 256, (ND, K1)).astype(np.uint8)
        L2 = rng.integers(0, 256, (ND, K2)).astype(np.uint8)
        m, d = pre3.siftmatch(L1, L2, 1.1, return_scores=True)
        mr, dr = orc.siftmatch(L1, L2, 1.1)
        assert np.array_equal(m, mr) and np.array_equal(d, dr)
    # the tiled float kernels (>= 64*64*16 pairs): ragged tiles, ND not a multiple of the 32-bin chunk, exact ties
    for dt, ND, K1, K2 in ((np.float32, 36, 257, 300), (np.float64, 130, 300, 257), (np.float32, 128, 64, 1024), (np.float64, 5, 1025, 65)):
        L1 = rng.random((ND, K1)).astype(dt)
        L2 = rng.random((ND, K2)).astype(dt)
        L2[:, 7] = L1[:, 3]; L2[:, K2 - 1] = L1[:, 3]          # two exact zero-distance candidates: first index wins, ratio test fails
        m, d = pre3.siftmatch(L1, L2, 1.2, return_scores=True)
        mr, dr = orc.siftmatch(L1, L2, 1.2)
        assert np.array_equal(m, mr) and np.array_equal(d, dr), (dt, ND, K1, K2)
    # maximum distance for uint8: 128 * 255^2 stays inside int32
    L1 = np.zeros((128, 2), np.uint8)
    L2 = np.full((128, 2), 255, np.uint8)
    m, d = pre3.siftmatch(L1, L2, 1.0, return_scores=True)
    assert d.tolist() == [128 * 255 * 255] * 2


@pytest.mark.parametrize("dt", [np.uint8, np.float64])
def test_shard_merge_invariance(pre3, orc, dt):
    """database sharded over G 'GPUs' -> per-shard partials -> merge == unsharded (C2 of DESIGN.md)"""
    rng = np.random.default_rng(77)
    K1, K2 = 200, 640
    L1 = rng.integers(0, 200, (128, K1)).astype(dt)
    L2 = rng.integers(0, 200, (128, K2)).astype(dt)
    L2[:, 300:400] = L1[:, :100]
    L2[:, 639] = L1[:, 0]                  # exact tie across shards: lowest global index must win
    ref_m, ref_d = orc.siftmatch(L1, L2, 1.5)
    for G in (1, 2, 4, 8):
        b, s, a = [], [], []
        for g in range(G):
            lo, hi = g * K2 // G, (g + 1) * K2 // G
            pb, ps, pa = pre3.siftmatch_partial(L1, L2[:, lo:hi], lo)
            b.append(pb); s.append(ps); a.append(pa)
        m, d = pre3.siftmatch_merge(dt, np.stack(b), np.stack(s), np.stack(a), 1.5, return_scores=True)
        assert np.array_equal(m, ref_m) and np.array_equal(d, ref_d)


def test_full_size_config4_properties(pre3):
    """4096 x 4096 x 128 uint8: permuted copy + noise -> every surviving match must be the planted one,
    and a checksum of the match list must be identical for the MFMA path and the exact fp32-class path."""
    rng = np.random.default_rng(5000)
    K = 4096
    base = np.minimum(np.round(np.abs(rng.standard_normal((128, K))) * 40), 255).astype(np.uint8)
    perm = rng.permutation(K)
    L2 = base[:, perm].astype(int) + rng.integers(-2, 3, (128, K))
    L2 = np.clip(L2, 0, 255).astype(np.uint8)
    m = pre3.siftmatch(base, L2, 1.5)
    inv = np.empty(K, int)
    inv[perm] = np.arange(K)
    assert m.shape[1] > 0.95 * K
    assert np.array_equal(m[1].astype(int) - 1, inv[m[0].astype(int) - 1])
    mf = pre3.siftmatch(base.astype(np.float32), L2.astype(np.float32), 1.5)
    assert np.array_equal(m, mf)


def test_knn(pre3, orc):
    kat = json.load(open(os.path.join(GOLDEN, "siftmatch_kat.json")))["knn_docstring_example"]
    ids, d = pre3.kNearestNeighbors(kat["data"], kat["query"], kat["k"])
    assert ids.tolist() == kat["neighbors"] and np.allclose(np.round(d, 4), kat["distances"])
    rng = np.random.default_rng(3)
    data = np.round(rng.random((500, 2)) * 50)           # 2-D pixel coordinates with many exact ties (inittialize_depth.m:13 use)
    query = np.round(rng.random((40, 2)) * 50)
    for k in (1, 4, 17):
        ids, d = pre3.kNearestNeighbors(data, query, k)
        ir, dr = orc.knn(data, query, k)
        assert np.array_equal(ids, ir) and np.array_equal(d, dr)
    data = rng.random((70, 128))
    ids, d = pre3.kNearestNeighbors(data, data[:5], 3)
    ir, dr = orc.knn(data, data[:5], 3)
    assert np.array_equal(ids, ir) and np.array_equal(d, dr)
