"""The float / double classes of siftmatch on the matrix cores (pre3_match.hip: k_rank_pack, k_match_rank): a bf16 distance GEMM ranks the
database, the candidates inside the guard band are re-evaluated in the reference's own arithmetic (sift/siftmatch.c:97-116).  Everything
here must be BIT-identical to the oracle / to the exact VALU kernels, including scores, ties and the second-best value."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Form:
    """PRE3_MATCH_FLOAT_FORM (or another switch the library reads per call) for the duration of a block (0: exact kernels only, 2: never the int8 route)"""
    def __init__(self, v, name="PRE3_MATCH_FLOAT_FORM"):
        self.v, self.name = v, name

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = str(self.v)

    def __exit__(self, *a):
        if self.old is None:
            del os.environ[self.name]
        else:
            os.environ[self.name] = self.old


def _route(pre3, L1, L2):
    """[route, queries scanned in full, candidates re-evaluated] of one resident run (pre3_match_bench_*), or None when the data is refused"""
    lib = pre3._lib.lib
    cls = {np.dtype(np.float64): 0, np.dtype(np.float32): 1}[L1.dtype]
    a, b = np.asfortranarray(L1), np.asfortranarray(L2)
    h = lib.pre3_match_bench_create_cls(0, cls, a.shape[0], a.shape[1], a.ctypes.data_as(C.c_void_p), b.shape[1], b.ctypes.data_as(C.c_void_p))
    if not h:
        return None
    ms = C.c_double(0)
    assert lib.pre3_match_bench_run(C.c_void_p(h), 1, C.byref(ms)) == 0
    info = (C.c_int32 * 3)()
    assert lib.pre3_match_bench_info(C.c_void_p(h), info) == 0
    K1 = a.shape[1]
    best, second, arg = np.zeros(K1), np.zeros(K1), np.zeros(K1, np.int32)
    assert lib.pre3_match_bench_fetch(C.c_void_p(h), best.ctypes.data_as(C.c_void_p), second.ctypes.data_as(C.c_void_p), arg.ctypes.data_as(C.c_void_p)) == 0
    lib.pre3_match_bench_destroy(C.c_void_p(h))
    return list(info), (best, second, arg)


def _exact_partial(pre3, L1, L2):
    with _Form(0):
        return pre3.siftmatch_partial(L1, L2, 0)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_routes(pre3, dt):
    rng = np.random.default_rng(1)
    L1, L2 = rng.random((128, 256)).astype(dt), rng.random((128, 512)).astype(dt)
    info, got = _route(pre3, L1, L2)
    assert info[0] == 2 and info[1] == 0 and 2 * 256 <= info[2] <= 8 * 256          # bf16 rank; a handful of candidates per query
    ref = _exact_partial(pre3, L1, L2)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref))
    Li1, Li2 = np.floor(L1 * 256).astype(dt), np.floor(L2 * 256).astype(dt)        # integer-valued doubles (Lowe-format files such as sift/data/box.sift)
    info, got = _route(pre3, Li1, Li2)
    assert info[0] == 1                                                              # integers in [0, 255]: the int8 kernel is exact
    ref = _exact_partial(pre3, Li1, Li2)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref))
    bad = L2.copy(); bad[5, 7] = np.nan
    assert _route(pre3, L1, bad) is None                                             # outside the bounds: the stateless entry takes the exact kernels
    big = L2.copy(); big[0, 0] = 1e30
    assert _route(pre3, L1, big) is None


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_near_ties_and_duplicates_bit_exact(pre3, orc, dt):
    """perturbations far below the GEMM's resolution, exact duplicates (first index wins, second == best), more duplicates than the
    candidate list holds (the wave scans the whole database), against the oracle with scores"""
    rng = np.random.default_rng(2)
    K1, K2 = 192, 1100
    eps = np.finfo(dt).eps
    L1 = (rng.random((128, K1)) * 3).astype(dt)
    L2 = (rng.random((128, K2)) * 3).astype(dt)
    L2[:, 10:74] = L1[:, :64] * (1 + 4 * eps * rng.integers(-3, 4, (128, 64))).astype(dt)       # a few ulps away from the query
    L2[:, 200:264] = L2[:, 10:74]                                                              # ... twice: exact duplicates of the best
    L2[:, 300:364] = L1[:, :64] * (1 + 1e-5 * rng.standard_normal((128, 64))).astype(dt)        # inside the guard band, ordered only exactly
    L2[:, 400:500] = L1[:, 100][:, None]                                                       # 100 copies of query 100 (> 64 candidates)
    L2[:, 1099] = L1[:, 191]
    for th in (1.5, 1.0):
        m, d = pre3.siftmatch(L1, L2, th, return_scores=True)
        mr, dr = orc.siftmatch(L1, L2, th)
        assert np.array_equal(m, mr) and np.array_equal(d, dr)
    info, got = _route(pre3, L1, L2)
    assert info[0] == 2 and info[1] >= 1                                                        # query 100 overflowed its list
    ref = _exact_partial(pre3, L1, L2)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_scales_and_forced_rank(pre3, orc, dt):
    rng = np.random.default_rng(3)
    base1, base2 = rng.standard_normal((128, 130)), rng.standard_normal((128, 700))
    base2[:, 50:100] = base1[:, :50] + 0.01 * rng.standard_normal((128, 50))
    for scale in (1.0, 1e9, 1e-9, 3e17 if dt == np.float64 else 1e15):                          # squares stay inside f32 for the GEMM
        L1, L2 = (base1 * scale).astype(dt), (base2 * scale).astype(dt)
        m, d = pre3.siftmatch(L1, L2, 1.3, return_scores=True)
        mr, dr = orc.siftmatch(L1, L2, 1.3)
        assert np.array_equal(m, mr) and np.array_equal(d, dr), scale
    # mixed magnitudes inside one descriptor, negative values, a zero column
    L1 = (base1 * np.logspace(-3, 3, 128)[:, None]).astype(dt)
    L2 = (base2 * np.logspace(-3, 3, 128)[:, None]).astype(dt)
    L2[:, 5] = 0
    m, d = pre3.siftmatch(L1, L2, 1.1, return_scores=True)
    mr, dr = orc.siftmatch(L1, L2, 1.1)
    assert np.array_equal(m, mr) and np.array_equal(d, dr)
    # integer-valued descriptors forced through the bf16 rank (integer distances: dense ties)
    Li1 = rng.integers(0, 256, (128, 140)).astype(dt)
    Li2 = rng.integers(0, 256, (128, 600)).astype(dt)
    Li2[:, 100:200] = np.clip(Li1[:, :100] + rng.integers(-1, 2, (128, 100)), 0, 255)
    Li2[:, 300] = Li2[:, 100]
    with _Form(2):
        m, d = pre3.siftmatch(Li1, Li2, 1.5, return_scores=True)
    mr, dr = orc.siftmatch(Li1, Li2, 1.5)
    assert np.array_equal(m, mr) and np.array_equal(d, dr)
    m1, d1 = pre3.siftmatch(Li1, Li2, 1.5, return_scores=True)                                   # the int8 route
    assert np.array_equal(m1, mr) and np.array_equal(d1, dr)


def test_the_reference_s_own_descriptors_through_the_ranked_route(pre3, orc, sr4000):
    """What matching_sift_based.m:104-118 really hands siftmatch: siftdescriptor.c:125-141's UNIT-NORM real-valued descriptors as doubles (the
    185 landmark descriptors of the SR4000 snapshot: max bin 0.48, norm 1) against a scan made of their perturbed, permuted copies plus
    clutter.  The guard band E (n_q + n_b) = 3.6e-4 is then small against match distances of 0.01-0.3 -- few candidates per query -- and the
    result must be the oracle's bit for bit, scores included.  PRE3_MATCH_FLOAT_FORM=2 keeps the problem (185 x 700, below the size cut-off
    of the matrix-core route) on that route."""
    rng = np.random.default_rng(11)
    D = np.ascontiguousarray(sr4000["descriptor"].reshape(-1, 128).T)              # 128 x 185, column = descriptor
    assert D.shape == (128, 185) and np.allclose(np.linalg.norm(D, axis=0), 1.0, atol=1e-6) and D.max() < 0.6

    def sift_like(X):                                                               # siftdescriptor.c:125-141: normalise, clip at 0.2, renormalise
        X = np.abs(X)
        X = X / np.linalg.norm(X, axis=0)
        X = np.minimum(X, 0.2)
        return X / np.linalg.norm(X, axis=0)

    perm = rng.permutation(185)
    scan = np.concatenate([sift_like(D[:, perm] + 0.02 * np.abs(rng.standard_normal((128, 185)))),      # the landmarks seen again, perturbed
                           sift_like(rng.standard_normal((128, 515)))], axis=1)                        # clutter keypoints
    scan[:, 300] = scan[:, 7]                                                       # an exact duplicate in the scan: a tie for the best
    for dt in (np.float64, np.float32):
        L1, L2 = D.astype(dt), scan.astype(dt)
        mr, dr = orc.siftmatch(L1, L2, 1.5)
        with _Form(2):
            m, d = pre3.siftmatch(L1, L2, 1.5, return_scores=True)
            info, got = _route(pre3, L1, L2)
        assert np.array_equal(m, mr) and np.array_equal(d, dr), dt
        assert mr.shape[1] > 100                                                    # most landmarks find their perturbed copy
        assert info[0] == 2 and info[1] == 0 and info[2] <= 8 * 185, info           # ranked route, nobody scanned in full, a few candidates each
        ref = _exact_partial(pre3, L1, L2)
        assert all(np.array_equal(g, r) for g, r in zip(got, ref))


def test_full_size_double_class(pre3):
    """BASELINE configs[3] shape in the class the reference calls (double, matching_sift_based.m:104-118): 4096 x 4096 x 128 real-valued
    descriptors; (best, second, arg) of the ranked path == the exact kernels for every query, and the planted matches are found"""
    rng = np.random.default_rng(4)
    K = 4096
    L1 = np.abs(rng.standard_normal((128, K))) * 40
    perm = rng.permutation(K)
    L2 = L1[:, perm] + rng.uniform(-2, 2, (128, K))
    info, got = _route(pre3, L1, L2)
    assert info[0] == 2 and info[1] == 0
    ref = _exact_partial(pre3, L1, L2)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref))
    inv = np.empty(K, int); inv[perm] = np.arange(K)
    assert np.array_equal(got[2], inv)
    L1f, L2f = L1.astype(np.float32), L2.astype(np.float32)
    info, got = _route(pre3, L1f, L2f)
    ref = _exact_partial(pre3, L1f, L2f)
    assert info[0] == 2 and all(np.array_equal(g, r) for g, r in zip(got, ref))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_one_launch_and_tiled_forms_agree(pre3, orc, dt):
    """PRE3_MATCH_RANK_FORM: the tiled form (three launches, default) and the one-launch form rank with differently packed planes and
    different reduction trees -- the exact re-evaluation makes both land on the oracle's bits"""
    rng = np.random.default_rng(6)
    L1 = (rng.standard_normal((128, 333)) * 7).astype(dt)
    L2 = (rng.standard_normal((128, 901)) * 7).astype(dt)
    L2[:, 100:300] = L1[:, :200] + (0.05 * rng.standard_normal((128, 200))).astype(dt)
    L2[:, 900] = L2[:, 100]
    mr, dr = orc.siftmatch(L1, L2, 1.4)
    for form in (1, 0, 2):                     # (2, round 6: 128 queries per workgroup, the database slice staged through LDS -- measured slower, kept as an option)
        with _Form(form, "PRE3_MATCH_RANK_FORM"):
            m, d = pre3.siftmatch(L1, L2, 1.4, return_scores=True)
        assert np.array_equal(m, mr) and np.array_equal(d, dr), form


def test_lds_staged_rank_form_at_full_size_is_bit_identical(pre3):
    """4096 x 4096 x 128 unit-norm doubles (BASELINE configs[3]'s float class): PRE3_MATCH_RANK_FORM=2 against the default form"""
    rng = np.random.default_rng(11)
    L1 = rng.random((128, 4096)); L1 /= np.linalg.norm(L1, axis=0)
    L2 = rng.random((128, 4096)); L2 /= np.linalg.norm(L2, axis=0)
    L2[:, 7] = L1[:, 1234]
    res = []
    for form in (1, 2):
        with _Form(form, "PRE3_MATCH_RANK_FORM"):
            res.append(pre3.siftmatch(L1, L2, 1.5, return_scores=True))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
