"""Host-side mirror of the reference's descriptor matcher and kNN over the C ABI.

    siftmatch(L1, L2, thresh=1.5)          <-> sift/siftmatch.c:139-250 (MEX gateway) / :83-132 (compare_<class>)
    kNearestNeighbors(data, query, k)      <-> kNearestNeighbors.m:1-39

Argument checks and messages follow the MEX gateway (siftmatch.c:154-190).
"""
import ctypes as C

import numpy as np

from ._lib import Pre3Error, check, dptr, lib

_CLS = {np.dtype(np.float64): (0, "pre3_siftmatch_f64"), np.dtype(np.float32): (1, "pre3_siftmatch_f32"),
        np.dtype(np.uint8): (2, "pre3_siftmatch_u8"), np.dtype(np.int8): (3, "pre3_siftmatch_i8")}


def _check_args(L1, L2):
    L1, L2 = np.asarray(L1), np.asarray(L2)
    if L1.ndim > 2 or L2.ndim > 2 or not (np.issubdtype(L1.dtype, np.number) and np.issubdtype(L2.dtype, np.number)):
        raise Pre3Error(-1, "L1 and L2 must be two dimensional numeric arrays")
    L1, L2 = np.atleast_2d(L1), np.atleast_2d(L2)
    if L1.shape[0] != L2.shape[0]:
        raise Pre3Error(-1, "L1 and L2 must have the same number of rows")
    if L1.dtype != L2.dtype:
        raise Pre3Error(-1, "L1 and L2 must be of the same class")
    if L1.dtype not in _CLS:
        raise Pre3Error(-1, "Unsupported numeric class")
    return L1, L2


def siftmatch(L1, L2, thresh=1.5, device=0, return_scores=False):
    """MATCHES = siftmatch(L1, L2[, THRESH]); [MATCHES, D] = ... with return_scores=True.

    L1: ND x K1, L2: ND x K2 (one descriptor per column, any of double/single/int8/uint8).
    MATCHES: 2 x M float64, 1-based (k1; k2) in increasing k1; D: 1 x M best squared distances."""
    L1, L2 = _check_args(L1, L2)
    if not np.isscalar(thresh) or isinstance(thresh, complex):
        raise Pre3Error(-1, "THRESH should be a real scalar")
    ND, K1 = L1.shape
    K2 = L2.shape[1]
    a, b = np.asfortranarray(L1), np.asfortranarray(L2)
    pairs = np.zeros(2 * max(K1, 1))
    score = np.zeros(max(K1, 1))
    M = C.c_int(0)
    fn = getattr(lib, _CLS[L1.dtype][1])
    check(fn(int(device), ND, K1, a.ctypes.data_as(C.c_void_p), K2, b.ctypes.data_as(C.c_void_p), C.c_double(float(thresh)),
             dptr(pairs), dptr(score), C.byref(M)))
    matches = pairs[:2 * M.value].reshape(M.value, 2).T.copy()
    return (matches, score[:M.value].copy()) if return_scores else matches


def siftmatch_partial(L1, L2_local, k2_offset, device=0):
    """(best, second, arg) of every L1 column against a database shard (global index = local + k2_offset)."""
    L1, L2 = _check_args(L1, L2_local)
    ND, K1 = L1.shape
    a, b = np.asfortranarray(L1), np.asfortranarray(L2)
    best, second, arg = np.zeros(K1), np.zeros(K1), np.zeros(K1, np.int32)
    check(lib.pre3_siftmatch_partial(int(device), _CLS[L1.dtype][0], ND, K1, a.ctypes.data_as(C.c_void_p), L2.shape[1],
                                     b.ctypes.data_as(C.c_void_p), int(k2_offset), dptr(best), dptr(second), dptr(arg)))
    return best, second, arg


def siftmatch_merge(dtype, best, second, arg, thresh=1.5, return_scores=False):
    """Merge G shards' partials (arrays G x K1) and apply the ratio test (host-side, O(G*K1))."""
    best, second = np.ascontiguousarray(best, np.float64), np.ascontiguousarray(second, np.float64)
    arg = np.ascontiguousarray(arg, np.int32)
    G, K1 = best.shape
    pairs = np.zeros(2 * max(K1, 1))
    score = np.zeros(max(K1, 1))
    M = C.c_int(0)
    check(lib.pre3_siftmatch_merge(_CLS[np.dtype(dtype)][0], G, K1, dptr(best), dptr(second), dptr(arg), C.c_double(float(thresh)),
                                   dptr(pairs), dptr(score), C.byref(M)))
    matches = pairs[:2 * M.value].reshape(M.value, 2).T.copy()
    return (matches, score[:M.value].copy()) if return_scores else matches


def kNearestNeighbors(dataMatrix, queryMatrix, k, device=0):
    """[neighborIds, neighborDistances] = kNearestNeighbors(dataMatrix, queryMatrix, k)  (1-based ids)."""
    data = np.asfortranarray(np.atleast_2d(np.asarray(dataMatrix, dtype=np.float64)))
    query = np.asfortranarray(np.atleast_2d(np.asarray(queryMatrix, dtype=np.float64)))
    N, D = data.shape
    M = query.shape[0]
    if query.shape[1] != D:
        raise Pre3Error(-1, "kNearestNeighbors: data and query must have the same number of columns")
    ids = np.zeros((M, k), order="F")
    dist = np.zeros((M, k), order="F")
    check(lib.pre3_knn_f64(int(device), D, N, data.ctypes.data_as(C.c_void_p), M, query.ctypes.data_as(C.c_void_p), int(k),
                           ids.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
    return np.ascontiguousarray(ids), np.ascontiguousarray(dist)


class MatchShard:
    """Device-resident database shard of the sharded matcher (pre3_match_shard_*): uint8 / double / single descriptors, L1 (128 x K1) replicated, L2_local
    (128 x K2_local) = this rank's columns [k2_offset, ...) of the database.  run() leaves the per-query partials on the device and returns
    (device pointer, number of doubles) for the caller's all-gather; merge() takes the DEVICE address of the gathered double[G][3][K1]."""

    def __init__(self, L1, L2_local, k2_offset, device=0):
        L1, L2 = np.asarray(L1), np.asarray(L2_local)
        cls = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.uint8): 2}.get(L1.dtype)
        if cls is None or L2.dtype != L1.dtype:
            raise Pre3Error(-1, "MatchShard: uint8 (BASELINE.json configs[3]), double (what matching_sift_based.m passes) or single descriptors, both of one class")
        a, b = np.asfortranarray(L1), np.asfortranarray(L2)          # one descriptor per column, contiguous (MATLAB's layout)
        self.K1, self.ND = int(L1.shape[1]), int(L1.shape[0])
        self._h = C.c_void_p()
        check(lib.pre3_match_shard_create_cls(C.byref(self._h), int(device), cls, self.ND, self.K1, a.ctypes.data_as(C.c_void_p), int(L2.shape[1]),
                                         b.ctypes.data_as(C.c_void_p) if L2.shape[1] else None, int(k2_offset)))

    def run(self):
        p, n = C.c_void_p(), C.c_int(0)
        check(lib.pre3_match_shard_run(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def merge(self, G, gathered_ptr, thresh=1.5, return_scores=False):
        pairs, sc, M = np.zeros((self.K1, 2)), np.zeros(self.K1), C.c_int(0)
        check(lib.pre3_match_shard_merge(self._h, int(G), C.c_void_p(int(gathered_ptr)), C.c_double(float(thresh)), dptr(pairs), dptr(sc), C.byref(M)))
        m = pairs[:M.value].T.copy()
        return (m, sc[:M.value].copy()) if return_scores else m

    def set_comm(self, comm):
        check(lib.pre3_match_shard_set_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm

    def match(self, thresh=1.5, return_scores=False):
        """pre3_match_shard_match: distance kernel on the slice, ncclAllGather of the partials, merge + ratio test + compaction on the
        shard's own stream; the merge kernel writes the list to pinned host memory -- one wait, no copy call"""
        if getattr(self, "_out", None) is None:                  # the wrapper's time is GPU idle time: outputs allocated once
            self._out = (np.zeros((self.K1, 2)), np.zeros(self.K1), C.c_int(0))
            self._out_p = (dptr(self._out[0]), dptr(self._out[1]), C.byref(self._out[2]))
        pairs, sc, M = self._out
        check(lib.pre3_match_shard_match(self._h, C.c_double(float(thresh)), self._out_p[0], self._out_p[1], self._out_p[2]))
        m = pairs[:M.value].T.copy()
        return (m, sc[:M.value].copy()) if return_scores else m

    def test_stall(self, code):
        """test hook: 0 parks a kernel on the shard's stream until test_stall(1); 2: the next match's distance kernels 'fail'"""
        check(lib.pre3_match_shard_test_stall(self._h, int(code)))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:          # (lib is None while the interpreter shuts down)
            lib.pre3_match_shard_destroy(self._h)
            self._h = None

    __del__ = close
