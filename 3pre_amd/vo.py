"""Host-side mirror of the reference's visual-odometry RANSAC over the C ABI.

    vodometry_dr_ye.m:162-236            -> vo_ransac / vo_ransac_frames (all hypotheses, winner, final fit, statistics)
    ransac_dr_ye.m:28-46                 -> draw_hypotheses (the reference's own rejection rule, any numpy Generator)
    vodometry_dr_ye.m:171                -> vo_rst
    Calculate_V_Omega_RANSAC_dr_ye.m:40-50 -> result["u"] = [T; R2q(R)], the argument of EkfFilter.ekf_prediction

All compute runs in libpre3.so on the GPU; this module marshals numpy arrays and draws random numbers.
"""
import ctypes as C
from math import comb

import numpy as np

from ._lib import check, dptr, f64, i32, lib


class VoResult(C.Structure):
    _fields_ = [("rot", C.c_double * 9), ("trans", C.c_double * 3), ("euler", C.c_double * 3), ("u", C.c_double * 7),
                ("error_mean", C.c_double), ("error_std", C.c_double), ("dist", C.c_double),
                ("sta", C.c_int32), ("n_support", C.c_int32), ("n_iterations", C.c_int32), ("best", C.c_int32)]


def vo_rst(pnum):
    """rst = min(700, nchoosek(pnum, 4))  (vodometry_dr_ye.m:171)"""
    return min(700, comb(int(pnum), 4))


def draw_hypotheses(match, n_hyp, rng):
    """ransac_dr_ye.m:28-46, n_hyp times: positions num_rs(1:4) = round((pnum-1)*rand+1), redrawn while they repeat or share
    a keypoint -- including the reference's mixed-row comparisons in ind_dup3 (match(1,.) against match(2,.)).
    match: (2, pnum) keypoint numbers.  Returns 0-based positions (n_hyp, 4)."""
    m = np.asarray(match)
    pnum = m.shape[1]
    out = np.zeros((n_hyp, 4), np.int32)

    def rnd():
        return int(np.floor((pnum - 1) * rng.random() + 1 + 0.5)) - 1          # MATLAB round(), then 0-based

    def dup1(r):
        return m[0, r[0]] == m[0, r[1]] or m[1, r[0]] == m[1, r[1]]

    def dup2(r):
        return m[0, r[0]] == m[0, r[2]] or m[0, r[1]] == m[0, r[2]] or m[1, r[0]] == m[1, r[2]] or m[1, r[1]] == m[1, r[2]]

    def dup3(r):
        return (m[0, r[0]] == m[0, r[3]] or m[0, r[1]] == m[1, r[3]] or m[0, r[2]] == m[0, r[3]] or m[1, r[0]] == m[0, r[3]]
                or m[1, r[1]] == m[1, r[3]] or m[1, r[2]] == m[1, r[3]])

    for h in range(n_hyp):
        r = [rnd() for _ in range(4)]
        while r[1] == r[0] or dup1(r):
            r[1] = rnd()
        while r[2] == r[0] or r[2] == r[1] or dup2(r):
            r[2] = rnd()
        while r[3] == r[0] or r[3] == r[1] or r[3] == r[2] or dup3(r):
            r[3] = rnd()
        out[h] = r
    return out


def _result(res, cnum, state, inl):
    return dict(rot=np.array(res.rot).reshape(3, 3), trans=np.array(res.trans), euler=np.array(res.euler), u=np.array(res.u),
                error_mean=res.error_mean, error_std=res.error_std, dist=res.dist, sta=int(res.sta), n_support=int(res.n_support),
                n_iterations=int(res.n_iterations), best=int(res.best), cnum=cnum, state=state, inliers=inl)


def vo_ransac(pset1, pset2, draws, device=0):
    """pset1, pset2: (3, pnum) matched points of frame 1 / 2 (ransac_dr_ye.m:13-19); draws: (n_hyp, 4) 0-based."""
    p1, p2 = np.ascontiguousarray(f64(pset1).T), np.ascontiguousarray(f64(pset2).T)
    assert p1.shape == p2.shape and p1.shape[1] == 3
    draws = i32(draws).reshape(-1, 4)
    pnum, n_hyp = p1.shape[0], draws.shape[0]
    cnum, state, inl = np.zeros(n_hyp, np.int32), np.zeros(n_hyp, np.int32), np.zeros(max(pnum, 1), np.int32)
    res = VoResult()
    check(lib.pre3_vo_ransac(int(device), pnum, dptr(p1), dptr(p2), n_hyp, dptr(draws), dptr(cnum), dptr(state), dptr(inl), C.byref(res)))
    return _result(res, cnum, state, inl[:pnum])


def vo_ransac_frames(frm1, frm2, match, x1, y1, z1, x2, y2, z2, draws, device=0):
    """ransac_dr_ye's own argument list (frm: (>=2, K) SIFT frames, match: (2, pnum) 1-based, x/y/z: (rows, cols))."""
    imgs = [np.asfortranarray(f64(a)) for a in (x1, y1, z1, x2, y2, z2)]
    rows, cols = imgs[0].shape
    assert all(a.shape == (rows, cols) for a in imgs)
    f1, f2 = np.asfortranarray(f64(frm1)), np.asfortranarray(f64(frm2))
    assert f1.shape[0] == f2.shape[0] >= 2
    mt = np.asfortranarray(f64(match))
    pnum = mt.shape[1]
    draws = i32(draws).reshape(-1, 4)
    n_hyp = draws.shape[0]
    p1, p2 = np.zeros((max(pnum, 1), 3)), np.zeros((max(pnum, 1), 3))
    cnum, state, inl = np.zeros(n_hyp, np.int32), np.zeros(n_hyp, np.int32), np.zeros(max(pnum, 1), np.int32)
    res = VoResult()
    P = [a.ctypes.data_as(C.c_void_p) for a in imgs]
    check(lib.pre3_vo_ransac_frames(int(device), rows, cols, *P, f1.shape[0], f1.shape[1], f1.ctypes.data_as(C.c_void_p), f2.shape[1],
                                    f2.ctypes.data_as(C.c_void_p), pnum, mt.ctypes.data_as(C.c_void_p), n_hyp, dptr(draws), dptr(p1), dptr(p2),
                                    dptr(cnum), dptr(state), dptr(inl), C.byref(res)))
    out = _result(res, cnum, state, inl[:pnum])
    out["pset1"], out["pset2"] = p1[:pnum].T.copy(), p2[:pnum].T.copy()
    return out


def vo_bench(pset1, pset2, draws, reps=20, device=0):
    p1, p2 = np.ascontiguousarray(f64(pset1).T), np.ascontiguousarray(f64(pset2).T)
    draws = i32(draws).reshape(-1, 4)
    ms = C.c_double(0)
    check(lib.pre3_vo_bench(int(device), p1.shape[0], dptr(p1), dptr(p2), draws.shape[0], dptr(draws), int(reps), C.byref(ms)))
    return ms.value
