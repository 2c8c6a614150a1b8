"""ctypes front-end of the CPU oracle (oracle/pre3_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under 3pre_amd/ imports this package.

All wrappers take/return numpy fp64 arrays; landmark tables are (lm_type, lm_off) int32 arrays
(type 0 = inverse depth (6 params), 1 = Cartesian (3 params); lm_off = 0-based offset into x).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpre3_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pre3_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Cam(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("f", "Cx", "Cy", "k1", "k2", "nRows", "nCols")]


def make_cam(v):
    """v = [f, Cx, Cy, k1, k2, nRows, nCols]"""
    return Cam(*[float(t) for t in v])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        for name in ("orc_siftmatch_f64", "orc_siftmatch_f32", "orc_siftmatch_i8", "orc_siftmatch_u8", "orc_support"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def landmark_table(types):
    types = _i(types)
    dims = np.where(types == 0, 6, 3)
    off = 13 + np.concatenate([[0], np.cumsum(dims)[:-1]]) if len(types) else np.zeros(0)
    return types, _i(off), int(13 + dims.sum())


def process_noise():
    Pn = np.zeros((7, 7))
    lib().orc_process_noise(_p(Pn))
    return Pn


def predict(x, P, u):
    x, P, u = _d(x), _d(P), _d(u)
    n = x.shape[0]
    xo, Po = np.empty(n), np.empty((n, n))
    rc = lib().orc_predict(n, _p(x), _p(P), _p(u), _p(xo), _p(Po))
    assert rc == 0
    return xo, Po


def project(types, off, x, cam, h=None, has_h=None):
    N = len(types)
    h = np.zeros((N, 2)) if h is None else _d(h).copy()
    has_h = np.zeros(N, np.int32) if has_h is None else _i(has_h).copy()
    c = make_cam(cam)
    lib().orc_project(N, _p(_i(types)), _p(_i(off)), _p(_d(x)), C.byref(c), _p(h), _p(has_h))
    return h, has_h


def jacobian(types, off, x, cam, h, has_h):
    N = len(types)
    Hc, Hl = np.zeros((N, 2, 7)), np.zeros((N, 2, 6))
    c = make_cam(cam)
    lib().orc_jacobian(N, _p(_i(types)), _p(_i(off)), _p(_d(x)), C.byref(c), _p(_d(h)), _p(_i(has_h)), _p(Hc), _p(Hl))
    return Hc, Hl


def innovation(types, off, P, Hc, Hl, has_h):
    N = len(types)
    n = P.shape[0]
    S = np.zeros((N, 2, 2))
    lib().orc_innovation(n, N, _p(_i(types)), _p(_i(off)), _p(_d(P)), _p(_d(Hc)), _p(_d(Hl)), _p(_i(has_h)), _p(S))
    return S


def window_gate(pred_idx, k1, zc, h, S, has_S, strict_reference=True):
    M = len(k1)
    acc = np.zeros(M, np.int32)
    lib().orc_window_gate(M, _p(_i(pred_idx)), len(pred_idx), _p(_i(k1)), _p(_d(zc)), _p(_d(h)), _p(_d(S)), _p(_i(has_S)),
                          int(strict_reference), _p(acc))
    return acc


def rescue(types, off, P, Hc, Hl, h, z, ic, li, chi2=5.9915):
    N = len(types)
    n = P.shape[0]
    hi = np.zeros(N, np.int32)
    d2 = np.full(N, np.nan)
    lib().orc_rescue(n, N, _p(_i(types)), _p(_i(off)), _p(_d(P)), _p(_d(Hc)), _p(_d(Hl)), _p(_d(h)), _p(_d(np.nan_to_num(z))),
                     _p(_i(ic)), _p(_i(li)), C.c_double(chi2), _p(hi), _p(d2))
    return hi, d2


def update(x, P, H, R, z, h, want_K=False):
    x, P = _d(x), _d(P)
    n = x.shape[0]
    z, h = _d(z).ravel(), _d(h).ravel()
    r = z.shape[0]
    H = _d(H).reshape(r, n) if r else np.zeros((0, n))
    Rm = None if R is None else _d(R)
    xo, Po = np.empty(n), np.empty((n, n))
    K = np.zeros((n, r)) if want_K else None
    rc = lib().orc_update(n, r, _p(x), _p(P), _p(H), _p(Rm), _p(z), _p(h), _p(xo), _p(Po), _p(K))
    assert rc == 0, rc
    return (xo, Po, K) if want_K else (xo, Po)


def update_landmarks(types, off, sel, x, P, Hc, Hl, z, h):
    x, P = _d(x), _d(P)
    n = x.shape[0]
    sel = _i(sel)
    xo, Po = np.empty(n), np.empty((n, n))
    rc = lib().orc_update_landmarks(n, len(types), _p(_i(types)), _p(_i(off)), len(sel), _p(sel), _p(x), _p(P),
                                    _p(_d(Hc)), _p(_d(Hl)), _p(_d(np.nan_to_num(z))), _p(_d(h)), _p(xo), _p(Po))
    assert rc == 0, rc
    return xo, Po


def support(meas, types, off, xi, cam, z_meas, threshold):
    m = len(meas)
    mask = np.zeros(m, np.int32)
    res = np.zeros(m)
    c = make_cam(cam)
    s = lib().orc_support(m, _p(_i(meas)), _p(_i(types)), _p(_i(off)), _p(_d(xi)), C.byref(c), _p(_d(z_meas)),
                          C.c_double(threshold), _p(mask), _p(res))
    return s, mask, res


def hypothesis_state(sel, types, off, x, P, Hc, Hl, z, h):
    x = _d(x)
    n = x.shape[0]
    xi = np.empty(n)
    sel = _i(sel)
    lib().orc_hypothesis_state(n, len(sel), _p(sel), _p(_i(types)), _p(_i(off)), _p(x), _p(_d(P)), _p(_d(Hc)), _p(_d(Hl)),
                               _p(_d(np.nan_to_num(z))), _p(_d(h)), _p(xi))
    return xi


def ransac(types, off, x, P, Hc, Hl, z, h, ic_list, meas, cam, hyp, threshold, early_exit=True):
    x = _d(x)
    n = x.shape[0]
    hyp = _i(hyp)
    n_draw, k = hyp.shape
    m = len(meas)
    sup = np.zeros(n_draw, np.int32)
    li = np.zeros(m, np.int32)
    out = [C.c_int(0) for _ in range(4)]
    c = make_cam(cam)
    lib().orc_ransac(n, len(types), _p(_i(types)), _p(_i(off)), _p(x), _p(_d(P)), _p(_d(Hc)), _p(_d(Hl)),
                     _p(_d(np.nan_to_num(z))), _p(_d(h)), len(ic_list), _p(_i(ic_list)), m, _p(_i(meas)), C.byref(c),
                     n_draw, k, _p(hyp), C.c_double(threshold), int(early_exit), _p(sup), _p(li),
                     C.byref(out[0]), C.byref(out[1]), C.byref(out[2]), C.byref(out[3]))
    return dict(support=sup, li_mask=li, best=out[0].value, iters=out[1].value, n_hyp=out[2].value, max_support=out[3].value)


_SIFT = {np.dtype(np.float64): "orc_siftmatch_f64", np.dtype(np.float32): "orc_siftmatch_f32",
         np.dtype(np.int8): "orc_siftmatch_i8", np.dtype(np.uint8): "orc_siftmatch_u8"}


def siftmatch(L1, L2, thresh=1.5):
    """L1: ND x K1, L2: ND x K2 (MATLAB orientation: one descriptor per column).
    Returns (matches 2 x M float64 1-based, scores M)."""
    L1, L2 = np.asarray(L1), np.asarray(L2)
    assert L1.dtype == L2.dtype and L1.shape[0] == L2.shape[0]
    fn = getattr(lib(), _SIFT[L1.dtype])
    ND, K1 = L1.shape
    K2 = L2.shape[1]
    a = np.asfortranarray(L1)
    b = np.asfortranarray(L2)
    pairs = np.zeros(2 * max(K1, 1))
    score = np.zeros(max(K1, 1))
    M = fn(ND, K1, a.ctypes.data_as(C.c_void_p), K2, b.ctypes.data_as(C.c_void_p), C.c_double(thresh), _p(pairs), _p(score))
    return pairs[:2 * M].reshape(M, 2).T.copy(), score[:M].copy()


def knn(data, query, k):
    data, query = _d(data), _d(query)
    N, D = data.shape
    M = query.shape[0]
    ids, dist = np.zeros((M, k)), np.zeros((M, k))
    rc = lib().orc_knn(D, N, _p(data), M, _p(query), k, _p(ids), _p(dist))
    assert rc == 0
    return ids, dist


def step(types, off, cam, x_kk, P_kk, u, meas_idx, z_meas, hyp, threshold, early_exit=True, chi2=5.9915):
    """One '1PRE' filter step in the reference's order (mono_slam.m:153-187) built from the oracle pieces.
    meas_idx: measured (= individually compatible) landmarks, ascending; z_meas: their pixels (m x 2);
    hyp: (n_draw x k) positions in the IC list."""
    N = len(types)
    meas_idx = np.asarray(meas_idx, np.int32)
    x1, P1 = predict(x_kk, P_kk, u)                                   # ekf_prediction
    h, has_h = project(types, off, x1, cam)                           # search_IC_matches.m:31
    Hc, Hl = jacobian(types, off, x1, cam, h, has_h)                  # :32
    S = innovation(types, off, P1, Hc, Hl, has_h)                     # :33-44
    z = np.zeros((N, 2))
    z[meas_idx] = z_meas
    ic = np.zeros(N, np.int32)
    ic[meas_idx] = 1
    li = np.zeros(N, np.int32)
    out = dict(x_km1=x1, P_km1=P1, S=S, h_km1=h.copy())
    if len(meas_idx) >= hyp.shape[1] and len(meas_idx) > 0:
        r = ransac(types, off, x1, P1, Hc, Hl, z, h, meas_idx, meas_idx, cam, hyp, threshold, early_exit)
        li[meas_idx] = r["li_mask"]
        out["ransac"] = r
    x2, P2 = update_landmarks(types, off, np.nonzero(li)[0], x1, P1, Hc, Hl, z, h)      # ekf_update_li_inliers
    h2, has2 = project(types, off, x2, cam, h, has_h)                                     # rescue_hi_inliers.m:32
    Hc2, Hl2 = jacobian(types, off, x2, cam, h2, has2)
    hi, d2 = rescue(types, off, P2, Hc2, Hl2, h2, z, ic, li, chi2)
    x3, P3 = update_landmarks(types, off, np.nonzero(hi)[0], x2, P2, Hc2, Hl2, z, h2)   # ekf_update_hi_inliers
    out.update(x_kk=x3, P_kk=P3, li=li[meas_idx], hi=hi[meas_idx], x_li=x2, P_li=P2, d2=d2)
    return out


# ---- map management (unpinned; see pre3_oracle.c)
def map_delete(types, off, x, P, del_idx):
    x, P = _d(x), _d(P)
    n = x.shape[0]
    del_idx = _i(sorted(del_idx))
    xo, Po = np.empty(n), np.empty(n * n)
    nn = lib().orc_map_delete(n, len(types), _p(_i(types)), _p(_i(off)), len(del_idx), _p(del_idx), _p(x), _p(P), _p(xo), _p(Po))
    keep = np.ones(len(types), bool)
    keep[del_idx] = False
    return xo[:nn].copy(), Po[:nn * nn].reshape(nn, nn).copy(), np.asarray(types)[keep]


def map_add(x, P, cam, uvd, std_pxl, initial_rho):
    x, P = _d(x), _d(P)
    n = x.shape[0]
    uvd = _d(uvd).reshape(-1, 2)
    k = uvd.shape[0]
    rho = _d(np.broadcast_to(initial_rho, (k,)))
    nm = n + 6 * k
    xo, Po = np.empty(nm), np.empty(nm * nm)
    c = make_cam(cam)
    nn = lib().orc_map_add(n, k, _p(uvd), C.c_double(std_pxl), _p(rho), C.byref(c), _p(x), _p(P), _p(xo), _p(Po))
    return xo[:nn].copy(), Po[:nn * nn].reshape(nn, nn).copy()


def map_convert(types, x, P, threshold=0.1):
    x, P = _d(x), _d(P)
    n = x.shape[0]
    N = len(types)
    xo, Po = np.empty(n), np.empty(n * n)
    to, conv = np.zeros(N, np.int32), np.zeros(N, np.int32)
    nn = lib().orc_map_convert(n, N, _p(_i(types)), C.c_double(threshold), _p(x), _p(P), _p(xo), _p(Po), _p(to), _p(conv))
    return xo[:nn].copy(), Po[:nn * nn].reshape(nn, nn).copy(), to, conv


def ic_search(types, off, x_km1, P_km1, cam, bank, scan_desc, scan_pos, thresh=1.5, strict=True):
    """search_IC_matches.m:31-44 + matching_sift_based.m:104-149 composed from the restated pieces.
    bank (128, N) landmark descriptors, scan_desc (128, K2), scan_pos (4, K2).  Returns the new bank too."""
    N = len(types)
    h, has_h = project(types, off, x_km1, cam)
    Hc, Hl = jacobian(types, off, x_km1, cam, h, has_h)
    S = innovation(types, off, P_km1, Hc, Hl, has_h)
    has_S = has_h                                        # search_IC_matches.m:36-43: S exists exactly where h does
    pred = np.nonzero(has_h)[0].astype(np.int32)
    out = dict(h=h, has_h=has_h, S=S, pred=pred, match_idx=np.zeros((2, 0), np.int32), accepted=np.zeros(0, np.int32),
               meas_idx=np.zeros(0, np.int32), z=np.zeros((0, 2)), bank=np.array(bank, dtype=np.float64, copy=True))
    if pred.size == 0:                                   # matching_sift_based.m:115-117
        return out
    des1 = np.ascontiguousarray(np.asarray(bank, dtype=np.float64)[:, pred])
    pairs, _ = siftmatch(des1, np.asarray(scan_desc, dtype=np.float64), thresh)       # 1-based (2, M)
    k1 = (pairs[0] - 1).astype(np.int32)
    k2 = (pairs[1] - 1).astype(np.int32)
    zc = np.asarray(scan_pos, dtype=np.float64)[0:2, k2].T.copy()
    acc = window_gate(pred, k1, zc, h, S, has_S, strict)
    z_all, ic = np.zeros((N, 2)), np.zeros(N, np.int32)
    for c in range(len(k1)):
        if acc[c]:
            lm = pred[k1[c]]
            ic[lm] = 1; z_all[lm] = zc[c]                                              # matching_sift_based.m:131-132
            out["bank"][:, lm] = np.asarray(scan_desc)[:, k2[c]]                       # matching_sift_based.m:135
    meas = np.nonzero(ic)[0].astype(np.int32)
    out.update(match_idx=np.stack([k1, k2]), accepted=np.asarray(acc, np.int32), meas_idx=meas, z=z_all[meas])
    return out


# ---- VO front end -----------------------------------------------------------------------------------
def vo_find_transform(pset1, pset2):
    """pset: (3, pnum).  -> rot (3,3), trans (3,), state"""
    p1, p2 = np.ascontiguousarray(_d(pset1).T), np.ascontiguousarray(_d(pset2).T)
    rot, tr = np.zeros(9), np.zeros(3)
    st = lib().orc_vo_find_transform(p1.shape[0], None, _p(p1), _p(p2), _p(rot), _p(tr))
    return rot.reshape(3, 3), tr, int(st)


def vo_gather(x, y, z, frm, sel):
    """x, y, z: (rows, cols) range images; frm (>=2, K) SIFT frames; sel: 1-based keypoint numbers -> (3, pnum)"""
    x, y, z = (np.asfortranarray(_d(a)) for a in (x, y, z))
    frm = np.asfortranarray(_d(frm))
    sel = _d(sel)
    out = np.zeros((len(sel), 3))
    rc = lib().orc_vo_gather(x.shape[0], x.shape[1], x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                             frm.shape[0], frm.ctypes.data_as(C.c_void_p), len(sel), _p(sel), _p(out))
    if rc:
        raise IndexError("vo_gather: a keypoint rounds to a pixel outside the range image")
    return out.T.copy()


def vo_ransac(pset1, pset2, draws):
    p1, p2 = np.ascontiguousarray(_d(pset1).T), np.ascontiguousarray(_d(pset2).T)
    draws = _i(draws).reshape(-1, 4)
    pnum = p1.shape[0]
    cn, inl = np.zeros(len(draws), np.int32), np.zeros(pnum, np.int32)
    out, out2, iout = np.zeros(16), np.zeros(2), np.zeros(6, np.int32)
    rc = lib().orc_vo_ransac(pnum, _p(p1), _p(p2), len(draws), _p(draws), _p(cn), _p(inl), _p(out), _p(out2), _p(iout))
    if rc:
        raise ValueError("vo_ransac: %s" % {-1: "no point farther than 0.4 m (dist undefined)", -2: "fewer than 4 matches"}[rc])
    return dict(cnum=cn, inliers=inl, rot=out[:9].reshape(3, 3).copy(), trans=out[9:12].copy(), euler=out[12:15].copy(), error_mean=out[15],
                error_std=out2[0], dist=out2[1], sta=int(iout[0]), n_support=int(iout[1]), n_iterations=int(iout[2]), best=int(iout[3]))


def R2q(R):
    q = np.zeros(4)
    lib().orc_R2q(_p(_d(R)), _p(q))
    return q
