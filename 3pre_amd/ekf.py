"""Host-side mirror of the reference's EKF interface for the hot path, over the C ABI (include/pre3.h).

Names, argument meaning and error behaviour follow the MATLAB functions they replace (paths under
matlab_code/ of the reference):

    EkfFilter                     <-> @ekf_filter/ekf_filter.m:27-89 (x_k_k, p_k_k, x_k_km1, p_k_km1, std_z ...)
      .ekf_prediction(u)          <-> @ekf_filter/ekf_prediction.m:29 -> predict_state_and_covariance.m:27
      .search_IC_matches()        <-> search_IC_matches.m:31-44 (projection, Jacobians, S_i)
      .matching(k1, zc)           <-> matching_sift_based.m:119-134 (window gate on siftmatch's output)
      .ransac_hypotheses(hyp)     <-> ransac_hypotheses.m:27-85 (draws are an input: MATLAB's RNG is not reproducible)
      .ekf_update_li_inliers()    <-> @ekf_filter/ekf_update_li_inliers.m:45-58
      .rescue_hi_inliers()        <-> @ekf_filter/rescue_hi_inliers.m:29-47
      .ekf_update_hi_inliers()    <-> @ekf_filter/ekf_update_hi_inliers.m:45-58
      .ekf_update_all()           <-> @ekf_filter/ekf_update_all.m:46-62
    update(x, P, H, R, z, h)      <-> update.m:27   (stateless drop-in: host arrays in, host arrays out)
    predict_state_and_covariance  <-> predict_state_and_covariance.m:27 (u passed explicitly instead of fv.m's disk read)

All compute runs in libpre3.so on the GPU; this module only marshals numpy arrays.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Cam, Pre3Error, addr, check, dptr, f64, i32, lib

CHI2INV_2_95 = 5.9915          # rescue_hi_inliers.m:29


def _cam(cam):
    if isinstance(cam, dict):
        cam = [cam[k] for k in ("f", "Cx", "Cy", "k1", "k2", "nRows", "nCols")]
    return Cam(*[float(v) for v in cam])


class EkfFilter:
    """Device-resident filter state.  dtype: 'f64' or 'f32' (covariance path); geometry is always fp64."""

    def __init__(self, cam, lm_type, dtype="f32", device=0, max_hyp=1000, max_landmarks=None, std_z=1.0):
        lm_type = i32(lm_type)
        self.N = int(lm_type.shape[0])
        self.dtype = {"f64": _lib.F64, "f32": _lib.F32}[dtype]
        self.std_z = float(std_z)
        self._ctx = C.c_void_p()
        check(lib.pre3_create(C.byref(self._ctx), int(device), self.dtype, int(max_landmarks or max(self.N, 1)), int(max_hyp)))
        c = _cam(cam)
        check(lib.pre3_set_cam(self._ctx, C.byref(c)))
        check(lib.pre3_set_map(self._ctx, self.N, dptr(lm_type)))
        self.n = int(lib.pre3_state_size(self._ctx))
        self.lm_type = lm_type
        self.m = 0
        self._st = np.zeros(8, np.int32)                 # pre3_step's statistics block (kept: the wrapper's time is GPU idle time)
        self._st_addr = self._st.ctypes.data

    def close(self):
        if getattr(self, "_ctx", None) and lib is not None:        # (lib is None while the interpreter shuts down)
            lib.pre3_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    # ---- state (set_x_k_k.m / get_x_k_k.m ...)
    def set_x_p_k_k(self, x, P):
        x, P = f64(x), f64(P)
        check(lib.pre3_set_state(self._ctx, _lib.X_K_K, x.shape[0], dptr(x), dptr(P)))

    def set_x_p_k_km1(self, x, P):
        x, P = f64(x), f64(P)
        check(lib.pre3_set_state(self._ctx, _lib.X_K_KM1, x.shape[0], dptr(x), dptr(P)))

    def _get(self, which, want_P=True):
        x = np.empty(self.n)
        P = np.empty((self.n, self.n)) if want_P else None
        check(lib.pre3_get_state(self._ctx, which, self.n, dptr(x), dptr(P)))
        return x, P

    def get_x_k_k(self):
        return self._get(_lib.X_K_K, False)[0]

    def get_p_k_k(self):
        return self._get(_lib.X_K_K)[1]

    def get_x_k_km1(self):
        return self._get(_lib.X_K_KM1, False)[0]

    def get_p_k_km1(self):
        return self._get(_lib.X_K_KM1)[1]

    def sync(self):
        check(lib.pre3_sync(self._ctx))

    def defer_hi_update(self, on=True):
        """PRE3_OPT_DEFER_HI: step() returns once the rescue stage is enqueued; the HI update is completed by the next call.
        step()'s n_hi then belongs to the previous step."""
        check(lib.pre3_set_option(self._ctx, 1, int(bool(on))))

    def k9_bf16x3(self, on=None):
        """PRE3_OPT_K9_BF16X3 (fp32 contexts): P <- P - W'W as a three-way bf16 split on the bf16 matrix cores (default) or, off,
        on the f32 MFMA.  Returns the setting in force."""
        if on is not None:
            check(lib.pre3_set_option(self._ctx, 2, int(bool(on))))
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 2, C.byref(v)))
        return bool(v.value)

    def ic_search_was_ranked(self):
        """PRE3_OPT_IC_RANKED: whether the last matching_sift_based() matched on the matrix cores"""
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 4, C.byref(v)))
        return bool(v.value)

    def ic_search_route(self):
        """PRE3_OPT_IC_ROUTE: 2 fused small-problem route, 1 ranked (matrix cores), 0 exact tiled kernel"""
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 7, C.byref(v)))
        return int(v.value)

    def chol_persist(self, on=None):
        """PRE3_OPT_CHOL_PERSIST (fp32 contexts): update.m:32-33 -- the factorisation of S and W = L^-1 [HP | nu] -- as one persistent launch
        (default) or, off, one launch per 64-column panel.  Returns whether the persistent form is in effect."""
        if on is not None:
            check(lib.pre3_set_option(self._ctx, 3, int(bool(on))))
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 3, C.byref(v)))
        return bool(v.value)

    def k9_overlap(self, on=None):
        """PRE3_OPT_K9_OVERLAP (fp32 contexts with the persistent factorisation): update.m:37's P - K*S*K' = P - sum_J W_J'W_J accumulated panel by
        panel inside the factorisation's launch, on the CUs it leaves idle (default); off: the down-date starts when the factorisation has
        finished.  Bit-identical results.  Returns the setting in force."""
        if on is not None:
            check(lib.pre3_set_option(self._ctx, 5, int(bool(on))))
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 5, C.byref(v)))
        return bool(v.value)

    def pend_hi(self, on=None):
        """PRE3_OPT_PEND_HI (fp32 contexts with the persistent factorisation, together with defer_hi_update): the HI update's down-date of P is not
        launched behind the update but taken along by the next step's launches (prediction, H*P, the LI update's consumers): P is swept once per
        step.  Results agree with the default form to fp32 rounding.  Returns the setting in force."""
        if on is not None:
            check(lib.pre3_set_option(self._ctx, 8, int(bool(on))))
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 8, C.byref(v)))
        return bool(v.value)

    def step_tail(self, on=None):
        """PRE3_OPT_STEP_TAIL (fp32 contexts): rescue_hi_inliers + ekf_update_hi_inliers (up to 32 landmarks) inside the LI update's persistent
        launch, P swept once per step; off (default: measured no faster on MI355X): as launches of their own behind it.  Same inlier sets, x / P equal to fp32 rounding.
        Returns the setting in force."""
        if on is not None:
            check(lib.pre3_set_option(self._ctx, 6, int(bool(on))))
        v = C.c_int(0)
        check(lib.pre3_get_option(self._ctx, 6, C.byref(v)))
        return bool(v.value)

    # ---- map management between steps (map_management.m:27-79); the policy stays with the caller
    def _refresh_map(self):
        self.N = int(lib.pre3_get_map(self._ctx, None))
        t = np.zeros(max(self.N, 1), np.int32)
        lib.pre3_get_map(self._ctx, dptr(t))
        self.lm_type = t[:self.N].copy()
        self.n = int(lib.pre3_state_size(self._ctx))
        self.m = 0

    def delete_features(self, del_idx):
        """delete_features.m:54-74 -> delete_a_feature.m:47-51 (0-based landmark indices)."""
        d = i32(sorted(int(v) for v in del_idx))
        check(lib.pre3_map_delete(self._ctx, int(d.shape[0]), dptr(d)))
        self._refresh_map()

    def add_features_inverse_depth(self, uvd, std_pxl, initial_rho):
        """add_features_inverse_depth.m:27-47; uvd is (k, 2) distorted pixels, initial_rho scalar or (k,)."""
        uvd = f64(uvd).reshape(-1, 2)
        rho = f64(np.broadcast_to(initial_rho, (uvd.shape[0],)))
        check(lib.pre3_map_add_inverse_depth(self._ctx, int(uvd.shape[0]), dptr(uvd), C.c_double(std_pxl), dptr(rho)))
        self._refresh_map()

    def inversedepth_2_cartesian(self, linearity_index_threshold=0.1):
        """inversedepth_2_cartesian.m:27-76; returns the per-landmark converted flags."""
        conv = np.zeros(max(self.N, 1), np.int32)
        check(lib.pre3_map_inversedepth_2_cartesian(self._ctx, C.c_double(linearity_index_threshold), dptr(conv)))
        conv = conv[:self.N].copy()
        self._refresh_map()
        return conv

    def map_management(self, del_idx=(), new_uvd=None, std_pxl=1.0, initial_rho=0.5, linearity_index_threshold=None):
        """map_management.m:27-79 as one call (delete_features, inversedepth_2_cartesian unless the threshold is None, the new features of
        initialize_features): one pass over the covariance.  Returns the per-landmark converted flags of the map before the call."""
        d = i32(sorted(int(v) for v in del_idx))
        uvd = f64(new_uvd).reshape(-1, 2) if new_uvd is not None else np.zeros((0, 2))
        rho = f64(np.broadcast_to(initial_rho, (uvd.shape[0],)))
        conv = np.zeros(max(self.N, 1), np.int32)
        n_before = self.N
        check(lib.pre3_map_management(self._ctx, int(d.shape[0]), dptr(d), C.c_double(-1.0 if linearity_index_threshold is None else linearity_index_threshold),
                                      dptr(conv), int(uvd.shape[0]), dptr(uvd), C.c_double(std_pxl), dptr(rho)))
        self._refresh_map()
        return conv[:n_before].copy()

    # ---- IC search on the device (search_IC_matches.m:31-44 + matching_sift_based.m:104-149)
    def set_descriptors(self, desc, first=0):
        """features_info(first+i).Descriptor; desc is (128, count) as MATLAB stores it (or (count, 128) C-order rows)."""
        desc = np.asarray(desc, dtype=np.float64)
        cols = np.ascontiguousarray(desc.T) if desc.shape[0] == 128 and desc.ndim == 2 else f64(desc)
        assert cols.ndim == 2 and cols.shape[1] == 128, "descriptors are 128-vectors"
        check(lib.pre3_set_descriptors(self._ctx, int(first), int(cols.shape[0]), dptr(cols)))

    def get_descriptors(self):
        out = np.zeros((max(self.N, 1), 128))
        check(lib.pre3_get_descriptors(self._ctx, 0, self.N, dptr(out)))
        return out[:self.N].T.copy()                     # (128, N) like [features_info.Descriptor]

    def load_scan(self, Descriptor_RAW, SCALE_ORIENT_POS_RAW):
        """SCAN_SIFT.Descriptor_RAW (128 x K2) and SCAN_SIFT.SCALE_ORIENT_POS_RAW (4 x K2) of SIFT_result%04d.mat."""
        d = np.ascontiguousarray(np.asarray(Descriptor_RAW, dtype=np.float64).reshape(128, -1).T)
        p = np.ascontiguousarray(np.asarray(SCALE_ORIENT_POS_RAW, dtype=np.float64).reshape(4, -1).T)
        assert d.shape[0] == p.shape[0], "Descriptor_RAW and SCALE_ORIENT_POS_RAW disagree on K2"
        check(lib.pre3_set_scan(self._ctx, int(d.shape[0]), dptr(d), dptr(p)))

    def matching_sift_based(self, thresh=1.5, strict_reference=True):
        """The whole IC-search stage in one call; the accepted matches become the measurement list."""
        N = max(self.N, 1)
        nm, m = C.c_int32(0), C.c_int32(0)
        meas, z, pairs = np.zeros(N, np.int32), np.zeros((N, 2)), np.zeros((N, 3), np.int32)
        check(lib.pre3_ic_search(self._ctx, C.c_double(thresh), int(bool(strict_reference)), C.byref(nm), C.byref(m), dptr(meas), dptr(z), dptr(pairs)))
        self.m = int(m.value)
        return dict(meas_idx=meas[:self.m].copy(), z=z[:self.m].copy(), match_idx=pairs[:nm.value, :2].T.copy(),
                    accepted=pairs[:nm.value, 2].copy())

    # ---- step stages
    def ekf_prediction(self, u):
        u = f64(u)
        assert u.shape == (7,), "u = [dX(3); dq(4)]"
        check(lib.pre3_predict(self._ctx, dptr(u)))

    def predict_camera_measurements(self, which=_lib.X_K_KM1, clear_first=True):
        """Also computes the Jacobians (calculate_derivatives.m) -- the reference always calls them as a pair."""
        check(lib.pre3_project(self._ctx, int(which), int(bool(clear_first))))

    def search_IC_matches(self):
        check(lib.pre3_project(self._ctx, _lib.X_K_KM1, 1))
        check(lib.pre3_innovation(self._ctx))

    def landmark_fields(self):
        N = self.N
        h, has_h = np.zeros((N, 2)), np.zeros(N, np.int32)
        Hc, Hl, S = np.zeros((N, 2, 7)), np.zeros((N, 2, 6)), np.zeros((N, 2, 2))
        check(lib.pre3_get_landmark_fields(self._ctx, dptr(h), dptr(has_h), dptr(Hc), dptr(Hl), dptr(S)))
        return dict(h=h, has_h=has_h, Hc=Hc, Hl=Hl, S=S)

    def matching(self, k1, zc, strict_reference=True):
        """Window gate on siftmatch's pairs: k1[c] = 0-based column of L1 (= c-th predicted landmark list entry
        matched), zc[c] = matched pixel.  Returns the accept flags; accepted candidates become measurements."""
        k1, zc = i32(k1), f64(zc)
        M = int(k1.shape[0])
        acc = np.zeros(M, np.int32)
        check(lib.pre3_window_gate(self._ctx, M, dptr(k1), dptr(zc), int(bool(strict_reference)), dptr(acc)))
        return acc

    def set_measurements(self, meas_idx, z):
        meas_idx, z = i32(meas_idx), f64(z)
        self.m = int(meas_idx.shape[0])
        check(lib.pre3_set_measurements(self._ctx, self.m, dptr(meas_idx), dptr(z)))

    def ransac_hypotheses(self, hyp, threshold=None, early_exit=True):
        hyp = i32(hyp)
        n_draw, k = hyp.shape
        sup = np.zeros(n_draw, np.int32)
        li = np.zeros(max(self.m, 1), np.int32)
        st = np.zeros(4, np.int32)
        thr = self.std_z if threshold is None else float(threshold)     # ransac_hypotheses.m:33
        check(lib.pre3_ransac(self._ctx, n_draw, k, dptr(hyp), C.c_double(thr), int(bool(early_exit)), dptr(sup), dptr(li), dptr(st)))
        return dict(support=sup, li_mask=li[:self.m], best=int(st[0]), iters=int(st[1]), n_hyp=int(st[2]), max_support=int(st[3]))

    def ransac_score_shard(self, hyp, threshold, hyp_begin, hyp_end):
        """Score hypotheses [hyp_begin, hyp_end) only; returns (support_dev_ptr, mask_dev_ptr, mask_words)."""
        hyp = i32(hyp)
        n_draw, k = hyp.shape
        sp, mp, mw = C.c_void_p(), C.c_void_p(), C.c_int(0)
        check(lib.pre3_ransac_score(self._ctx, n_draw, k, dptr(hyp), C.c_double(float(threshold)), int(hyp_begin), int(hyp_end),
                                    C.byref(sp), C.byref(mp), C.byref(mw)))
        return sp.value, mp.value, mw.value

    def ransac_select(self, n_draw, k, early_exit=True, fetch=True):
        """fetch=False: only the four statistics come back (through the mailbox: no stream sync, no D2H of the supports / the mask); the
        winner's flags stay on the device for ekf_update_li_inliers"""
        st = np.zeros(4, np.int32)
        if not fetch:
            check(lib.pre3_ransac_select(self._ctx, int(n_draw), int(k), int(bool(early_exit)), None, None, dptr(st)))
            return dict(best=int(st[0]), iters=int(st[1]), n_hyp=int(st[2]), max_support=int(st[3]))
        sup = np.zeros(n_draw, np.int32)
        li = np.zeros(max(self.m, 1), np.int32)
        check(lib.pre3_ransac_select(self._ctx, int(n_draw), int(k), int(bool(early_exit)), dptr(sup), dptr(li), dptr(st)))
        return dict(support=sup, li_mask=li[:self.m], best=int(st[0]), iters=int(st[1]), n_hyp=int(st[2]), max_support=int(st[3]))

    def set_comm(self, comm):
        """attach a comm.Comm (None detaches): ransac_sharded_stream then runs its all-reduce on this filter's stream"""
        check(lib.pre3_set_comm(self._ctx, comm._h if comm is not None else None))
        self._comm = comm                                   # keeps the communicator alive as long as the filter borrows it

    def ransac_sharded_stream(self, hyp, threshold, early_exit=True, fetch=True):
        """pre3_ransac_sharded: ransac_hypotheses with the draws dealt to the communicator's ranks -- scoring, ncclAllReduce and selection
        on the library's stream, one wait; same dict as ransac_hypotheses, identical on every rank"""
        hyp = i32(hyp)
        n_draw, k = hyp.shape
        st = np.zeros(4, np.int32)
        if not fetch:
            check(lib.pre3_ransac_sharded(self._ctx, n_draw, k, dptr(hyp), C.c_double(float(threshold)), int(bool(early_exit)), None, None, dptr(st)))
            return dict(best=int(st[0]), iters=int(st[1]), n_hyp=int(st[2]), max_support=int(st[3]))
        sup = np.zeros(n_draw, np.int32)
        li = np.zeros(max(self.m, 1), np.int32)
        check(lib.pre3_ransac_sharded(self._ctx, n_draw, k, dptr(hyp), C.c_double(float(threshold)), int(bool(early_exit)), dptr(sup), dptr(li), dptr(st)))
        return dict(support=sup, li_mask=li[:self.m], best=int(st[0]), iters=int(st[1]), n_hyp=int(st[2]), max_support=int(st[3]))

    def test_stall(self, release):
        """test hook: 0 parks a kernel on the context's stream until test_stall(1) (a peer that stalls inside a collective)"""
        check(lib.pre3_test_stall(self._ctx, int(release)))

    def ransac_export(self, n_draw, support_ptr, mask_ptr):
        check(lib.pre3_ransac_export(self._ctx, int(n_draw), C.c_void_p(support_ptr), C.c_void_p(mask_ptr)))

    def ransac_import(self, n_draw, support_ptr, mask_ptr):
        check(lib.pre3_ransac_import(self._ctx, int(n_draw), C.c_void_p(support_ptr), C.c_void_p(mask_ptr)))

    def ekf_update_li_inliers(self):
        check(lib.pre3_update_li(self._ctx))

    def rescue_hi_inliers(self, chi2=CHI2INV_2_95):
        hi = np.zeros(max(self.m, 1), np.int32)
        check(lib.pre3_rescue(self._ctx, C.c_double(chi2), dptr(hi)))
        return hi[:self.m]

    def ekf_update_hi_inliers(self):
        check(lib.pre3_update_hi(self._ctx))

    def ekf_update_all(self):
        check(lib.pre3_update_all(self._ctx))

    def set_flags(self, li=None, hi=None):
        li = None if li is None else i32(li)
        hi = None if hi is None else i32(hi)
        check(lib.pre3_set_flags(self._ctx, dptr(li), dptr(hi)))

    def get_flags(self):
        li, hi = np.zeros(max(self.m, 1), np.int32), np.zeros(max(self.m, 1), np.int32)
        check(lib.pre3_get_flags(self._ctx, dptr(li), dptr(hi)))
        return li[:self.m], hi[:self.m]

    def step(self, u, meas_idx, z, hyp, threshold=None, early_exit=True, chi2=CHI2INV_2_95):
        """One '1PRE' filter step (mono_slam.m:153-187)."""
        u, meas_idx, z, hyp = f64(u), i32(meas_idx), f64(z), i32(hyp)
        self.m = int(meas_idx.shape[0])
        n_draw, k = hyp.shape
        st = self._st
        rc = lib.pre3_step(self._ctx, addr(u), self.m, addr(meas_idx), addr(z), n_draw, k, addr(hyp),
                           self.std_z if threshold is None else threshold, 1 if early_exit else 0, chi2, self._st_addr)
        if rc:
            check(rc)
        s = st.tolist()
        return dict(best=s[0], iters=s[1], n_hyp=s[2], max_support=s[3], n_li=s[4], n_hi=s[5])

    def step_all(self, u, meas_idx, z):
        """pre3_step_all -- mono_slam.m's 'PURE_EKF' branch (:153-162, :199): ekf_prediction, search_IC_matches' projection / Jacobians / S_i, the
        measurements, ekf_update_all, as one call (the same arithmetic as the four calls, fewer launches)"""
        u, meas_idx, z = f64(u), i32(meas_idx), f64(z)
        self.m = int(meas_idx.shape[0])
        check(lib.pre3_step_all(self._ctx, dptr(u), self.m, dptr(meas_idx), dptr(z)))

    def step_predicted(self, hyp, threshold=None, early_exit=True, chi2=CHI2INV_2_95):
        """mono_slam.m:178-187 (RANSAC, LI update, rescue, HI update) behind ekf_prediction() and matching_sift_based() (or search_IC_matches()
        + set_measurements()): the installed measurements, pre3_step's device-driven launches."""
        hyp = i32(hyp)
        n_draw, k = hyp.shape
        rc = lib.pre3_step_predicted(self._ctx, n_draw, k, addr(hyp), self.std_z if threshold is None else threshold, 1 if early_exit else 0, chi2, self._st_addr)
        if rc:
            check(rc)
        s = self._st.tolist()
        return dict(best=s[0], iters=s[1], n_hyp=s[2], max_support=s[3], n_li=s[4], n_hi=s[5])

    # ---- measurement hooks
    def timer_start(self):
        check(lib.pre3_timer_start(self._ctx))

    def timer_stop(self):
        ms = C.c_double(0)
        check(lib.pre3_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    def kernel_timing(self, enable):
        """True / 1: time every K9 launch; N > 1: one launch in N; False / 0: off."""
        check(lib.pre3_kernel_timing(self._ctx, int(enable)))

    def kernel_timing_read(self):
        """launches / total_ms / flops (SYRK count n(n+1)r) / bytes of the bracketed launches; fused = how many of them were launches of the
        persistent factorisation with the down-date inside, fact_flops = the factorisation + solve flops those also executed"""
        fu, ff = C.c_int(0), C.c_double(0)
        check(lib.pre3_kernel_timing_info(self._ctx, C.byref(fu), C.byref(ff)))
        n, ms, fl, by = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
        check(lib.pre3_kernel_timing_read(self._ctx, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return dict(launches=n.value, total_ms=ms.value, flops=fl.value, bytes=by.value, fused=fu.value, fact_flops=ff.value)

    def bench_downdate(self, r, reps):
        ms = C.c_double(0)
        check(lib.pre3_bench_downdate(self._ctx, int(r), int(reps), C.byref(ms)))
        return ms.value


# ---------------------------------------------------------------------------------------------------
# stateless drop-ins
# ---------------------------------------------------------------------------------------------------
def _to_ell(H, n):
    """rows of H (dense r x n ndarray or scipy.sparse) -> (width, nnz[r], col[r*width], val[r*width])"""
    try:
        import scipy.sparse as sp
        if sp.issparse(H):
            H = H.tocsr()
            r = H.shape[0]
            nnz = np.diff(H.indptr).astype(np.int32)
            width = max(int(nnz.max()) if r else 1, 1)
            col = np.zeros((r, width), np.int32)
            val = np.zeros((r, width))
            for a in range(r):
                s, e = H.indptr[a], H.indptr[a + 1]
                col[a, :e - s] = H.indices[s:e]
                val[a, :e - s] = H.data[s:e]
            return width, nnz, col, val
    except ImportError:  # pragma: no cover
        pass
    H = np.asarray(H, dtype=np.float64)
    r = H.shape[0]
    nz = H != 0
    nnz = nz.sum(axis=1).astype(np.int32)
    width = max(int(nnz.max()) if r else 1, 1)
    col = np.zeros((r, width), np.int32)
    val = np.zeros((r, width))
    for a in range(r):
        idx = np.nonzero(nz[a])[0]
        col[a, :len(idx)] = idx
        val[a, :len(idx)] = H[a, idx]
    return width, nnz, col, val


def update(x_km1_k, p_km1_k, H, R, z, h, dtype="f64", device=0, want_K=True):
    """[x_k_k, p_k_k, K] = update(x_km1_k, p_km1_k, H, R, z, h)   (update.m:27).

    H: r x n (dense or scipy sparse, at most 16 non-zeros per row -- the reference's rows have 13);
    R: r x r or None for eye(r).  An empty z returns the inputs and K = 0 (update.m:50-55)."""
    x, P = f64(x_km1_k).ravel(), f64(p_km1_k)
    n = x.shape[0]
    z, h = f64(z).ravel(), f64(h).ravel()
    r = z.shape[0]
    xo, Po = np.empty(n), np.empty((n, n))
    if r == 0:
        check(lib.pre3_update_ell(int(device), {"f64": 0, "f32": 1}[dtype], n, 0, dptr(x), dptr(P), 1, None, None, None, None, None, None,
                                  dptr(xo), dptr(Po), None))
        return xo, Po, 0
    width, nnz, col, val = _to_ell(H, n)
    if width > 16:
        raise Pre3Error(-1, "update: H has a row with %d non-zeros; the measurement rows of this filter have 13 (max 16)" % width)
    col, val = i32(col), f64(val)
    Rm = None if R is None else f64(R)
    K = np.zeros((r, n)) if want_K else None
    check(lib.pre3_update_ell(int(device), {"f64": 0, "f32": 1}[dtype], n, r, dptr(x), dptr(P), width, dptr(nnz), dptr(col), dptr(val),
                              dptr(Rm), dptr(z), dptr(h), dptr(xo), dptr(Po), dptr(K)))
    return xo, Po, (K.T.copy() if want_K else None)       # K_out is n x r column-major == (r x n row-major)'


def predict_state_and_covariance(X_k, P_k, u, cam=None, dtype="f64", device=0):
    """[X_km1_k, P_km1_k] = predict_state_and_covariance(X_k, P_k, type, SD_A, SD_alpha) (predict_state_and_covariance.m:27) with
    u = [dX; dq] explicit (the .m file reads it from disk through fv.m:47).  Stateless: host arrays in, host arrays out
    (pre3_predict_dense).  `cam` is accepted for symmetry with the other mirrors and unused, as in the reference."""
    X_k, P_k, u = f64(X_k).ravel(), f64(P_k), f64(u).ravel()
    n = X_k.shape[0]
    if P_k.shape != (n, n) or u.shape[0] != 7:
        raise Pre3Error(-1, "predict_state_and_covariance: P must be %d x %d and u = [dX(3); dq(4)]" % (n, n))
    xo, Po = np.empty(n), np.empty((n, n))
    check(lib.pre3_predict_dense(int(device), {"f64": _lib.F64, "f32": _lib.F32}[dtype], n, dptr(X_k), dptr(P_k), dptr(u), dptr(xo), dptr(Po)))
    return xo, Po


def generate_state_vector_pattern(lm_type, has_z, z, n=None):
    """[state_vector_pattern, z_id, z_euc] = generate_state_vector_pattern(features_info, x) (generate_state_vector_pattern.m:29-53):
    lm_type[N] (0 inverse depth, 1 cartesian), has_z[N] (features_info(i).z non-empty), z[N][2].  Host bookkeeping (index arithmetic)."""
    lm_type = np.asarray(lm_type).astype(int)
    if n is None:
        n = 13 + int(np.sum(np.where(lm_type == _lib.INVDEPTH, 6, 3)))
    pat = np.zeros((n, 4))
    z_id, z_euc, pos = [], [], 13
    for i, t in enumerate(lm_type):
        if t == _lib.INVDEPTH:
            if has_z[i]:
                pat[pos:pos + 3, 0] = 1; pat[pos + 3:pos + 5, 1] = 1; pat[pos + 5, 2] = 1
                z_id.append(z[i])
            pos += 6
        else:
            if has_z[i]:
                pat[pos:pos + 3, 3] = 1
                z_euc.append(z[i])
            pos += 3
    return pat, np.array(z_id, float).reshape(-1, 2).T, np.array(z_euc, float).reshape(-1, 2).T


def compute_hypothesis_support_fast(xi, cam, state_vector_pattern, z_id, z_euc, threshold, device=0):
    """[hypothesis_support, positions_li_inliers_id, positions_li_inliers_euc] = compute_hypothesis_support_fast(xi, cam,
    state_vector_pattern, z_id, z_euc, threshold) (compute_hypothesis_support_fast.m:27).  z_id / z_euc are 2 x n as in MATLAB
    (empty arrays allowed); the masks come back as boolean vectors ([] for an empty class, as the reference returns)."""
    xi = f64(xi).ravel()
    n = xi.shape[0]
    pat = np.asfortranarray(np.asarray(state_vector_pattern, dtype=np.float64))
    if pat.shape != (n, 4):
        raise Pre3Error(-1, "compute_hypothesis_support_fast: state_vector_pattern must be %d x 4" % n)
    zi = np.asfortranarray(np.asarray(z_id, dtype=np.float64).reshape(2, -1)) if np.size(z_id) else np.zeros((2, 0), order="F")
    ze = np.asfortranarray(np.asarray(z_euc, dtype=np.float64).reshape(2, -1)) if np.size(z_euc) else np.zeros((2, 0), order="F")
    n_id, n_euc = zi.shape[1], ze.shape[1]
    sup = C.c_int32(0)
    pid, peu = np.zeros(max(n_id, 1), np.int32), np.zeros(max(n_euc, 1), np.int32)
    c = _cam(cam)
    check(lib.pre3_hypothesis_support(int(device), n, dptr(xi), C.addressof(c), dptr(pat), n_id, dptr(zi) if n_id else None, n_euc,
                                      dptr(ze) if n_euc else None, float(threshold), C.addressof(sup), dptr(pid), dptr(peu)))
    return int(sup.value), pid[:n_id].astype(bool), peu[:n_euc].astype(bool)
