/*
 * ekf_ctx_gateway.c -- MEX gateway `pre3_mex(command, ...)` exposing the device-resident filter context, for
 * users who want P to stay in HBM across calls instead of crossing PCIe at every `update`
 * (the stateless update gateway moves 2 n^2 doubles per call).  The @ekf_filter methods become one-liners:
 *
 *   ekf_prediction.m:29             pre3_mex('predict', u)                      % u = [dX; dq] from fv.m:47
 *   search_IC_matches.m:31-44       pre3_mex('project', 1, 1); pre3_mex('innovation'); f = pre3_mex('fields')
 *   matching_sift_based.m:119-134   acc = pre3_mex('window_gate', k1-1, zc, 1)
 *   ransac_hypotheses.m:40-80       [li, stats] = pre3_mex('ransac', hyp-1, threshold, 1)
 *   ekf_update_li_inliers.m:57      pre3_mex('update_li')
 *   rescue_hi_inliers.m:32-47       hi = pre3_mex('rescue', 5.9915)
 *   ekf_update_hi_inliers.m:57      pre3_mex('update_hi')
 *   get_x_k_k.m / get_p_k_k.m       [x, P] = pre3_mex('get_state', 0)
 *
 * The context lives in a static guarded by mexAtExit + mexLock (the convention of the reference's Coder MEX,
 * corrcoef_partitioned_mex.c:25-57).  NOT compiled in the build container (no MATLAB / mex.h there).
 */
#include <string.h>
#include "mex.h"
#include "pre3.h"

static pre3_ctx *g_ctx = NULL;
static void at_exit(void) { if (g_ctx) { pre3_destroy(g_ctx); g_ctx = NULL; } }
static void check(int rc) { if (rc != PRE3_OK) mexErrMsgTxt(pre3_last_error()); }

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    char cmd[32];
    if (nin < 1 || mxGetString(in[0], cmd, sizeof cmd)) mexErrMsgTxt("pre3_mex: first argument must be a command string");
    if (!strcmp(cmd, "create")) {            /* pre3_mex('create', cam_struct, types(0/1), 'f32'|'f64', max_hyp) */
        pre3_cam cam; int N = (int)mxGetNumberOfElements(in[2]), i; int32_t *t; char dt[8];
        const char *fn[7] = { "f", "Cx", "Cy", "k1", "k2", "nRows", "nCols" }; double *cf = &cam.f;
        for (i = 0; i < 7; ++i) { mxArray *v = mxGetField(in[1], 0, fn[i]); if (!v) mexErrMsgTxt("pre3_mex: cam field missing"); cf[i] = mxGetScalar(v); }
        mxGetString(in[3], dt, sizeof dt);
        at_exit();
        check(pre3_create(&g_ctx, 0, !strcmp(dt, "f64") ? PRE3_F64 : PRE3_F32, N, (int)mxGetScalar(in[4])));
        mexAtExit(at_exit); if (!mexIsLocked()) mexLock();
        t = (int32_t *)mxMalloc(sizeof(int32_t) * (N ? N : 1));
        for (i = 0; i < N; ++i) t[i] = (int32_t)mxGetPr(in[2])[i];
        check(pre3_set_cam(g_ctx, &cam)); check(pre3_set_map(g_ctx, N, t)); mxFree(t);
        return;
    }
    if (!g_ctx) mexErrMsgTxt("pre3_mex: call pre3_mex('create', ...) first");
    if (!strcmp(cmd, "set_state")) check(pre3_set_state(g_ctx, (int)mxGetScalar(in[1]), (int)mxGetNumberOfElements(in[2]), mxGetPr(in[2]), mxGetPr(in[3])));
    else if (!strcmp(cmd, "get_state")) {
        int n = pre3_state_size(g_ctx);
        out[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
        if (nout > 1) out[1] = mxCreateDoubleMatrix(n, n, mxREAL);
        check(pre3_get_state(g_ctx, (int)mxGetScalar(in[1]), n, mxGetPr(out[0]), nout > 1 ? mxGetPr(out[1]) : NULL));
    }
    else if (!strcmp(cmd, "predict")) check(pre3_predict(g_ctx, mxGetPr(in[1])));
    else if (!strcmp(cmd, "project")) check(pre3_project(g_ctx, (int)mxGetScalar(in[1]), (int)mxGetScalar(in[2])));
    else if (!strcmp(cmd, "innovation")) check(pre3_innovation(g_ctx));
    else if (!strcmp(cmd, "update_li")) check(pre3_update_li(g_ctx));
    else if (!strcmp(cmd, "update_hi")) check(pre3_update_hi(g_ctx));
    else if (!strcmp(cmd, "update_all")) check(pre3_update_all(g_ctx));
    else if (!strcmp(cmd, "fields")) {            /* f = pre3_mex('fields'): struct with h (2xN), has_h (1xN), S (2x2xN), Hc (7x2xN), Hl (6x2xN) */
        const char *fn[5] = { "h", "has_h", "S", "Hc", "Hl" };
        int N = (pre3_state_size(g_ctx) - 13) / 3 + 1, i;   /* upper bound on N; exact N is the length of has_h returned */
        mwSize d3[3];
        mxArray *h, *hh, *S, *Hc, *Hl; int32_t *tmp;
        /* the landmark count is what pre3_set_map received; callers pass it as the 2nd argument */
        if (nin > 1) N = (int)mxGetScalar(in[1]);
        h = mxCreateDoubleMatrix(2, N, mxREAL); hh = mxCreateDoubleMatrix(1, N, mxREAL);
        d3[0] = 2; d3[1] = 2; d3[2] = N; S = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        d3[0] = 7; Hc = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);      /* row-major 2x7 == column-major 7x2 */
        d3[0] = 6; Hl = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        tmp = (int32_t *)mxMalloc(sizeof(int32_t) * (N ? N : 1));
        check(pre3_get_landmark_fields(g_ctx, mxGetPr(h), tmp, mxGetPr(Hc), mxGetPr(Hl), mxGetPr(S)));
        for (i = 0; i < N; ++i) mxGetPr(hh)[i] = tmp[i];
        mxFree(tmp);
        out[0] = mxCreateStructMatrix(1, 1, 5, fn);
        mxSetField(out[0], 0, "h", h); mxSetField(out[0], 0, "has_h", hh); mxSetField(out[0], 0, "S", S);
        mxSetField(out[0], 0, "Hc", Hc); mxSetField(out[0], 0, "Hl", Hl);
    }
    else if (!strcmp(cmd, "window_gate")) {       /* acc = pre3_mex('window_gate', k1 (0-based), zc (2xM), strict) */
        int M = (int)mxGetNumberOfElements(in[1]), i; int32_t *k1 = (int32_t *)mxMalloc(sizeof(int32_t) * (M ? M : 1)), *acc = (int32_t *)mxMalloc(sizeof(int32_t) * (M ? M : 1));
        int rc;
        for (i = 0; i < M; ++i) k1[i] = (int32_t)mxGetPr(in[1])[i];
        rc = pre3_window_gate(g_ctx, M, k1, mxGetPr(in[2]), (int)mxGetScalar(in[3]), acc);
        out[0] = mxCreateDoubleMatrix(1, M, mxREAL);
        for (i = 0; i < M; ++i) mxGetPr(out[0])[i] = acc[i];
        mxFree(k1); mxFree(acc); check(rc);
    }
    else if (!strcmp(cmd, "set_measurements")) {  /* pre3_mex('set_measurements', idx (0-based, ascending), z (2xm)) */
        int m = (int)mxGetNumberOfElements(in[1]), i, rc; int32_t *idx = (int32_t *)mxMalloc(sizeof(int32_t) * (m ? m : 1));
        for (i = 0; i < m; ++i) idx[i] = (int32_t)mxGetPr(in[1])[i];
        rc = pre3_set_measurements(g_ctx, m, idx, mxGetPr(in[2])); mxFree(idx); check(rc);
    }
    else if (!strcmp(cmd, "ransac")) {            /* [li_mask, stats] = pre3_mex('ransac', hyp (n_draw x k, 0-based positions), thr, early_exit) */
        int n_draw = (int)mxGetM(in[1]), k = (int)mxGetN(in[1]), i, j, rc, m;
        int32_t *hyp = (int32_t *)mxMalloc(sizeof(int32_t) * n_draw * k), st[4], *li;
        for (i = 0; i < n_draw; ++i) for (j = 0; j < k; ++j) hyp[i * k + j] = (int32_t)mxGetPr(in[1])[(size_t)j * n_draw + i];
        m = nin > 4 ? (int)mxGetScalar(in[4]) : 4096;               /* number of measurements (size of li_mask) */
        li = (int32_t *)mxCalloc(m, sizeof(int32_t));
        rc = pre3_ransac(g_ctx, n_draw, k, hyp, mxGetScalar(in[2]), (int)mxGetScalar(in[3]), NULL, li, st);
        out[0] = mxCreateDoubleMatrix(1, m, mxREAL);
        for (i = 0; i < m; ++i) mxGetPr(out[0])[i] = li[i];
        if (nout > 1) { out[1] = mxCreateDoubleMatrix(1, 4, mxREAL); for (i = 0; i < 4; ++i) mxGetPr(out[1])[i] = st[i]; }
        mxFree(hyp); mxFree(li); check(rc);
    }
    else if (!strcmp(cmd, "step_predicted")) {    /* stats = pre3_mex('step_predicted', hyp (n_draw x k, 0-based positions), thr, early_exit, chi2): mono_slam.m:178-187 in one call,
                                                     behind pre3_mex('predict', u) and pre3_mex('ic_search', ...) (or 'set_measurements'); stats = [best iters n_hyp max_support n_li n_hi] */
        int n_draw = (int)mxGetM(in[1]), k = (int)mxGetN(in[1]), i, j, rc;
        int32_t *hyp = (int32_t *)mxMalloc(sizeof(int32_t) * n_draw * k), st[8];
        for (i = 0; i < n_draw; ++i) for (j = 0; j < k; ++j) hyp[i * k + j] = (int32_t)mxGetPr(in[1])[(size_t)j * n_draw + i];
        rc = pre3_step_predicted(g_ctx, n_draw, k, hyp, mxGetScalar(in[2]), (int)mxGetScalar(in[3]), nin > 4 ? mxGetScalar(in[4]) : 5.9915, st);
        out[0] = mxCreateDoubleMatrix(1, 6, mxREAL);
        for (i = 0; i < 6; ++i) mxGetPr(out[0])[i] = st[i];
        mxFree(hyp); check(rc);
    }
    else if (!strcmp(cmd, "rescue")) {            /* hi_mask = pre3_mex('rescue', chi2, m) */
        int m = nin > 2 ? (int)mxGetScalar(in[2]) : 4096, i, rc; int32_t *hi = (int32_t *)mxCalloc(m, sizeof(int32_t));
        rc = pre3_rescue(g_ctx, mxGetScalar(in[1]), hi);
        out[0] = mxCreateDoubleMatrix(1, m, mxREAL);
        for (i = 0; i < m; ++i) mxGetPr(out[0])[i] = hi[i];
        mxFree(hi); check(rc);
    }
    else if (!strcmp(cmd, "map_delete")) {        /* pre3_mex('map_delete', idx (0-based, ascending))   delete_features.m:54-74 */
        int k = (int)mxGetNumberOfElements(in[1]), i, rc; int32_t *d = (int32_t *)mxMalloc(sizeof(int32_t) * (k ? k : 1));
        for (i = 0; i < k; ++i) d[i] = (int32_t)mxGetPr(in[1])[i];
        rc = pre3_map_delete(g_ctx, k, d); mxFree(d); check(rc);
    }
    else if (!strcmp(cmd, "map_add")) {           /* pre3_mex('map_add', uvd (2xk), std_pxl, initial_rho (1xk))   add_features_inverse_depth.m:27-47 */
        check(pre3_map_add_inverse_depth(g_ctx, (int)mxGetN(in[1]), mxGetPr(in[1]), mxGetScalar(in[2]), mxGetPr(in[3])));
    }
    else if (!strcmp(cmd, "map_convert")) {       /* converted = pre3_mex('map_convert', 0.1)   inversedepth_2_cartesian.m:27-76 */
        int N = pre3_get_map(g_ctx, NULL), i, rc; int32_t *f = (int32_t *)mxCalloc(N ? N : 1, sizeof(int32_t));
        rc = pre3_map_inversedepth_2_cartesian(g_ctx, mxGetScalar(in[1]), f);
        out[0] = mxCreateDoubleMatrix(1, N, mxREAL);
        for (i = 0; i < N; ++i) mxGetPr(out[0])[i] = f[i];
        mxFree(f); check(rc);
    }
    else if (!strcmp(cmd, "map_management")) {    /* converted = pre3_mex('map_management', del_idx (0-based, ascending), linearity_thr (< 0: no conversion), uvd (2xk), std_pxl, initial_rho (1xk))
                                                     map_management.m:27-79 as one call: delete_features, inversedepth_2_cartesian, the new features -- one pass over P */
        int k = (int)mxGetNumberOfElements(in[1]), N = pre3_get_map(g_ctx, NULL), i, rc;
        int32_t *d = (int32_t *)mxMalloc(sizeof(int32_t) * (k ? k : 1)), *f = (int32_t *)mxCalloc(N ? N : 1, sizeof(int32_t));
        for (i = 0; i < k; ++i) d[i] = (int32_t)mxGetPr(in[1])[i];
        rc = pre3_map_management(g_ctx, k, d, mxGetScalar(in[2]), f, (int)mxGetN(in[3]), mxGetPr(in[3]), mxGetScalar(in[4]), mxGetPr(in[5]));
        out[0] = mxCreateDoubleMatrix(1, N, mxREAL);
        for (i = 0; i < N; ++i) mxGetPr(out[0])[i] = f[i];
        mxFree(d); mxFree(f); check(rc);
    }
    else if (!strcmp(cmd, "set_descriptors")) {   /* pre3_mex('set_descriptors', [features_info.Descriptor] (128xN), first (0-based)) */
        check(pre3_set_descriptors(g_ctx, nin > 2 ? (int)mxGetScalar(in[2]) : 0, (int)mxGetN(in[1]), mxGetPr(in[1])));
    }
    else if (!strcmp(cmd, "set_scan")) {          /* pre3_mex('set_scan', SCAN_SIFT.Descriptor_RAW, SCAN_SIFT.SCALE_ORIENT_POS_RAW) */
        check(pre3_set_scan(g_ctx, (int)mxGetN(in[1]), mxGetPr(in[1]), mxGetPr(in[2])));
    }
    else if (!strcmp(cmd, "ic_search")) {         /* [meas_idx, z, match_idx] = pre3_mex('ic_search', 1.5, strict)   matching_sift_based.m:104-149 */
        int N = pre3_get_map(g_ctx, NULL), i, rc; int32_t nm = 0, m = 0;
        int32_t *meas = (int32_t *)mxCalloc(N ? N : 1, sizeof(int32_t)), *pairs = (int32_t *)mxCalloc(3 * (N ? N : 1), sizeof(int32_t));
        double *z = (double *)mxCalloc(2 * (N ? N : 1), sizeof(double));
        rc = pre3_ic_search(g_ctx, mxGetScalar(in[1]), nin > 2 ? (int)mxGetScalar(in[2]) : 1, &nm, &m, meas, z, pairs);
        if (rc == PRE3_OK) {
            out[0] = mxCreateDoubleMatrix(1, m, mxREAL);
            for (i = 0; i < m; ++i) mxGetPr(out[0])[i] = meas[i] + 1;                       /* 1-based landmark numbers */
            if (nout > 1) { out[1] = mxCreateDoubleMatrix(2, m, mxREAL); memcpy(mxGetPr(out[1]), z, sizeof(double) * 2 * m); }
            if (nout > 2) { out[2] = mxCreateDoubleMatrix(2, nm, mxREAL);
                for (i = 0; i < nm; ++i) { mxGetPr(out[2])[2 * i] = pairs[3 * i] + 1; mxGetPr(out[2])[2 * i + 1] = pairs[3 * i + 1] + 1; } }
        }
        mxFree(meas); mxFree(pairs); mxFree(z); check(rc);
    }
    else if (!strcmp(cmd, "set_option")) {        /* pre3_mex('set_option', option, value): PRE3_OPT_DEFER_HI = 1, PRE3_OPT_K9_BF16X3 = 2, PRE3_OPT_CHOL_PERSIST = 3, PRE3_OPT_K9_OVERLAP = 5 (include/pre3.h) */
        check(pre3_set_option(g_ctx, (int)mxGetScalar(in[1]), (int)mxGetScalar(in[2])));
    }
    else if (!strcmp(cmd, "destroy")) { at_exit(); if (mexIsLocked()) mexUnlock(); }
    else mexErrMsgTxt("pre3_mex: unknown command");
}
