/*
 * update_gateway.c -- MEX gateway: `[x_k_k, p_k_k, K] = update(x_km1_k, p_km1_k, H, R, z, h)` (update.m:27)
 * on the MI355X through libpre3.so.  Build:  mex -output update mex/update_gateway.c -Iinclude -L3pre_amd/lib -lpre3
 * The resulting update.mex* shadows update.m; @ekf_filter/ekf_update_{li_inliers,hi_inliers,all}.m keep calling
 * `update(...)` unchanged.  H may be sparse (calculate_derivatives.m:46 wraps every H_i in sparse()) or full.
 *
 * NOT compiled in the build container (no MATLAB / mex.h there).
 */
#include <string.h>
#include "mex.h"
#include "pre3.h"

#define W 16   /* libpre3's ELL width; the filter's rows have 13 non-zeros */

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    int n, r, a, rc;
    int32_t *nnz, *col;
    double *val, *Rrow = NULL;
    const double *x, *P, *z, *h;
    if (nin != 6) mexErrMsgTxt("update: six inputs required (x, P, H, R, z, h)");
    if (nout > 3) mexErrMsgTxt("update: too many outputs");
    n = (int)mxGetNumberOfElements(in[0]);
    r = (int)mxGetM(in[4]) * (int)mxGetN(in[4]);
    if ((int)mxGetM(in[1]) != n || (int)mxGetN(in[1]) != n) mexErrMsgTxt("update: P must be n x n");
    x = mxGetPr(in[0]); P = mxGetPr(in[1]); z = mxGetPr(in[4]); h = mxGetPr(in[5]);
    out[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
    out[1] = mxCreateDoubleMatrix(n, n, mxREAL);
    if (r == 0) {                                  /* update.m:50-55 */
        memcpy(mxGetPr(out[0]), x, sizeof(double) * n);
        memcpy(mxGetPr(out[1]), P, sizeof(double) * (size_t)n * n);
        if (nout > 2) out[2] = mxCreateDoubleScalar(0);
        return;
    }
    if ((int)mxGetM(in[2]) != r || (int)mxGetN(in[2]) != n) mexErrMsgTxt("update: H must be length(z) x n");
    nnz = (int32_t *)mxCalloc(r, sizeof(int32_t));
    col = (int32_t *)mxCalloc((size_t)r * W, sizeof(int32_t));
    val = (double *)mxCalloc((size_t)r * W, sizeof(double));
    if (mxIsSparse(in[2])) {                       /* CSC -> rows */
        const mwIndex *ir = mxGetIr(in[2]), *jc = mxGetJc(in[2]);
        const double *pr = mxGetPr(in[2]);
        int j; mwIndex k;
        for (j = 0; j < n; ++j)
            for (k = jc[j]; k < jc[j + 1]; ++k) {
                a = (int)ir[k];
                if (pr[k] == 0.0) continue;
                if (nnz[a] >= W) mexErrMsgTxt("update: a row of H has more than 16 non-zeros");
                col[a * W + nnz[a]] = j; val[a * W + nnz[a]] = pr[k]; ++nnz[a];
            }
    } else {
        const double *H = mxGetPr(in[2]);
        int j;
        for (j = 0; j < n; ++j)
            for (a = 0; a < r; ++a) {
                double v = H[(size_t)j * r + a];
                if (v == 0.0) continue;
                if (nnz[a] >= W) mexErrMsgTxt("update: a row of H has more than 16 non-zeros");
                col[a * W + nnz[a]] = j; val[a * W + nnz[a]] = v; ++nnz[a];
            }
    }
    /* R: every caller in the reference passes eye(length(z)); pass it through (row-major copy; S must be symmetric anyway) */
    if (!mxIsEmpty(in[3])) {
        const double *Rm = mxGetPr(in[3]);
        int i, j, is_eye = 1;
        if ((int)mxGetM(in[3]) != r || (int)mxGetN(in[3]) != r) mexErrMsgTxt("update: R must be length(z) x length(z)");
        for (i = 0; i < r && is_eye; ++i) for (j = 0; j < r; ++j) if (Rm[(size_t)j * r + i] != (i == j ? 1.0 : 0.0)) { is_eye = 0; break; }
        if (!is_eye) {
            Rrow = (double *)mxMalloc(sizeof(double) * (size_t)r * r);
            for (i = 0; i < r; ++i) for (j = 0; j < r; ++j) Rrow[(size_t)i * r + j] = Rm[(size_t)j * r + i];
        }
    }
    if (nout > 2) out[2] = mxCreateDoubleMatrix(n, r, mxREAL);
    rc = pre3_update_ell(0, PRE3_F64, n, r, x, P, W, nnz, col, val, Rrow, z, h, mxGetPr(out[0]), mxGetPr(out[1]),
                         nout > 2 ? mxGetPr(out[2]) : NULL);
    mxFree(nnz); mxFree(col); mxFree(val); if (Rrow) mxFree(Rrow);
    if (rc != PRE3_OK) mexErrMsgTxt(pre3_last_error());
}
