/* vo_gateway.c -- MEX binding of pre3_vo_ransac_frames: replaces the hypothesis loop of vodometry_dr_ye.m:171-236
 * (ransac_dr_ye.m + find_transform_matrix_dr_ye.m) with one device call.
 *
 *   [rot, trans, sta, op_match, stat] = pre3_vo(frm1, frm2, match, x1, y1, z1, x2, y2, z2, draws)
 *
 *   frm1, frm2 : 4 x K SIFT frames (rows 1:2 = column,row, 1-based)       match : 2 x pnum (siftmatch output)
 *   x*, y*, z* : rows x cols range images                                  draws : 4 x rst positions (1-based), drawn with
 *                                                                                  ransac_dr_ye.m:28-46's rule by the caller
 *   stat = [nIterationRansac nSupport ErrorMean ErrorStd phi theta psi u(1:7)]
 * NOT compiled in the build container (no MATLAB / mex.h there); build: mex -output pre3_vo mex/vo_gateway.c -Iinclude -L3pre_amd/lib -lpre3
 */
#include <string.h>
#include "mex.h"
#include "pre3.h"

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    pre3_vo_result r;
    int pnum, n_hyp, i, rc, k;
    int32_t *draws, *inl;
    double *m;
    if (nin != 10) mexErrMsgTxt("pre3_vo: ten input arguments required");
    pnum = (int)mxGetN(in[2]); n_hyp = (int)mxGetN(in[9]);
    if (mxGetM(in[9]) != 4) mexErrMsgTxt("pre3_vo: draws must be 4 x rst");
    draws = (int32_t *)mxMalloc(sizeof(int32_t) * 4 * (n_hyp ? n_hyp : 1));
    inl = (int32_t *)mxCalloc(pnum ? pnum : 1, sizeof(int32_t));
    for (i = 0; i < 4 * n_hyp; ++i) draws[i] = (int32_t)mxGetPr(in[9])[i] - 1;
    rc = pre3_vo_ransac_frames(0, (int)mxGetM(in[3]), (int)mxGetN(in[3]), mxGetPr(in[3]), mxGetPr(in[4]), mxGetPr(in[5]), mxGetPr(in[6]),
                               mxGetPr(in[7]), mxGetPr(in[8]), (int)mxGetM(in[0]), (int)mxGetN(in[0]), mxGetPr(in[0]), (int)mxGetN(in[1]),
                               mxGetPr(in[1]), pnum, mxGetPr(in[2]), n_hyp, draws, NULL, NULL, NULL, NULL, inl, &r);
    mxFree(draws);
    if (rc != PRE3_OK) { mxFree(inl); mexErrMsgTxt(pre3_last_error()); }
    out[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
    for (i = 0; i < 3; ++i) for (k = 0; k < 3; ++k) mxGetPr(out[0])[i + 3 * k] = r.rot[3 * i + k];      /* row-major -> column-major */
    if (nout > 1) { out[1] = mxCreateDoubleMatrix(3, 1, mxREAL); memcpy(mxGetPr(out[1]), r.trans, sizeof r.trans); }
    if (nout > 2) out[2] = mxCreateDoubleScalar(r.sta);
    if (nout > 3) {
        out[3] = mxCreateDoubleMatrix(2, r.sta == 4 ? 0 : r.n_support, mxREAL);
        m = mxGetPr(out[3]);
        if (r.sta != 4) for (i = 0, k = 0; i < pnum; ++i) if (inl[i]) { m[2 * k] = mxGetPr(in[2])[2 * i]; m[2 * k + 1] = mxGetPr(in[2])[2 * i + 1]; ++k; }
    }
    if (nout > 4) {
        out[4] = mxCreateDoubleMatrix(1, 14, mxREAL);
        m = mxGetPr(out[4]);
        m[0] = r.n_iterations; m[1] = r.n_support; m[2] = r.error_mean; m[3] = r.error_std;
        memcpy(m + 4, r.euler, sizeof r.euler); memcpy(m + 7, r.u, sizeof r.u);
    }
    mxFree(inl);
}
