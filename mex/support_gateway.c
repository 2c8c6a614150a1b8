/*
 * support_gateway.c -- MEX gateway: `[hypothesis_support, positions_li_inliers_id, positions_li_inliers_euc] =
 * compute_hypothesis_support_fast(xi, cam, state_vector_pattern, z_id, z_euc, threshold)` (compute_hypothesis_support_fast.m:27,
 * called once per hypothesis from ransac_hypotheses.m:72) on the MI355X through libpre3.so.
 * Build:  mex -output compute_hypothesis_support_fast mex/support_gateway.c -Iinclude -L3pre_amd/lib -lpre3
 *
 * One call scores ONE hypothesis, as the reference's loop does; it is the drop-in for parity checks and for un-modified callers.
 * The fast path is `pre3_mex('ransac', ...)` (mex/ekf_ctx_gateway.c), which scores all hypotheses of a frame in one launch.
 * cam is the struct of initialize_cam.m:69-78.  Outputs: double scalar, logical row vectors ([] for an empty class, as the .m returns).
 *
 * NOT compiled in the build container (no MATLAB / mex.h there).
 */
#include <string.h>
#include "mex.h"
#include "pre3.h"

static double cam_field(const mxArray *cam, const char *name)
{
    const mxArray *f = mxGetField(cam, 0, name);
    if (!f || !mxIsDouble(f) || mxGetNumberOfElements(f) < 1) mexErrMsgTxt("compute_hypothesis_support_fast: cam lacks a numeric field (f, Cx, Cy, k1, k2, nRows, nCols)");
    return mxGetScalar(f);
}

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    pre3_cam cam;
    int n, n_id, n_euc, rc, j;
    int32_t support = 0, *pid = NULL, *peu = NULL;
    if (nin != 6) mexErrMsgTxt("compute_hypothesis_support_fast: six inputs required (xi, cam, state_vector_pattern, z_id, z_euc, threshold)");
    if (nout > 3) mexErrMsgTxt("compute_hypothesis_support_fast: too many outputs");
    if (!mxIsStruct(in[1])) mexErrMsgTxt("compute_hypothesis_support_fast: cam must be a struct");
    cam.f = cam_field(in[1], "f"); cam.Cx = cam_field(in[1], "Cx"); cam.Cy = cam_field(in[1], "Cy"); cam.k1 = cam_field(in[1], "k1");
    cam.k2 = cam_field(in[1], "k2"); cam.nRows = cam_field(in[1], "nRows"); cam.nCols = cam_field(in[1], "nCols");
    n = (int)mxGetNumberOfElements(in[0]);
    if (!mxIsDouble(in[2]) || mxIsSparse(in[2]) || (int)mxGetM(in[2]) != n || (int)mxGetN(in[2]) != 4)
        mexErrMsgTxt("compute_hypothesis_support_fast: state_vector_pattern must be a full length(xi) x 4 double matrix");
    n_id = mxIsEmpty(in[3]) ? 0 : (int)mxGetN(in[3]);
    n_euc = mxIsEmpty(in[4]) ? 0 : (int)mxGetN(in[4]);
    if ((n_id && mxGetM(in[3]) != 2) || (n_euc && mxGetM(in[4]) != 2)) mexErrMsgTxt("compute_hypothesis_support_fast: z_id and z_euc must be 2 x n");
    if (n_id) pid = (int32_t *)mxCalloc(n_id, sizeof(int32_t));
    if (n_euc) peu = (int32_t *)mxCalloc(n_euc, sizeof(int32_t));
    rc = pre3_hypothesis_support(0, n, mxGetPr(in[0]), &cam, mxGetPr(in[2]), n_id, n_id ? mxGetPr(in[3]) : NULL, n_euc,
                                 n_euc ? mxGetPr(in[4]) : NULL, mxGetScalar(in[5]), &support, pid, peu);
    if (rc != PRE3_OK) { if (pid) mxFree(pid); if (peu) mxFree(peu); mexErrMsgTxt(pre3_last_error()); }
    out[0] = mxCreateDoubleScalar((double)support);
    if (nout > 1) {
        if (n_id) { mxLogical *l; out[1] = mxCreateLogicalMatrix(1, n_id); l = mxGetLogicals(out[1]); for (j = 0; j < n_id; ++j) l[j] = pid[j] != 0; }
        else out[1] = mxCreateDoubleMatrix(0, 0, mxREAL);            /* positions_li_inliers_id = [] (:75) */
    }
    if (nout > 2) {
        if (n_euc) { mxLogical *l; out[2] = mxCreateLogicalMatrix(1, n_euc); l = mxGetLogicals(out[2]); for (j = 0; j < n_euc; ++j) l[j] = peu[j] != 0; }
        else out[2] = mxCreateDoubleMatrix(0, 0, mxREAL);            /* positions_li_inliers_euc = [] (:114) */
    }
    if (pid) mxFree(pid);
    if (peu) mxFree(peu);
}
