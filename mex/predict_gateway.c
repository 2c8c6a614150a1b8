/*
 * predict_gateway.c -- MEX gateway: `[X_km1_k, P_km1_k] = predict_state_and_covariance(X_k, P_k, type, SD_A_component_filter,
 * SD_alpha_component_filter)` (predict_state_and_covariance.m:27, called from @ekf_filter/ekf_prediction.m:29) on the MI355X
 * through libpre3.so.   Build:  mex -output predict_state_and_covariance mex/predict_gateway.c -Iinclude -L3pre_amd/lib -lpre3
 * The resulting predict_state_and_covariance.mex* shadows the .m file; ekf_prediction.m keeps calling it unchanged.
 *
 * The fork's .m file does not take the odometry increment as an argument: it reads the globals `step_global` / `myCONFIG` and gets
 * [dX_gt, dq_calc] from fv.m:47 (`Calculate_V_Omega_RANSAC_dr_ye` on the frame pair, i.e. a disk read).  The gateway resolves it the
 * same way, by calling back into MATLAB -- `[~,~,~,~,dX,dq] = fv(X_k(1:13), dt, type, sd_a, sd_alpha)` is evaluated exactly where the
 * .m file evaluates it (:59) -- or, when the optional 6th argument u = [dX; dq] is given, uses that (what the level-2 integration of
 * INTEGRATION.md passes).  `type` must be 'constant_velocity' (the only model the reference calls, :61).
 *
 * NOT compiled in the build container (no MATLAB / mex.h there).
 */
#include <string.h>
#include "mex.h"
#include "pre3.h"

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    int n, rc, i;
    double u[7];
    char type[32];
    if (nin < 5 || nin > 6) mexErrMsgTxt("predict_state_and_covariance: five inputs required (X_k, P_k, type, SD_A, SD_alpha) [+ optional u = [dX; dq]]");
    if (nout > 2) mexErrMsgTxt("predict_state_and_covariance: too many outputs");
    if (!mxIsDouble(in[0]) || !mxIsDouble(in[1]) || mxIsSparse(in[1])) mexErrMsgTxt("predict_state_and_covariance: X_k and P_k must be full doubles");
    n = (int)mxGetNumberOfElements(in[0]);
    if ((int)mxGetM(in[1]) != n || (int)mxGetN(in[1]) != n) mexErrMsgTxt("predict_state_and_covariance: P_k must be n x n");
    if (mxGetString(in[2], type, sizeof type) != 0 || strcmp(type, "constant_velocity") != 0)
        mexErrMsgTxt("predict_state_and_covariance: only the 'constant_velocity' model of the reference is implemented");
    if (nin == 6) {
        if (mxGetNumberOfElements(in[5]) != 7) mexErrMsgTxt("predict_state_and_covariance: u must be [dX(3); dq(4)]");
        memcpy(u, mxGetPr(in[5]), sizeof u);
    } else {
        /* [Xv_km1_k, v, w, dq__, dX__] = fv(X_k(1:13), delta_t, type, SD_A, SD_alpha)  (predict_state_and_covariance.m:39; fv.m:27,54-59) */
        mxArray *lhs[5], *rhs[5];
        rhs[0] = mxCreateDoubleMatrix(13, 1, mxREAL);
        memcpy(mxGetPr(rhs[0]), mxGetPr(in[0]), sizeof(double) * 13);
        rhs[1] = mxCreateDoubleScalar(0.1);                         /* delta_t = 0.1 (:35); it does not enter u */
        rhs[2] = (mxArray *)in[2]; rhs[3] = (mxArray *)in[3]; rhs[4] = (mxArray *)in[4];
        if (mexCallMATLAB(5, lhs, 5, rhs, "fv") != 0) mexErrMsgTxt("predict_state_and_covariance: fv() failed");
        if (mxGetNumberOfElements(lhs[4]) != 3 || mxGetNumberOfElements(lhs[3]) != 4) mexErrMsgTxt("predict_state_and_covariance: fv() returned an unexpected dX / dq");
        memcpy(u, mxGetPr(lhs[4]), sizeof(double) * 3);             /* u = [dX__; dq__] (:57) */
        memcpy(u + 3, mxGetPr(lhs[3]), sizeof(double) * 4);
        for (i = 0; i < 5; ++i) mxDestroyArray(lhs[i]);
        mxDestroyArray(rhs[0]); mxDestroyArray(rhs[1]);
    }
    out[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
    out[1] = mxCreateDoubleMatrix(n, n, mxREAL);
    rc = pre3_predict_dense(0, PRE3_F64, n, mxGetPr(in[0]), mxGetPr(in[1]), u, mxGetPr(out[0]), mxGetPr(out[1]));
    if (rc != PRE3_OK) mexErrMsgTxt(pre3_last_error());            /* outputs are MATLAB-owned: freed by the interpreter on error */
}
