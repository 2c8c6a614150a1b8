/*
 * siftmatch_gateway.c -- MEX gateway that makes libpre3.so a drop-in for the reference's `siftmatch` MEX
 * (matlab_code/sift/siftmatch.c:139-250).  Build inside MATLAB on a machine with ROCm:
 *
 *     mex -output siftmatch mex/siftmatch_gateway.c -Iinclude -L3pre_amd/lib -lpre3
 *
 * and put the resulting siftmatch.mex* ahead of matlab_code/sift on the MATLAB path (a MEX file shadows the
 * .m / older MEX of the same name; callers such as matching_sift_based.m:118 stay unchanged).
 *
 * NOT compiled in the build container (it has no MATLAB, hence no mex.h); the C ABI it calls is what the test
 * suite exercises.  Argument checks and messages are the reference gateway's (siftmatch.c:154-190).
 */
#include "mex.h"
#include "pre3.h"

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    enum { L1 = 0, L2, THRESH };
    double thresh = 1.5;
    int K1, K2, ND, M = 0, rc;
    double *pairs, *score;
    mxClassID cls;

    if (nin < 2) mexErrMsgTxt("At least two input arguments required");
    else if (nout > 2) mexErrMsgTxt("Too many output arguments");
    if (!mxIsNumeric(in[L1]) || !mxIsNumeric(in[L2]) || mxGetNumberOfDimensions(in[L1]) > 2 || mxGetNumberOfDimensions(in[L2]) > 2)
        mexErrMsgTxt("L1 and L2 must be two dimensional numeric arrays");
    K1 = (int)mxGetN(in[L1]); K2 = (int)mxGetN(in[L2]); ND = (int)mxGetM(in[L1]);
    if ((int)mxGetM(in[L2]) != ND) mexErrMsgTxt("L1 and L2 must have the same number of rows");
    cls = mxGetClassID(in[L1]);
    if (mxGetClassID(in[L2]) != cls) mexErrMsgTxt("L1 and L2 must be of the same class");
    if (nin == 3) {
        if (!mxIsDouble(in[THRESH]) || mxIsComplex(in[THRESH]) || mxGetNumberOfElements(in[THRESH]) != 1)
            mexErrMsgTxt("THRESH should be a real scalar");
        thresh = *mxGetPr(in[THRESH]);
    } else if (nin > 3) mexErrMsgTxt("At most three arguments are allowed");

    /* scratch owned by MATLAB's allocator so that an error long-jump cannot leak it */
    pairs = (double *)mxMalloc(sizeof(double) * 2 * (K1 > 0 ? K1 : 1));
    score = (double *)mxMalloc(sizeof(double) * (K1 > 0 ? K1 : 1));
    switch (cls) {
    case mxDOUBLE_CLASS: rc = pre3_siftmatch_f64(0, ND, K1, (const double *)mxGetData(in[L1]), K2, (const double *)mxGetData(in[L2]), thresh, pairs, score, &M); break;
    case mxSINGLE_CLASS: rc = pre3_siftmatch_f32(0, ND, K1, (const float *)mxGetData(in[L1]), K2, (const float *)mxGetData(in[L2]), thresh, pairs, score, &M); break;
    case mxINT8_CLASS:   rc = pre3_siftmatch_i8(0, ND, K1, (const int8_t *)mxGetData(in[L1]), K2, (const int8_t *)mxGetData(in[L2]), thresh, pairs, score, &M); break;
    case mxUINT8_CLASS:  rc = pre3_siftmatch_u8(0, ND, K1, (const uint8_t *)mxGetData(in[L1]), K2, (const uint8_t *)mxGetData(in[L2]), thresh, pairs, score, &M); break;
    default: mxFree(pairs); mxFree(score); mexErrMsgTxt("Unsupported numeric class"); return;
    }
    if (rc != PRE3_OK) { mxFree(pairs); mxFree(score); mexErrMsgTxt(pre3_last_error()); }
    out[0] = mxCreateDoubleMatrix(2, M, mxREAL);
    { double *p = mxGetPr(out[0]); int i; for (i = 0; i < 2 * M; ++i) p[i] = pairs[i]; }
    if (nout > 1) {
        int i; double *d;
        out[1] = mxCreateDoubleMatrix(1, M, mxREAL);
        d = mxGetPr(out[1]);
        for (i = 0; i < M; ++i) d[i] = score[i];
    }
    mxFree(pairs); mxFree(score);
}
