/*
 * knn_gateway.c -- MEX gateway: `[neighborIds, neighborDistances] = kNearestNeighbors(dataMatrix, queryMatrix, k)`
 * (kNearestNeighbors.m:1, called from inittialize_depth.m:13) on the MI355X through libpre3.so.
 * Build:  mex -output kNearestNeighbors mex/knn_gateway.c -Iinclude -L3pre_amd/lib -lpre3
 * dataMatrix: N x D, queryMatrix: M x D (one point per ROW, column-major doubles as MATLAB stores them); outputs M x k,
 * 1-based ids, Euclidean distances, ties in ascending index order (MATLAB's stable sort, kNearestNeighbors.m:36).
 *
 * NOT compiled in the build container (no MATLAB / mex.h there).
 */
#include "mex.h"
#include "pre3.h"

void mexFunction(int nout, mxArray *out[], int nin, const mxArray *in[])
{
    int N, D, M, k, rc;
    mxArray *dist;
    if (nin != 3) mexErrMsgTxt("kNearestNeighbors: three inputs required (dataMatrix, queryMatrix, k)");
    if (nout > 2) mexErrMsgTxt("kNearestNeighbors: too many outputs");
    if (!mxIsDouble(in[0]) || !mxIsDouble(in[1]) || mxIsSparse(in[0]) || mxIsSparse(in[1])) mexErrMsgTxt("kNearestNeighbors: dataMatrix and queryMatrix must be full doubles");
    N = (int)mxGetM(in[0]); D = (int)mxGetN(in[0]); M = (int)mxGetM(in[1]);
    if ((int)mxGetN(in[1]) != D) mexErrMsgTxt("kNearestNeighbors: dataMatrix and queryMatrix must have the same number of columns");
    k = (int)mxGetScalar(in[2]);
    if (k < 1 || k > N) mexErrMsgTxt("kNearestNeighbors: k must be between 1 and the number of data points");   /* the .m indexes position(1:k) */
    out[0] = mxCreateDoubleMatrix(M, k, mxREAL);
    dist = mxCreateDoubleMatrix(M, k, mxREAL);
    rc = pre3_knn_f64(0, D, N, mxGetPr(in[0]), M, mxGetPr(in[1]), k, mxGetPr(out[0]), mxGetPr(dist));
    if (rc != PRE3_OK) { mxDestroyArray(dist); mexErrMsgTxt(pre3_last_error()); }
    if (nout > 1) out[1] = dist; else mxDestroyArray(dist);
}
