/* Declarations-only subset of MATLAB's MEX C API (the C Matrix API / MEX library functions that the gateways under mex/ call), written from the public API
 * documentation for ONE purpose: `gcc -fsyntax-only` over this repository's own gateways (tests/test_mex_syntax.py), so that a typo in a
 * gateway is caught without MATLAB.  It defines nothing, links nothing, pins nothing, and is not used to build any reference code. */
#ifndef PRE3_TESTS_MEX_API_DECL_H
#define PRE3_TESTS_MEX_API_DECL_H
#include <stddef.h>
#include <stdbool.h>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS, mxSINGLE_CLASS,
               mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS,
               mxFUNCTION_CLASS } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
void mexErrMsgTxt(const char *msg);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexWarnMsgTxt(const char *msg);
int mexPrintf(const char *fmt, ...);
int mexAtExit(void (*fn)(void));
void mexLock(void);
void mexUnlock(void);
bool mexIsLocked(void);
int mexCallMATLAB(int nlhs, mxArray *plhs[], int nrhs, mxArray *prhs[], const char *name);

void *mxMalloc(size_t n);
void *mxCalloc(size_t n, size_t size);
void mxFree(void *p);
void mxDestroyArray(mxArray *a);

mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity flag);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag);
mxArray *mxCreateLogicalMatrix(mwSize m, mwSize n);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames);
mxArray *mxCreateString(const char *s);

size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
mxLogical *mxGetLogicals(const mxArray *a);
mwIndex *mxGetIr(const mxArray *a);
mwIndex *mxGetJc(const mxArray *a);
mxClassID mxGetClassID(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *fieldname);
void mxSetField(mxArray *a, mwIndex index, const char *fieldname, mxArray *value);

bool mxIsDouble(const mxArray *a);
bool mxIsSingle(const mxArray *a);
bool mxIsSparse(const mxArray *a);
bool mxIsComplex(const mxArray *a);
bool mxIsEmpty(const mxArray *a);
bool mxIsNumeric(const mxArray *a);
bool mxIsStruct(const mxArray *a);
bool mxIsLogical(const mxArray *a);
bool mxIsChar(const mxArray *a);
double mxGetInf(void);
double mxGetNaN(void);
#endif
