/*
 * pre3.h -- C ABI of libpre3.so: the MI355X (gfx950) implementation of the per-step hot path of
 * the MATLAB 1-point-RANSAC EKF-SLAM reference ahtamjidi/3PRE.
 *
 * The reference's only native plug-in boundary is the MATLAB MEX interface
 * (matlab_code/sift/siftmatch.c:139-141 `mexFunction`; Coder variant
 * matlab_code/mex_files/CorePar_Ver1/codegen/mex/corrcoef_partitioned/corrcoef_partitioned_mex.c:50-57).
 * A MEX file named like an .m function shadows it, so each entry point below states the .m (or .c)
 * function it replaces; mex/ holds the gateways and INTEGRATION.md the binding recipe.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative pre3_status;
 *     pre3_last_error() gives the message of the calling thread's last failure.  Nothing exits or throws.
 *   - host arrays are caller-owned `double`, MATLAB (column-major) order unless stated; the covariance is
 *     symmetric so its orientation does not matter.  Indices crossing the ABI are 0-based int32
 *     (the MEX gateways convert from MATLAB's 1-based doubles).
 *   - a pre3_ctx owns device-resident filter state (x, P, landmark table, per-landmark h/H/S/z/flags) on
 *     ONE GPU and one HIP stream; calls on one ctx must be serialised by the caller (MATLAB's
 *     interpreter thread does); different contexts are independent.
 *   - dtype selects the storage/compute type of the dense covariance path (P, H*P, S, Cholesky, the
 *     MFMA down-date): PRE3_F64 or PRE3_F32.  The state vector, camera geometry, Jacobians, innovations
 *     and gates are always evaluated in fp64.
 *   - there is no CPU fallback: without a HIP device every call fails with PRE3_E_NODEVICE.
 */
#ifndef PRE3_H
#define PRE3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRE3_API __attribute__((visibility("default")))

typedef struct pre3_ctx pre3_ctx;

typedef enum {
    PRE3_OK = 0,
    PRE3_E_ARG = -1,        /* bad argument / shape mismatch (the MEX gateways map this to mexErrMsgTxt) */
    PRE3_E_NODEVICE = -2,   /* no HIP device / HIP runtime failure at init */
    PRE3_E_HIP = -3,        /* a HIP call or kernel failed */
    PRE3_E_STATE = -4,      /* call order violated (e.g. update before predict/project) */
    PRE3_E_NUMERIC = -5,    /* S not positive definite */
    PRE3_E_NOMEM = -6,
    PRE3_E_COMM = -7        /* RCCL: library not found, communicator creation failed, or a collective reported an error */
} pre3_status;

enum { PRE3_F64 = 0, PRE3_F32 = 1 };
enum { PRE3_INVDEPTH = 0, PRE3_CARTESIAN = 1 };   /* features_info(i).type */
enum { PRE3_X_K_K = 0, PRE3_X_K_KM1 = 1 };        /* which estimate (ekf_filter.m:63-87 fields) */

/* cam struct fields used by the path (initialize_cam.m:69-78) */
typedef struct { double f, Cx, Cy, k1, k2, nRows, nCols; } pre3_cam;

PRE3_API const char *pre3_last_error(void);
PRE3_API int pre3_device_count(void);
PRE3_API const char *pre3_version(void);

/* ---- context ----------------------------------------------------------------------------------- */

/* device = HIP ordinal; max_landmarks / max_hyp size the device buffers once (no allocation afterwards). */
PRE3_API int pre3_create(pre3_ctx **out, int device, int dtype, int max_landmarks, int max_hyp);
PRE3_API int pre3_destroy(pre3_ctx *ctx);
PRE3_API int pre3_sync(pre3_ctx *ctx);                       /* wait for the ctx stream */
PRE3_API int pre3_set_cam(pre3_ctx *ctx, const pre3_cam *cam);

/* Landmark table = [features_info.type] in state order (add_feature_to_info_vector_my_version_sift.m:37-60).
 * Resets all per-landmark fields (h, H, S, z, flags), as update_features_info.m:30-44 does each step. */
PRE3_API int pre3_set_map(pre3_ctx *ctx, int N, const int32_t *lm_type);
PRE3_API int pre3_state_size(pre3_ctx *ctx);                 /* n = 13 + 6*N_id + 3*N_euc */

/* x(n), P(n x n) -> device as x_k_k / p_k_k (set_x_k_k.m / set_p_k_k.m), or as x_k_km1 / p_k_km1. */
PRE3_API int pre3_set_state(pre3_ctx *ctx, int which, int n, const double *x, const double *P);
PRE3_API int pre3_get_state(pre3_ctx *ctx, int which, int n, double *x, double *P);   /* P may be NULL */

/* ---- a2: predict_state_and_covariance.m:27-143 (called by @ekf_filter/ekf_prediction.m:29) ------ */
/* u = [dX(3); dq(4)], the visual-odometry increment the reference reads through fv.m:47.
 * (x_k_k, p_k_k) -> (x_k_km1, p_k_km1); the covariance is transformed IN PLACE (only rows/cols 4:7
 * and the 7x7 pose block change), so p_k_k is no longer available afterwards. */
PRE3_API int pre3_predict(pre3_ctx *ctx, const double u[7]);

/* Stateless drop-in for `[X_km1_k, P_km1_k] = predict_state_and_covariance(X_k, P_k, type, SD_A_component_filter,
 * SD_alpha_component_filter)` (predict_state_and_covariance.m:27, caller @ekf_filter/ekf_prediction.m:29): host in, host out.
 * `type` is 'constant_velocity' in every call of the reference and the two standard deviations are unused by this fork (:98-102 hard-
 * code Pn); the increment u = [dX; dq] the .m file reads from disk through fv.m:47 is an explicit argument (the MEX gateway resolves it,
 * INTEGRATION.md).  n = 13 + 6 N_id + 3 N_euc; P, P_out: n x n.
 * The device context behind this entry (P at capacity and the work buffers) is kept between calls, one per (device, dtype), grown when a larger
 * state arrives; calls are serialised on it; pre3_release_scratch() frees it. */
PRE3_API int pre3_predict_dense(int device, int dtype, int n, const double *x, const double *P, const double u[7], double *x_out, double *P_out);

/* ---- a3/a4: predict_camera_measurements.m:27-68 + calculate_derivatives.m:27-60 ------------------ */
/* Projects every landmark at the chosen estimate and linearises it (compact H_i = 2x7 pose block +
 * 2x6 landmark block; columns 8:13 of the reference's H are zero).  clear_first=1 empties h/H first
 * (the state search_IC_matches.m:31 sees after update_features_info.m:40); clear_first=0 keeps the
 * previous h of landmarks that are not predicted now (rescue_hi_inliers.m:32-33, quirk Q7). */
PRE3_API int pre3_project(pre3_ctx *ctx, int which, int clear_first);

/* ---- a5: search_IC_matches.m:33-44: S_i = H_i P H_i' + R_i (R_i = eye(2)) for predicted landmarks */
PRE3_API int pre3_innovation(pre3_ctx *ctx);

/* read back per-landmark fields (any pointer may be NULL): h[2N], has_h[N], Hc[14N] (2x7 row-major),
 * Hl[12N] (2x6 row-major), S[4N] */
PRE3_API int pre3_get_landmark_fields(pre3_ctx *ctx, double *h, int32_t *has_h, double *Hc, double *Hl, double *S);

/* matching_sift_based.m:119-134 window gate.  Candidate c pairs the k1[c]-th PREDICTED landmark (the
 * L1 column siftmatch matched) with pixel zc[2c..2c+1]; accepted candidates become individually
 * compatible and get z.  strict_reference=1 reproduces quirk Q5 (S of the c-th predicted landmark).
 * accept_out[M] may be NULL. */
PRE3_API int pre3_window_gate(pre3_ctx *ctx, int M, const int32_t *k1, const double *zc, int strict_reference,
                              int32_t *accept_out);

/* Direct form: landmarks meas_idx[m] (ascending) are individually compatible with pixels z[2m]
 * (what matching_sift_based.m:131-134 leaves in features_info). */
PRE3_API int pre3_set_measurements(pre3_ctx *ctx, int m, const int32_t *meas_idx, const double *z);

/* ---- a6-a8: ransac_hypotheses.m:27-85 ----------------------------------------------------------- */
/* hyp[n_draw*k]: per hypothesis k positions in the individually-compatible list (what
 * select_random_match.m:40-51 draws with randperm; MATLAB's legacy RNG stream is not reproducible,
 * so the draws are an input).  early_exit=1 replays the reference's adaptive termination (quirk Q1)
 * over the supports; 0 uses all n_draw.  [hyp_begin, hyp_end) restricts the hypotheses THIS context
 * scores (multi-GPU sharding, see pre3_ransac_score / pre3_ransac_select).
 * Outputs (may be NULL): support[n_draw] (int32; -1 for hypotheses after the exit point),
 * li_mask[m] (int32 0/1 in measurement order), stats[4] = {best, iterations, n_hyp, max_support}. */
PRE3_API int pre3_ransac(pre3_ctx *ctx, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit,
                         int32_t *support, int32_t *li_mask, int32_t stats[4]);

/* Stateless drop-in for `[hypothesis_support, positions_li_inliers_id, positions_li_inliers_euc] =
 * compute_hypothesis_support_fast(xi, cam, state_vector_pattern, z_id, z_euc, threshold)` (compute_hypothesis_support_fast.m:27,
 * called from ransac_hypotheses.m:72): host in, host out.  xi: hypothesis state (n); state_vector_pattern: n x 4 column-major
 * 0/1 doubles exactly as generate_state_vector_pattern.m:29-53 builds it (columns: inverse-depth position / angles / rho, cartesian
 * xyz -- xi(logical(pattern(:,c))) is taken in state order); z_id: 2 x n_id, z_euc: 2 x n_euc column-major.  Inverse-depth rule
 * residual < min(residual) + threshold (:70), cartesian rule residual < threshold (:109); the quaternion is used un-normalised (:46).
 * positions_*: int32 0/1 per measurement (may be NULL).  A pattern whose counts do not match n_id / n_euc is PRE3_E_ARG
 * (MATLAB's reshape at :40 fails there too). */
PRE3_API int pre3_hypothesis_support(int device, int n, const double *xi, const pre3_cam *cam, const double *state_vector_pattern,
                                     int n_id, const double *z_id, int n_euc, const double *z_euc, double threshold,
                                     int32_t *support_out, int32_t *positions_li_inliers_id, int32_t *positions_li_inliers_euc);

/* Sharded form: score hypotheses [hyp_begin,hyp_end) only, leaving their supports (int32, others 0)
 * and inlier bitmasks in device buffers; *support_dev / *mask_dev receive DEVICE pointers to
 * int32[n_draw] and uint32[n_draw * mask_words] so the caller can all-reduce them (RCCL) in place;
 * pre3_ransac_select then replays the termination rule on the (reduced) buffers. */
PRE3_API int pre3_ransac_score(pre3_ctx *ctx, int n_draw, int k, const int32_t *hyp, double threshold,
                               int hyp_begin, int hyp_end, void **support_dev, void **mask_dev, int *mask_words);
PRE3_API int pre3_ransac_select(pre3_ctx *ctx, int n_draw, int k, int early_exit,
                                int32_t *support, int32_t *li_mask, int32_t stats[4]);
/* Copy the context's support / mask buffers to (export) or from (import) caller-owned DEVICE buffers of
 * int32[n_draw] and uint32[n_draw*mask_words] -- e.g. torch tensors handed to an RCCL all-reduce.  Either
 * pointer may be NULL.  Synchronous on return. */
PRE3_API int pre3_ransac_export(pre3_ctx *ctx, int n_draw, void *support_dst_dev, void *mask_dst_dev);
PRE3_API int pre3_ransac_import(pre3_ctx *ctx, int n_draw, const void *support_src_dev, const void *mask_src_dev);

/* ---- a9: update.m:27-56 via the @ekf_filter wrappers --------------------------------------------- */
/* ekf_update_li_inliers.m:45-58: prior (x_k_km1,p_k_km1), rows = low-innovation inliers */
PRE3_API int pre3_update_li(pre3_ctx *ctx);
/* rescue_hi_inliers.m:29-47: re-project + re-linearise at x_k_k, chi-square gate (no +R, quirk Q6).
 * hi_mask[m] (may be NULL) in measurement order. */
PRE3_API int pre3_rescue(pre3_ctx *ctx, double chi2, int32_t *hi_mask);
/* ekf_update_hi_inliers.m:45-58: prior (x_k_k,p_k_k), rows = high-innovation inliers */
PRE3_API int pre3_update_hi(pre3_ctx *ctx);
/* ekf_update_all.m:46-62 ('PURE_EKF'): prior km1, rows = all individually compatible landmarks */
PRE3_API int pre3_update_all(pre3_ctx *ctx);
/* read/force the per-measurement flags: li/hi[m] int32 in measurement order (NULL = leave) */
PRE3_API int pre3_get_flags(pre3_ctx *ctx, int32_t *li, int32_t *hi);
PRE3_API int pre3_set_flags(pre3_ctx *ctx, const int32_t *li, const int32_t *hi);

/* One whole filter step of mono_slam.m:153-187 ('1PRE'): predict, project+linearise, S_i,
 * [measurements given], RANSAC, LI update, rescue, HI update -- enqueued back to back on the ctx
 * stream.  stats[8] = {best, iterations, n_hyp, max_support, n_li, n_hi, 0, 0} (may be NULL). */
PRE3_API int pre3_step(pre3_ctx *ctx, const double u[7], int m, const int32_t *meas_idx, const double *z,
                       int n_draw, int k, const int32_t *hyp, double threshold, int early_exit, double chi2,
                       int32_t stats[8]);

/* mono_slam.m:178-187 -- RANSAC, LI update, rescue, HI update -- behind a prediction and an IC search the caller has already run on the
 * context (pre3_predict, then pre3_ic_search or pre3_project + pre3_innovation + pre3_set_measurements): the measurements are the installed
 * ones, the launches are pre3_step's (the persistent factorisation with the down-date inside, the device-driven HI update), none of them
 * sized by a host poll.  stats as pre3_step.  This is what a frame loop that matches on the device calls instead of pre3_step. */
/* mono_slam.m:153-162 + :199, the 'PURE_EKF' branch (config_file.m:21, EST_METHOD): prediction, projection / Jacobians / S_i of every landmark
 * and ONE update with all individually compatible measurements (@ekf_filter/ekf_update_all.m:46-62) as one call -- the same arithmetic as
 * pre3_predict + pre3_project + pre3_innovation + pre3_set_measurements + pre3_update_all, four launches fewer.  meas_idx strictly ascending. */
PRE3_API int pre3_step_all(pre3_ctx *ctx, const double u[7], int m, const int32_t *meas_idx, const double *z /* [2 m] */);
PRE3_API int pre3_step_predicted(pre3_ctx *ctx, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit, double chi2,
                                 int32_t stats[8]);

/* Stateless drop-in for `[x,P,K] = update(x,P,H,R,z,h)` (update.m:27): host in, host out.
 * H is r x n given by rows in ELL form: row a has nnz[a] <= width entries (col[a*width+t], val[...]);
 * R is r x r dense or NULL for eye(r) (every caller in the reference passes eye).  K_out (n x r,
 * column-major) may be NULL.  r == 0 returns the inputs (update.m:50-55). */
PRE3_API int pre3_update_ell(int device, int dtype, int n, int r, const double *x, const double *P, int width,
                             const int32_t *nnz, const int32_t *col, const double *val, const double *R,
                             const double *z, const double *h, double *x_out, double *P_out, double *K_out);

/* Options.  PRE3_OPT_DEFER_HI = 1: pre3_step returns as soon as the rescue stage is enqueued; the HI count is polled and the HI
 * update (ekf_update_hi_inliers.m) launched by the NEXT call on the context (any entry point: step, get_state, sync ...), so the
 * caller's own time between two steps overlaps the device's rescue stage.  Results are identical; stats[5] of pre3_step then
 * reports the previous step's HI count and stats[7] = 1 says so.  Default 0: pre3_step completes the HI update itself. */
#define PRE3_OPT_DEFER_HI 1
/* PRE3_OPT_K9_BF16X3 (fp32 contexts; default 1, or the environment's PRE3_K9_B3): the covariance down-date of update.m:38-46,
 * P <- P - W'W, multiplies on the bf16 matrix cores: every f32 entry of W is split exactly into three bf16 values and six of
 * the nine partial products are accumulated in f32 (the dropped ones are below f32 rounding), 16/6 of the f32 matrix rate at
 * f32 accuracy.  0: plain f32 MFMA (bitwise an fmaf chain).  No effect on fp64 contexts. */
#define PRE3_OPT_K9_BF16X3 2
/* PRE3_OPT_CHOL_PERSIST (fp32 contexts with PRE3_OPT_K9_BF16X3; default 1, or the environment's PRE3_CHOL_FORM): update.m:32-33 -- the
 * factorisation of S and W = L^-1 [HP | nu] -- as ONE persistent launch (pre3_cholp.hip) for updates of up to 13 x 64 rows;
 * 0: one launch per 64-column panel (the form every fp64 context and every larger update uses).  get: whether the form is in effect. */
#define PRE3_OPT_CHOL_PERSIST 3
/* PRE3_OPT_IC_RANKED (read only): 1 if the last pre3_ic_search ran matching_sift_based.m:118's siftmatch on the matrix cores (bf16 distance
 * GEMM ranks + exact re-evaluation; N * K2 >= 65536 and every descriptor inside the route's bounds), 0 if it took the exact VALU kernel.
 * The environment's PRE3_IC_RANK=0 forces the latter.  Results are bit-identical either way. */
#define PRE3_OPT_IC_RANKED 4
/* PRE3_OPT_K9_OVERLAP (fp32 contexts with PRE3_OPT_CHOL_PERSIST; default 1, or the environment's PRE3_K9_OVERLAP): update.m:37's
 * P - K*S*K' = P - sum_J W_J'W_J is accumulated panel by panel INSIDE the persistent factorisation's launch, by workgroups on the CUs that
 * launch leaves idle (as many tile groups as there are CUs left; the rest, if any, in the launch that follows).  0: the down-date only
 * starts when the factorisation has finished.  Results are bit-identical either way. */
#define PRE3_OPT_K9_OVERLAP 5
/* PRE3_OPT_STEP_TAIL (fp32 contexts with PRE3_OPT_K9_OVERLAP; default 0, or the environment's PRE3_TAIL): inside pre3_step / pre3_step_predicted
 * the rescue stage (rescue_hi_inliers.m:29-47) and the HI update of up to 32 rescued landmarks (ekf_update_hi_inliers.m:45-58, update.m:27-48) run
 * INSIDE the LI update's persistent launch: the rescued landmarks' rows are one more panel of the same block factorisation, P is read and written
 * once per step (P - W'W - W~'W~), and update.m:42-46 of both updates is one rows / columns 3..6 pass with the product of the two normalisation
 * Jacobians, carried by the next prediction's launch.  Same inlier sets; x / P agree with the call-by-call sequence to fp32 rounding (the
 * intermediate P_LI is never rounded to fp32).  0 (default): the rescue stage and the HI update as launches of their own -- on MI355X the
 * in-launch form's hand-offs between workgroups cost what the second sweep of P saves (DESIGN.md section 5d has the measured timeline).
 * pre3_get_option returns the setting; whether a step used it also depends on the launch carrying every tile of P (one live fp32 context). */
#define PRE3_OPT_STEP_TAIL 6
/* PRE3_OPT_IC_ROUTE (read only): how the last pre3_ic_search matched -- 2: the fused small-problem route (N * K2 <= 2^20 pairs, N <= 4096, K2 <= 2048:
 * the sizes of the reference's own runs): siftmatch.c:97-116 exactly on 32 x 32 pair tiles over every landmark of the map, then ONE workgroup that
 * stacks the predicted landmarks, merges the tiles, applies Lowe's test and the window gate, refreshes the accepted landmarks' descriptors and
 * writes the result block into host memory -- two launches; 1: the ranked route of PRE3_OPT_IC_RANKED; 0: the exact 64 x 64 tiled kernel.
 * The environment's PRE3_IC_FUSED=0 disables route 2.  Results are bit-identical on all three. */
#define PRE3_OPT_IC_ROUTE 7
/* PRE3_OPT_PEND_HI (fp32 contexts with PRE3_OPT_K9_OVERLAP; default: the environment's PRE3_PEND_HI, else 0): inside pre3_step the covariance down-date of
 * the HI update (update.m:37-38 as ekf_update_hi_inliers.m:58 calls it) is not launched behind the update; P - W~'W~ stays PENDING across the step
 * boundary: the next pre3_step's prediction transforms W~ with P, its H*P / S_i launch subtracts (H W~')W~, and the consumers of its LI update's
 * persistent launch take W~ as the panels in front of panel 0 -- P is read and written ONCE per step instead of twice.  Every other call on the
 * context (and every path of pre3_step that cannot take the pending rows) first runs the down-date as the launch it would have been.  The same
 * arithmetic except that P between the two updates is never rounded to fp32: results agree with the default form to fp32 rounding, not to the bit. */
#define PRE3_OPT_PEND_HI 8
PRE3_API int pre3_set_option(pre3_ctx *ctx, int option, int value);
PRE3_API int pre3_get_option(pre3_ctx *ctx, int option, int *value_out);

/* ---- map management on the device (map_management.m:27-79) ----------------------------- */
/* These act on (x_k_k, p_k_k) between steps, as map_management.m:140 does, and keep P resident: P <- A P A' (+D) with
 * a sparse A (selection rows / the 6x13 and 3x6 Jacobians of the reference), evaluated by two gather passes through a
 * second ld x ld buffer (allocated on first use).  Per-landmark fields (h, H, S, z, flags) are cleared afterwards,
 * as update_features_info.m:30-44 does.  The POLICY (which landmarks to delete, which pixels to initialise) stays on
 * the host, as in delete_features.m:31-52 / initialize_features.m. */
/* delete_features.m:54-74 -> delete_a_feature.m:47-51: remove landmarks del_idx[n_del] (ascending, 0-based). */
PRE3_API int pre3_map_delete(pre3_ctx *ctx, int n_del, const int32_t *del_idx);
/* add_features_inverse_depth.m:27-47 (hinv_my_version.m:26-53 + add_a_feature_covariance_inverse_depth.m:27-90):
 * append n_new inverse-depth landmarks observed at distorted pixels uvd[2*n_new] with initial inverse depths
 * initial_rho[n_new]; std_rho = initial_rho^2 * 0.01 as the reference hard-codes (:41). */
PRE3_API int pre3_map_add_inverse_depth(pre3_ctx *ctx, int n_new, const double *uvd, double std_pxl, const double *initial_rho);
/* inversedepth_2_cartesian.m:27-76: convert every inverse-depth landmark whose linearity index is below the threshold
 * (0.1 in the reference) to a Cartesian point; converted_out[N] (may be NULL) receives the flags. */
PRE3_API int pre3_map_inversedepth_2_cartesian(pre3_ctx *ctx, double linearity_threshold, int32_t *converted_out);
/* map_management.m:27-79 as one call: delete_features (:33), inversedepth_2_cartesian (:48; convert_threshold < 0: skipped) and the
 * addition of initialize_features (:58-66), in that order, with ONE pass over the covariance (their row maps compose: each only selects or
 * recombines rows of the old state).  converted_out (may be null): per landmark of the map BEFORE the call, 1 = converted.
 * The result is what the three calls in sequence give (bit for bit when nothing is converted; to rounding in the entries that pair
 * a converted landmark with a new one). */
PRE3_API int pre3_map_management(pre3_ctx *ctx, int n_del, const int32_t *del_idx, double convert_threshold, int32_t *converted_out,
                                 int n_new, const double *uvd, double std_pxl, const double *initial_rho);
/* current landmark table: returns N, writes lm_type_out[N] if not NULL */
PRE3_API int pre3_get_map(pre3_ctx *ctx, int32_t *lm_type_out);

/* ---- the IC-search stage on the device (search_IC_matches.m:31-44 + matching_sift_based.m:104-149) ---- */
/* features_info(i).Descriptor (128 x 1 double each, add_feature_to_info_vector_my_version_sift.m): desc is 128 x count
 * column-major for landmarks first .. first+count-1.  The bank follows the map through pre3_map_* (deleted landmarks
 * drop out, new ones start at zero until set).  Asynchronous: the array is copied into pinned memory of the context's
 * before the call returns (the caller may reuse it at once) and reaches the device on the context's stream. */
PRE3_API int pre3_set_descriptors(pre3_ctx *ctx, int first, int count, const double *desc);
PRE3_API int pre3_get_descriptors(pre3_ctx *ctx, int first, int count, double *desc);
/* the frame's SIFT set as stored in SIFT_result%04d.mat (SIFT_extract_save.m:68-69): SCAN_SIFT.Descriptor_RAW (128 x K2)
 * and SCAN_SIFT.SCALE_ORIENT_POS_RAW (4 x K2; rows 1:2 = pixel u,v), both column-major doubles.  Asynchronous like
 * pre3_set_descriptors: copied (and checked against the matrix-core matcher's bounds) on the host, pulled over PCIe and
 * packed on the context's stream; the call neither waits for queued work nor ends in a read-back. */
PRE3_API int pre3_set_scan(pre3_ctx *ctx, int K2, const double *descriptor_raw, const double *scale_orient_pos_raw);
/* One call = search_IC_matches.m:31-44 (h, H, S at the prediction) + matching_sift_based.m:104-149: stack the descriptors
 * of the predicted landmarks, siftmatch(des1', Descriptor_RAW) (double class, thresh 1.5 in the reference), window
 * gate (strict_reference = 1 reproduces quirk Q5), z / individually_compatible / descriptor refresh for the accepted.
 * Outputs: *n_matches = size(match_idx, 2); *m = number of accepted (= measurements installed, ascending landmark order);
 * meas_idx_out[m], z_out[2m] (optional); pairs_out[3*n_matches] (optional) = (k1, k2, accepted) per match, 0-based. */
PRE3_API int pre3_ic_search(pre3_ctx *ctx, double thresh, int strict_reference, int32_t *n_matches_out, int32_t *m_out,
                            int32_t *meas_idx_out, double *z_out, int32_t *pairs_out);

/* ---- the VO front end's 4-point 3D-3D RANSAC (code_from_dr_ye/) ------------------------------------ */
typedef struct pre3_vo_result {
    double rot[9];          /* final rotation, row-major (find_transform_matrix_dr_ye.m on the winner's inliers, vodometry_dr_ye.m:221) */
    double trans[3];
    double euler[3];        /* R2e(rot): phi, theta, psi (vodometry_dr_ye.m:245-248); zeros when sta < 1 */
    double u[7];            /* Calculate_V_Omega_RANSAC_dr_ye.m:40-50: [T; R2q(R)] -- what pre3_predict takes; identity unless sta == 1 */
    double error_mean, error_std;   /* RANSAC_STAT.ErrorMean / ErrorStd (vodometry_dr_ye.m:222-225) */
    double dist;            /* inlier radius scale of ransac_dr_ye.m:20-23 (threshold = 0.001*dist on squared distances) */
    int32_t sta;            /* state of the final fit: 1 ok, 2 co-planar, -1 / 0 failed; 4 = no consensus (op_num < 3, :198-205) */
    int32_t n_support;      /* op_num */
    int32_t n_iterations;   /* RANSAC_STAT.nIterationRansac = min(rst, nIterations) (:226) */
    int32_t best;           /* rs_ind (0-based): the first hypothesis with the largest consensus */
} pre3_vo_result;
/* vodometry_dr_ye.m:162-236 over ransac_dr_ye.m:48-72.  pset1/pset2: 3 x pnum column-major matched 3-D points (frame 1 / 2).
 * draws[n_hyp][4]: 0-based positions in the match list -- the caller draws them with the reference's own rejection rule
 * (ransac_dr_ye.m:28-46; MATLAB's rand stream is not reproducible) and passes rst = min(700, nchoosek(pnum,4)) of them:
 * the reference evaluates all rst (its for-range is fixed at loop entry).  cnum_out[n_hyp], state_out[n_hyp],
 * inlier_out[pnum] optional. */
PRE3_API int pre3_vo_ransac(int device, int pnum, const double *pset1, const double *pset2, int n_hyp, const int32_t *draws,
                            int32_t *cnum_out, int32_t *state_out, int32_t *inlier_out, pre3_vo_result *res);
/* The same with ransac_dr_ye's own inputs: range images x,y,z (rows x cols column-major) of both frames, SIFT frames
 * frm1 (ldf x K1), frm2 (ldf x K2) (rows 1:2 = pixel column,row; 1-based), match (2 x pnum doubles, 1-based, as
 * siftmatch returns).  The point sets of ransac_dr_ye.m:13-19 are gathered on the device (and returned if asked). */
PRE3_API int pre3_vo_ransac_frames(int device, int rows, int cols, const double *x1, const double *y1, const double *z1,
                                   const double *x2, const double *y2, const double *z2, int ldf, int K1, const double *frm1,
                                   int K2, const double *frm2, int pnum, const double *match, int n_hyp, const int32_t *draws,
                                   double *pset1_out, double *pset2_out, int32_t *cnum_out, int32_t *state_out,
                                   int32_t *inlier_out, pre3_vo_result *res);
/* measurement only: average device time of one RANSAC (inputs resident) */
PRE3_API int pre3_vo_bench(int device, int pnum, const double *pset1, const double *pset2, int n_hyp, const int32_t *draws, int reps,
                           double *ms_per_call);

/* ---- a10: sift/siftmatch.c:83-132,139-250 ------------------------------------------------------- */
/* L1: ND x K1, L2: ND x K2, one descriptor per column (column-major, as mxGetData returns them).
 * pairs_out[2*K1] receives 1-based (k1,k2) doubles in increasing k1 exactly as the MEX writes them
 * (:241-242); score_out[K1] (may be NULL) the best squared distance; *M_out the number of matches.
 * Accumulation follows the class promotion of :61-64 (double, float, int, int); the ratio test is in
 * float (:122).  Ties keep the first index (:110-116). */
PRE3_API int pre3_siftmatch_f64(int device, int ND, int K1, const double *L1, int K2, const double *L2, double thresh,
                                double *pairs_out, double *score_out, int *M_out);
PRE3_API int pre3_siftmatch_f32(int device, int ND, int K1, const float *L1, int K2, const float *L2, double thresh,
                                double *pairs_out, double *score_out, int *M_out);
PRE3_API int pre3_siftmatch_u8(int device, int ND, int K1, const uint8_t *L1, int K2, const uint8_t *L2, double thresh,
                               double *pairs_out, double *score_out, int *M_out);
PRE3_API int pre3_siftmatch_i8(int device, int ND, int K1, const int8_t *L1, int K2, const int8_t *L2, double thresh,
                               double *pairs_out, double *score_out, int *M_out);

/* Sharded matcher (database columns [k2_begin,k2_end) of L2 on this GPU): per query the local best,
 * second best (as double) and the GLOBAL 0-based index of the best, for the all-gather + merge of
 * DESIGN.md section "multi-GPU".  cls: 0 f64, 1 f32, 2 u8, 3 i8.  best/second/arg: host arrays [K1]. */
PRE3_API int pre3_siftmatch_partial(int device, int cls, int ND, int K1, const void *L1, int K2_local, const void *L2_local,
                                    int k2_offset, double *best, double *second, int32_t *arg);
/* merge G shards' partials (best[g*K1+k1] ...) and apply the ratio test; same outputs as pre3_siftmatch_* */
PRE3_API int pre3_siftmatch_merge(int cls, int G, int K1, const double *best, const double *second, const int32_t *arg,
                                  double thresh, double *pairs_out, double *score_out, int *M_out);

/* Device-resident database shard of the sharded matcher (uint8 class: BASELINE.json configs[3]).  The queries L1 (replicated on every
 * rank) and this rank's database slice L2_local = columns [k2_offset, k2_offset + K2_local) of the whole L2 are packed once into HBM.
 * pre3_match_shard_run leaves the slice's per-query partials on the DEVICE as double[3][K1] = best | second | global arg (arg < 0: none)
 * and returns their address: the caller all-gathers them with RCCL (device memory, no host staging) into double[G][3][K1] and hands that
 * DEVICE buffer to pre3_match_shard_merge, which merges (ties -> lowest index), applies Lowe's test in float (siftmatch.c:122) and
 * compacts the matches in increasing k1 on the device; only the M pairs cross PCIe.  Results are identical to pre3_siftmatch_u8 on the
 * whole database for any number of shards. */
typedef struct pre3_match_shard pre3_match_shard;
PRE3_API int pre3_match_shard_create(pre3_match_shard **out, int device, int ND, int K1, const uint8_t *L1, int K2_local, const uint8_t *L2_local, int k2_offset);
/* the same for any class the matrix cores serve: cls 0 double (what matching_sift_based.m:104-118 passes), 1 float, 2 uint8.  double / float
 * need ND <= 128 and K1 * K2_local >= 65536 (else PRE3_E_ARG: use pre3_siftmatch_partial); the ratio test is siftmatch.c:122's for every class */
PRE3_API int pre3_match_shard_create_cls(pre3_match_shard **out, int device, int cls, int ND, int K1, const void *L1, int K2_local, const void *L2_local, int k2_offset);
PRE3_API int pre3_match_shard_run(pre3_match_shard *s, void **partial_dev, int *n_doubles);
PRE3_API int pre3_match_shard_merge(pre3_match_shard *s, int G, const void *gathered_dev, double thresh, double *pairs_out, double *score_out, int *M_out);
PRE3_API int pre3_match_shard_destroy(pre3_match_shard *s);

/* ---- the RCCL communicator: collectives enqueued by the library on its own stream ------------------------------------------------------
 * SURVEY section 8(e) / BASELINE.json north_star ("RCCL all-reduce of inlier counts over xGMI"): the two stages that shard exchange data once
 * per round.  With a communicator attached, the library itself enqueues that collective between its kernels, on the stream they run on:
 *   pre3_ransac_sharded    H*P / H*P*H' of the slice's measurements, scoring of hypotheses [lo, hi) of this rank, ncclAllReduce (int32 sum, in
 *                          place, supports + inlier bitmasks in ONE call), the selection kernel (ransac_hypotheses.m:41-62 replayed on the
 *                          reduced buffers) -- one host wait (the pinned mailbox) per round;
 *   pre3_match_shard_match distance kernel on the resident slice, ncclAllGather of the per-query partials, merge + Lowe's test + compaction,
 *                          result block written to pinned host memory by the merge kernel -- one host wait per match.
 * One process per GPU.  Rank 0 makes the 128-byte id (pre3_comm_unique_id) and the host program hands it to the other ranks by any means
 * (3pre_amd/comm.py: torch.distributed broadcast; INTEGRATION.md: MPI_Bcast / a file); every rank then calls pre3_comm_create (collective:
 * it returns when all `world` ranks have called it).  world == 1 is allowed (the collectives run, over one rank).
 * RCCL is bound at run time: librccl.so.1 as already loaded in the process (e.g. PyTorch's copy), else from the loader path, else
 * /opt/rocm/lib; the environment's PRE3_RCCL_LIB overrides.  Without it these calls return PRE3_E_COMM; nothing else in libpre3 needs it.
 * A collective whose peer has died never completes: the wait of pre3_ransac_sharded / pre3_match_shard_match polls ncclCommGetAsyncError,
 * aborts the communicator when it reports one and returns PRE3_E_COMM.  A peer that merely stalls -- or never entered the collective -- is
 * caught by a wall-clock deadline on every host wait that has a collective in front of it (pre3_comm_set_timeout; default 10 s): on expiry
 * the communicator is aborted (ncclCommAbort: the collective the stream is stuck in returns), marked broken, and the call returns
 * PRE3_E_COMM; the context (or shard) stays usable once a fresh communicator is attached.  No wait of the library ends in an unbounded
 * synchronisation behind a collective: with a communicator attached EVERY drain of the context's stream -- pre3_sync, pre3_set_state, the
 * staging blocks of pre3_set_scan / pre3_set_descriptors / map management, pre3_set_comm, pre3_destroy -- polls the stream against the same
 * deadline (round 6).  The abort runs on a thread that pre3_set_comm, pre3_comm_destroy, pre3_destroy and pre3_match_shard_destroy join (bounded:
 * 10 s or the deadline, whichever is longer) before the handle or any buffer the collective touches goes away; while the stream has not
 * drained after an abort, calls that need it idle keep returning PRE3_E_COMM.  A rank whose own part of a round fails BEFORE the collective (a bad table, a failed launch) still
 * enters it, with an empty slice and a "missing" word that travels with the data: every rank then returns PRE3_E_COMM for that round. */
#define PRE3_COMM_ID_BYTES 128
typedef struct pre3_comm pre3_comm;
PRE3_API int pre3_comm_unique_id(void *id_out /* PRE3_COMM_ID_BYTES */);
PRE3_API int pre3_comm_create(pre3_comm **out, int device, const void *id, int rank, int world);
PRE3_API int pre3_comm_destroy(pre3_comm *comm);
PRE3_API int pre3_comm_info(pre3_comm *comm, int *rank, int *world, int *rccl_version, char *lib_path, int lib_path_len);
PRE3_API int pre3_comm_set_timeout(pre3_comm *comm, int milliseconds);      /* deadline of the host waits behind this communicator's collectives (default 10000) */
/* attach: the context / the shard borrows `comm` (must be on the same device; NULL detaches).  pre3_comm_init = create + attach, owned by the
 * context and destroyed with it. */
PRE3_API int pre3_set_comm(pre3_ctx *ctx, pre3_comm *comm);
PRE3_API int pre3_comm_init(pre3_ctx *ctx, const void *id, int rank, int world);
PRE3_API int pre3_match_shard_set_comm(pre3_match_shard *s, pre3_comm *comm);
/* pre3_ransac with the hypotheses dealt to the communicator's ranks (contiguous slices, sizes differing by at most one): same arguments, same
 * outputs on every rank, bit-identical to pre3_ransac for any number of ranks.  Every rank must hold the same state, measurements and draws. */
PRE3_API int pre3_ransac_sharded(pre3_ctx *ctx, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit,
                                 int32_t *support, int32_t *li_mask, int32_t stats[4]);
/* one whole sharded match (run + all-gather + merge); outputs as pre3_match_shard_merge.  Without a communicator: the slice alone (G = 1). */
PRE3_API int pre3_match_shard_match(pre3_match_shard *s, double thresh, double *pairs_out, double *score_out, int *M_out);

/* ---- a11: kNearestNeighbors.m:29-39 ------------------------------------------------------------- */
/* data: N x D, query: M x D, MATLAB column-major.  ids_out (M x k, column-major, 1-based doubles),
 * dist_out (M x k, Euclidean).  Ties: lowest index first (MATLAB's stable sort). */
/* the stateless matcher / kNN calls keep their device scratch in a small pool between calls, pre3_predict_dense its context; this frees the idle part */
PRE3_API int pre3_release_scratch(void);
PRE3_API int pre3_knn_f64(int device, int D, int N, const double *data, int M, const double *query, int k,
                          double *ids_out, double *dist_out);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------- */
/* HIP-event timing on the ctx stream: the timed region of bench.py and the per-launch duration of
 * the covariance down-date kernel (K9) are measured with these, not with torch events.  An event's record is a barrier packet with a
 * completion signal in the stream: the launch behind it starts ~6 us late, so a bracket costs ~12 us of stream time -- time one launch in N. */
PRE3_API int pre3_timer_start(pre3_ctx *ctx);
PRE3_API int pre3_timer_stop(pre3_ctx *ctx, double *ms_out);             /* synchronises */
PRE3_API int pre3_kernel_timing(pre3_ctx *ctx, int enable);             /* 1: bracket every K9 launch of >= 128 rows (the matrix-bound ones) with events; N > 1: one such launch in N; 0: off */
PRE3_API int pre3_kernel_timing_read(pre3_ctx *ctx, int *launches_out, double *total_ms_out, double *flops_out,
                                     double *bytes_out);                /* synchronises, then resets */
/* Of the launches pre3_kernel_timing_read reports: how many were launches of the persistent factorisation that carry the down-date inside
 * (PRE3_OPT_K9_OVERLAP: the bracket then spans update.m:32-38 -- factorisation, solve, x-update and P - W'W -- of an update of the predicted
 * state), and the factorisation + solve flops (r^3/3 + n r^2) those launches executed besides the SYRK count.  Call it BEFORE
 * pre3_kernel_timing_read (both reset their sums). */
PRE3_API int pre3_kernel_timing_info(pre3_ctx *ctx, int *fused_launches_out, double *fact_flops_out);
/* run only the K9 down-date P <- P - W'W with a synthetic W of r rows `reps` times (roofline probe) */
PRE3_API int pre3_bench_downdate(pre3_ctx *ctx, int r, int reps, double *ms_per_launch_out);
/* Matcher probe (bench.py's `matcher` object, tests/test_gpu_match_rank.py): descriptors uploaded and packed ONCE, then `reps` back-to-back
 * launches of the distance + best/second-best stage bracketed by HIP events -- the kernel time of siftmatch.c:91-129's loop without the
 * per-call upload, packing and result copy of pre3_siftmatch_*.  cls: 0 double, 1 float, 2 uint8 (pre3_match_bench_create: uint8).
 * info[3] = { route taken (0 exact kernels, 1 int8 matrix cores, 2 bf16 rank + exact re-evaluation), queries the ranked route had to scan
 * in full, candidates it listed over all queries }; fetch: per query best / second-best squared distance and 0-based argument,
 * as siftmatch.c:110-116 leaves them.  Handles are independent of any pre3_ctx; NULL on failure (pre3_last_error). */
PRE3_API void *pre3_match_bench_create(int device, int ND, int K1, const uint8_t *L1, int K2, const uint8_t *L2);
PRE3_API void *pre3_match_bench_create_cls(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2);
PRE3_API int pre3_match_bench_info(void *h, int32_t info[3]);
PRE3_API int pre3_match_bench_run(void *h, int reps, double *ms_per_launch_out);
PRE3_API int pre3_match_bench_fetch(void *h, double *best, double *second, int32_t *arg);
PRE3_API void pre3_match_bench_destroy(void *h);

#ifdef __cplusplus
}
#endif
#endif /* PRE3_H */
