// pre3_internal.h -- shared host-side declarations of libpre3.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pre3.h"

namespace pre3 {

void set_error(const char *fmt, ...);

#define PRE3_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pre3::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PRE3_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define PRE3_CHECK(cond, status, ...)                                                          \
    do {                                                                                       \
        if (!(cond)) { pre3::set_error(__VA_ARGS__); return (status); }                        \
    } while (0)

#define PRE3_TRY(expr)                                                                         \
    do { int rc_ = (expr); if (rc_ != PRE3_OK) return rc_; } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

constexpr int TILE = 128;       // K9 output tile; P's leading dimension is a multiple of this
constexpr int NB = 64;          // Cholesky / triangular-solve panel width
constexpr int ELLW = 16;        // ELL width of measurement rows (7 pose + 6 landmark = 13 used)
constexpr int MAXK = 4;         // landmarks per RANSAC hypothesis (reference uses 3 or 1)

// one term of an ELL row product (k_ell_G and everything that restates its sums entry by entry: the scorer, the LI gather): an explicit fma,
// so that every kernel holding that sum rounds it the same way
#ifdef __HIPCC__
__device__ __forceinline__ float ell_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double ell_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif

// flags of the per-landmark table
struct LmBuffers {
    int32_t *type = nullptr, *off = nullptr;          // [N]
    double *h = nullptr;                              // [2N]
    int32_t *has_h = nullptr;                         // [N]
    double *Hc = nullptr, *Hl = nullptr;              // [14N], [12N]
    double *S = nullptr;                              // [4N]
    int32_t *has_S = nullptr;                         // [N]
    double *z = nullptr;                              // [2N]
    int32_t *ic = nullptr, *li = nullptr, *hi = nullptr;   // [N]
};

struct KernelTiming {
    bool enabled = false;
    int every = 1, seen = 0;        // bracket one K9 launch out of `every` (an event pair costs ~11 us of stream time)
    std::vector<hipEvent_t> ev;     // pairs
    int used = 0;
    double flops = 0, bytes = 0;
    // launches of the persistent factorisation that carry the down-date (pre3_cholp.hip) are bracketed instead of the K9 launch they replace:
    // `flops` keeps the SYRK count of those updates, `fact_flops` the factorisation + solve the same launch executes (r^3/3 + n r^2)
    int fused = 0; double fact_flops = 0; bool pending = false;     // pending: a bracketed launch whose row count the host does not know yet
};

}  // namespace pre3

struct pre3_ctx {
    int device = 0, dtype = PRE3_F32;
    size_t esz = 4;
    hipStream_t stream = nullptr;
    int capN = 0, capn = 0, capm = 0, caph = 0;
    int N = 0, n = 0;
    int ld = 0;                 // leading dimension of P (multiple of TILE), rows/cols >= n are zero
    int ldw = 0;                // leading dimension of the row workspace W/HP: ld + NB (column `ld` carries nu)
    int rcap = 0;               // max update rows (2*capm rounded to NB)
    pre3_cam cam{};
    bool have_cam = false;
    // device state
    double *x_kk = nullptr, *x_km1 = nullptr;     // [capn]
    void *P = nullptr;                            // [ld*ld] T
    void *tiles = nullptr;                        // int2[n_tiles]: (I,J) of every 64x64 upper-triangle tile, XCD-aware order
    int n_tiles = 0;
    void *tiles_flat = nullptr;                   // int2[n_tiles]: the 8 lists interleaved (block b -> list b % 8) for the one-tile kernel
    unsigned int *tile_ctr = nullptr;             // device ticket counter of the persistent K9 grid
    int *tile_cnt = nullptr; int tiles_stride = 0;  // per-list lengths [8] and list stride
    bool tile_ctr_clean = false;                  // the 8 counters are zero (k_update_x resets them ahead of K9)
    int num_cus = 256;
    void *Wp = nullptr;                           // fp32 path: bf16 planes of W in stage-image order (k_split_w), ld x rcap x 6 B
    bool k9_b3 = false;                           // fp32: K9 as three-way bf16 split on the bf16 matrix cores (PRE3_K9_B3=0 turns it off)
    int split_rows = 0;                           // rows of W whose bf16 planes the factorisation launches have already produced
    void *Sp = nullptr;                           // fp32 path: bf16 planes of the factorisation's S blocks (pending updates), rcap/64 x rcap/64 x 24 KB
    void *tiles128 = nullptr; int n_tiles128 = 0; // int2[n_tiles128]: 128x128 upper-triangle tiles of k_downdate_b3, XCD-interleaved
    unsigned int *chol_arrive = nullptr; unsigned int chol_target = 0;   // [0] panel arrivals, [1] scoring done, [2] rescue done, [3],[4] rider producers
    unsigned int ride_target[2] = { 0, 0 };
    void *comm = nullptr; bool comm_owned = false;                 // RCCL communicator (pre3_comm.hip): pre3_comm_init / pre3_set_comm
    bool leave_jn_to_predict = false, jn_pending = false;          // the deferred HI update's rows/cols 3..6 <- Jn pass rides in the next k_predict
    bool defer_hi = false, hi_pending = false; int last_n_hi = 0;   // PRE3_OPT_DEFER_HI (pre3_set_option)
    bool ride_rescue_projection = false;          // request: the next K9 launch also projects at x_k_k (pre3_step sets it before the LI update)
    bool rescue_projected = false;                // h / H at the current x_k_k are on the device (set by that launch, consumed by pre3_rescue)
    //   // panel kernels: arrivals of the workgroups that read the raw diagonal block
    int p_which = -1;                             // which estimate P currently holds (-1: none)
    bool x_valid[2] = {false, false};
    pre3::LmBuffers lm;
    // measurements (host mirrors kept: m and the index list are host-known)
    int m = 0;
    std::vector<int32_t> meas_host;
    int32_t *meas = nullptr;                      // [capm] landmark index per measurement
    // measurement-row workspace
    int32_t *row_col = nullptr;                   // [rcap*ELLW]
    void *row_val = nullptr;                      // [rcap*ELLW] T
    double *row_nu = nullptr;                     // [rcap]  z-h per row
    void *HP = nullptr;                           // [rcap*ldw] T   H*P for all measured rows (RANSAC)
    void *G = nullptr;                            // [rcap*rcap] T  H*P*H' (no +R)
    void *W = nullptr;                            // [rcap*ldw] T   update workspace: HP rows -> W = L^-1 HP
    void *Smat = nullptr;                         // [rcap*rcap] T  S -> L
    void *Rdense = nullptr;                       // [rcap*rcap] T  optional dense R (stateless update)
    int32_t *sel_rows = nullptr;                  // [rcap] compacted row list (indices into the measured rows)
    // RANSAC
    int32_t *hyp = nullptr;                       // [caph*MAXK]
    int32_t *support = nullptr;                   // [caph]
    uint32_t *masks = nullptr;                    // [caph*mask_words_cap]
    int mask_words_cap = 0;
    int scored_n_draw = 0, scored_k = 0;          // the round pre3_ransac_score / pre3_ransac laid the support + mask buffer out for
    int32_t *stats = nullptr;                     // [16] device: best, iters, n_hyp, max_support, n_li, n_hi, status
    int32_t *li_meas = nullptr, *hi_meas = nullptr;   // [capm] flags in measurement order
    double *pred_params = nullptr;                // [64] predict: Qq1(16) Jn(16) Q(49->7x7) etc.
    int32_t *pinned_stats = nullptr;              // host pinned [16]
    // mailbox: pinned, host-coherent; the selection and HI-collection stages store their counts here and bump a
    // sequence word, so the host learns r for the next launches by polling instead of a copy + stream sync
    int32_t *mail_host = nullptr, *mail_dev = nullptr;
    int32_t seq_select = 0, seq_collect = 0;
    bool g_valid = false;                         // c->G holds H*P*H' of all measured rows (PRE3_INLINE_G=0); otherwise the scorer and the LI gather compute their entries
    // pre3_step: the selection stage of the RANSAC round is not launched by pre3_ransac but rides in the LI gather's launch (k_select_gather)
    bool defer_select = false, select_pending = false;
    int sel_n_draw = 0, sel_k = 0, sel_early_exit = 0;
    int li_from_host = -1, hi_from_host = -1;     // row counts forced through pre3_set_flags (-1: use the kernels' counts)
    bool li_kernel = false, hi_kernel = false;    // a select / collect kernel has run for the current measurement set
    // per-step inbox: [meas | ic | hyp | z] contiguous on the device, mirrored in pinned host memory -> ONE H2D copy
    void *inbox_dev = nullptr; unsigned char *inbox_host = nullptr; void *inbox_host_dev = nullptr;   // pinned + device-mapped: the device address of inbox_host
    size_t inbox_bytes = 0, off_meas = 0, off_ic = 0, off_hyp = 0, off_z = 0, off_flags = 0, flags_bytes = 0;
    bool ride_innovation = false;                 // pre3_step: the S_i pass goes out with the next k_ell_HP_build launch instead of its own
    int32_t seq_inbox = 0;                        // sequence number of the last inbox pull (published by the kernel in mailbox word 10)
    bool inbox_pending = false;
    // map management (allocated on first use)
    void *P_alt = nullptr; double *x_alt = nullptr; int32_t *map_col = nullptr; void *map_val = nullptr; int32_t *map_desc = nullptr; int32_t *map_src0 = nullptr; double *map_conv = nullptr;
    double *map_feat = nullptr; int32_t *map_flags = nullptr;
    void *map_stage[2] = { nullptr, nullptr }; hipEvent_t map_stage_ev[2] = { nullptr, nullptr }; bool map_stage_used[2] = { false, false };
    int map_stage_next = 0; size_t map_stage_bytes = 0;   // pinned staging blocks of the map operations (pre3_map.hip)
    std::vector<int32_t> lm_type_host;
    // IC search (matching_sift_based.m): landmark descriptor bank [capN][128], the current scan's SIFT set, match scratch
    double *bank = nullptr, *bank_alt = nullptr; bool bank_set = false;
    double *scan_desc = nullptr, *scan_pos = nullptr; int scan_K2 = 0, scan_cap = 0;
    // pinned staging of host arrays on their way to the device (the per-frame scan, descriptors of new landmarks): two blocks used alternately,
    // an event each; the caller's data is copied (and bounds-checked) into a block, the device pulls it over PCIe on the context's stream --
    // no synchronisation, no read-back (pre3_api.hip: stage_acquire / stage_release)
    void *up_stage[2] = { nullptr, nullptr }; hipEvent_t up_stage_ev[2] = { nullptr, nullptr }; bool up_stage_used[2] = { false, false };
    int up_stage_next = 0; size_t up_stage_bytes = 0;
    int32_t stage_seq[4] = { 0, 0, 0, 0 };          // sequence number of the last pull of staging block k (0, 1: uploads; 2, 3: map management)
    void *ic_result_host = nullptr, *ic_result_host_dev = nullptr; int32_t seq_ic = 0;    // the IC search's result block in mapped pinned memory: written by the device, announced through mailbox word 12
    int32_t *ic_pred = nullptr, *ic_counts = nullptr, *ic_arg = nullptr, *ic_pairs = nullptr, *ic_newk2 = nullptr; double *ic_best = nullptr, *ic_second = nullptr;
    int32_t *bank_src = nullptr;
    int ic_route = 0;                             // the last pre3_ic_search's matcher: 0 exact tiled kernel, 1 ranked (matrix cores), 2 fused small-problem route
    bool ic_last_ranked = false;                  // the last pre3_ic_search matched on the matrix cores (PRE3_OPT_IC_RANKED)
    void *ic_rank = nullptr; bool bank_ok = false;   // the scan packed for the matrix-core matcher (pre3_match.hip: IcRank); every bank descriptor inside its bounds
    size_t ic_pcap = 0;                           // elements in each of ic_pb / ic_ps / ic_pa
    double *ic_pb = nullptr, *ic_ps = nullptr; int32_t *ic_pa = nullptr;     // per (column tile, landmark) partials of the tiled matcher [scan_cap/64][capN]
    // timing
    hipEvent_t t0 = nullptr, t1 = nullptr;
    pre3::KernelTiming kt;
    bool measurements_set = false, projected = false, innovated = false;
    int32_t *need = nullptr; int need_tag = 0;    // [capm] sharded RANSAC: need[s] == need_tag where this rank's hypothesis slice draws measurement s
    // persistent factorisation (pre3_cholp.hip)
    unsigned int *cholp_flags = nullptr; void *cholp_tp = nullptr; unsigned int cholp_epoch = 0; bool chol_persist = true;
    bool cholp_counted = false;                   // this context is in pre3_cholp.hip's per-device count
    bool cholp_done = false;                      // the speculative launch of an LI update has already factored and solved
    // down-date consumers inside the persistent factorisation (pre3_cholp.hip): group records, all tiles in group order, first tile per group
    int32_t *dd_groups = nullptr; int dd_n_groups = 0; void *dd_tiles = nullptr; std::vector<int> dd_tile_off;
    bool k9_overlap = true;                       // PRE3_OPT_K9_OVERLAP
    bool shard_round = false;                     // the last RANSAC round was pre3_ransac_sharded (its selection publishes the missing-slice word in mail[11])
    bool hi_fused = false;                        // the rescue stage's collection + HI update went out as k_hi_fused (pre3_step): pre3_update_hi only has the count to read
    bool x_done = false;                          // ... and its strips have computed x_k_k = x_prior + W'(L^-1 nu) as well (update.m:36,42,48)
    float *jn_q = nullptr; bool jn_q_valid = false;   // un-normalised rows 3..6 of P, left by the persistent launch's consumers for the gate that rides with the Jnorm pass (GateRide)
    bool want_gate_ride = false, rescue_gated = false; double rescue_chi2 = 0.0;
    bool proj_in_cholp = false;                   // the persistent launch's strips have projected every landmark at x_k_k (strip_proj_body): the gate needs no projection
    bool proj_with_jnorm = false;                 // the rescue's projection rides in the next k_jnorm_P launch (no K9 launch to carry it)
    int dd_done = 0;                              // groups the last k_cholp launch has down-dated (consumed by the next launch_downdate)
    bool hp_all_valid = false;                    // HP / G hold H*P, H*P*H' of ALL measured rows at the current prior (ransac_prepare)
    // the rescue stage + HI update inside the persistent launch (pre3_cholp.hip, CpTail): per landmark the planes of y = H J W' and the row H J;
    // crit's published list
    void *tail_yp = nullptr; float *tail_hb = nullptr; int32_t *tail_hib = nullptr; float *tail_wt = nullptr;
    // PRE3_OPT_PEND_HI (PendW, pre3_geomdev.h): k_hi_fused's W~ goes to its own buffers and the down-date is not launched; pend_rows > 0 from the moment the host
    // has the HI count (pre3_update_hi) until k_cholp's consumers have taken the panels (launch_cholp) or pend_flush() has run k_downdate_b3 on them
    bool pend_opt = false; bool hi_pend_launched = false; bool pend_keep = false; int pend_rows = 0;
    unsigned long long *hf_sx = nullptr;          // k_hi_fused, two panels: S as [128][128] (sequence number, value) pairs, dealt over the workgroups (hf_S_dealt)
    float *W_pend = nullptr; void *Wp_pend = nullptr; unsigned *hf_xy = nullptr;      // hf_xy: [0] k_hi_fused's "L^-1 nu is out" word, [64 .. 191] L^-1 nu (hf_x_update)
    bool step_tail = false;                       // PRE3_OPT_STEP_TAIL (default: the environment's PRE3_TAIL, else off)
    bool tail_want = false; double tail_chi2 = 0;  // pre3_step asks the LI update's launch to carry the tail
    bool tail_launched = false;                   // the last launch_cholp carried it
    bool tail_done = false;                       // ... and it ran (the LI update had rows): P holds P - W'W - W~'W~ with ONE pending rows/cols 3..6 pass (params[96..])
};

namespace pre3 {

// normJac.m:27-38
__device__ inline void d_normjac(const double *q, double *J)
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    double s = pow(r * r + x * x + y * y + z * z, -1.5);
    J[0] = s * (x * x + y * y + z * z); J[1] = s * (-r * x); J[2] = s * (-r * y); J[3] = s * (-r * z);
    J[4] = s * (-x * r); J[5] = s * (r * r + y * y + z * z); J[6] = s * (-x * y); J[7] = s * (-x * z);
    J[8] = s * (-y * r); J[9] = s * (-y * x); J[10] = s * (r * r + x * x + z * z); J[11] = s * (-y * z);
    J[12] = s * (-z * r); J[13] = s * (-z * x); J[14] = s * (-z * y); J[15] = s * (r * r + x * x + y * y);
}

// ---- pooled device scratch of the stateless entry points (pre3_match.hip)
int scratch_acquire(size_t bytes, void **p_out, int *slot_out);
void scratch_release(int slot, void *p);

// ---- RCCL communicator (pre3_comm.hip): collectives on the caller's stream
int comm_all_reduce_i32(void *comm, void *buf, size_t count, hipStream_t st);                       /* in place, sum */
int comm_all_gather_f64(void *comm, const void *src, void *dst, size_t count_per_rank, hipStream_t st);
int comm_timeout_ms(void *comm);                                                                   /* deadline of a host wait behind a collective */
int comm_give_up(void *comm, const char *what);                                                     /* deadline passed: abort, mark broken, PRE3_E_COMM */
int comm_poll_error(void *comm);
int comm_abort_ms(void *comm);                                                                       /* bound of a join of the abort thread (>= 10 s) */
bool comm_broken(void *comm);                                                                       /* the communicator has been aborted (deadline or asynchronous error) */
bool comm_abort_wait(void *comm, int ms);                                                           /* an abort started by comm_give_up has returned (joined) within ms; true also when none is running */
/* Wait until everything queued on a stream has finished.  Without a communicator: hipStreamSynchronize.  With one (a collective may be queued on the
   stream, and a peer may never enter it): hipStreamQuery polls with the communicator's wall-clock deadline; on expiry the communicator is aborted and the
   call returns PRE3_E_COMM -- no host wait of the library ends in an unbounded synchronisation behind a collective. */
int stream_drain_on(hipStream_t st, void *comm, const char *what);
int stream_drain(pre3_ctx *c, const char *what);                                                                    /* PRE3_E_COMM once the communicator has failed (and is aborted) */
void comm_rank_world(void *comm, int *rank, int *world);
int comm_device(void *comm);

// ---- IC search (pre3_match.hip)
int launch_ic_search(pre3_ctx *c, double thresh, int strict);
bool ic_search_fused_applies(const pre3_ctx *c);      /* N * K2 <= 2^20 pairs, N <= 4096, K2 <= 2048: the two-launch route */
int launch_ic_search_fused(pre3_ctx *c, double thresh, int strict, int32_t seq, int slot, bool matched);      /* matched: the matcher rode in the projection's launch */
int launch_bank_gather(pre3_ctx *c, int N_new, const int32_t *src_host);
int ic_rank_set_scan(pre3_ctx *c, bool in_bounds);
int launch_pull(pre3_ctx *c, const void *pinned_host, void *dst_dev, size_t bytes, int done_slot = -1);      // done_slot >= 0: the pull announces itself in mailbox word 16 + slot (stage_wait)
int stage_wait(pre3_ctx *c, int k);      // pinned host block -> device, by a kernel (pre3_api.hip)
void ic_rank_free(pre3_ctx *c);
constexpr int DESC_DIM = 128;

// ---- geometry / RANSAC kernels (pre3_geom.hip)
int launch_project(pre3_ctx *c, int which, int clear_first);
int launch_innovation(pre3_ctx *c, int mode /*0: S=HPH'+I for predicted; 1: rescue gate + HI list*/, double chi2, bool clear_flags = false, bool collect = true /* mode 1: the HI list follows (k_collect_hi) */);
struct IcMatchRide;
int launch_project_innovation(pre3_ctx *c, int which, int clear_first, int mode, double chi2, bool collect = true, bool clear_ic = false, const IcMatchRide *ride = nullptr);
int launch_update_x(pre3_ctx *c, int which_prior, int r);
int launch_jnorm(pre3_ctx *c, int which);

int run_hypothesis_support(int n, const double *xi, const pre3_cam &cam, int n_id, const int32_t *i1, const int32_t *i2, const int32_t *i3,
                           const double *z_id, int n_euc, const int32_t *i4, const double *z_euc, double threshold, int32_t *out_host /* [1 + n_id + n_euc] */);

// ---- dense update kernels (pre3_update.hip)
// rows: ELL rows [r] in c->row_col/row_val with nu in c->row_nu; computes W = H*P (+ nu column),
// S = H*P*H' + R, Cholesky, W = L^-1 [HP | nu], x += W' y, P -= W'W, Jnorm + normalise.
int run_update(pre3_ctx *c, int which_prior, int r, bool dense_R, void *Kt_out_dev /*nullable, r_pad x ldw T*/, bool prebuilt = false, bool first_done = false,
               bool hp_built = false /* rows and W = H*P are in place (launch_ell_HP_build_sel), S is not */);
int launch_ell_HP_build_sel(pre3_ctx *c, int nsel, const int32_t *sel_dev /* nullable: the first nsel measurements */, void *dst);
int launch_chol_first_spec(pre3_ctx *c, int nsel_max);
struct InboxRide;
int launch_ell_HP_build(pre3_ctx *c, void *dst, const int32_t *need = nullptr, int need_tag = 0 /* sharded RANSAC: only the measurements with need[s] == need_tag */,
                        const InboxRide *ib = nullptr /* a pull from the pinned inbox that rides in the launch */);
int launch_ell_G_hyp(pre3_ctx *c, int k, int lo, int hi, int ldg);                 /* H*P*H' entries among each hypothesis' own rows, hypotheses [lo, hi) */
int launch_gather_li(pre3_ctx *c, int nsel /* < 0: count read on the device */, int nsel_max, const int32_t *sel_dev, int ldg);
int launch_select_gather(pre3_ctx *c, int n_draw, int k, int early_exit, int mask_words);     // selection + LI gather in one launch (pre3_geom.hip)
bool select_gather_usable(const pre3_ctx *c);
int launch_ell_HP(pre3_ctx *c, int r, void *dst /*r_pad x ldw*/, bool with_nu);
int launch_ell_G(pre3_ctx *c, int r, const void *HPsrc, void *dst, int ldg, int add_identity, const void *Rdense, bool lower_only = false);
int launch_downdate(pre3_ctx *c, int r, const void *W, int which_prior = -1 /* >= 0: also x <- x_prior + W'y (update.m:36) */);
bool hi_fused_usable(const pre3_ctx *c);
int hi_fused_max(const pre3_ctx *c);                  /* landmarks k_hi_fused updates with on its own (64: two panels; 32) */
int launch_hi_fused(pre3_ctx *c, int32_t seq);        /* the HI collection + update of up to 32 landmarks without the host (k_hi_fused + its down-date) */

}  // namespace pre3
