// pre3_api.hip -- the C ABI of include/pre3.h: context management, host<->device marshalling, and the
// stage order of one filter step (mono_slam.m:153-187).  No compute happens on the host.
#include <time.h>
#include <algorithm>
#include <immintrin.h>
#include <cmath>
#include <cstdarg>
#include <new>

#include <chrono>
#include "pre3_internal.h"
#include "pre3_test_hooks.h"
#include "pre3_geomdev.h"
#include <mutex>
#include "pre3_cholp.h"

namespace pre3 {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// launchers defined in the kernel files
int launch_predict_impl(pre3_ctx *c, const double u[7], bool with_projection = false, size_t inbox_n16 = 0, int32_t inbox_seq = 0);
IcMatchRide ic_match_ride(const pre3_ctx *c);
int launch_inbox_pull(pre3_ctx *c, const void *src_host_mapped, void *dst_dev, size_t n16, int32_t seq, int slot = 10, int32_t *clear = nullptr, int n_clear = 0);
int launch_slice_prepare(pre3_ctx *c, const void *src_host_mapped, size_t n16, int32_t seq, int n_zero, int k, int lo, int hi, int tag);   // (either direction: 16-byte words between device memory and a mapped pinned block, then `seq` into mailbox word `slot`)
int launch_window_gate(pre3_ctx *c, int M, const int32_t *pred_idx_dev, const int32_t *k1_dev, const double *zc_dev, int strict, int32_t *accept_dev);
int launch_build_rows_impl(pre3_ctx *c, int nsel, const int32_t *sel_dev, int r_pad);
int launch_ransac_score_impl(pre3_ctx *c, int k, double threshold, int hyp_begin, int hyp_end, int ldg, int32_t *support_dev, uint32_t *mask_dev, int mask_words,
                             int select_n_draw = 0, int early_exit = 0);
int launch_ransac_select_impl(pre3_ctx *c, int n_draw, int k, int early_exit, int32_t *support_dev, const uint32_t *mask_dev, int mask_words, int err_idx = -1);
int launch_fill_w(pre3_ctx *c, int r_pad);
void release_scratch();
int match_partial(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2, int k2_offset, double *best, double *second, int32_t *arg);
int knn_run(int device, int D, int N, const double *data, int M, const double *query, int k, double *ids, double *dist);
void *match_bench_create(int cls, int ND, int K1, const void *L1, int K2, const void *L2);
int match_bench_info(void *h, int32_t info[3]);
int match_bench_run(void *h, int reps, double *ms_per);
int match_bench_fetch(void *h, double *best, double *second, int32_t *arg);
void match_bench_destroy(void *h);
void *match_shard_create(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2, int k2_offset);
int match_shard_run(void *h, void **partial_dev, int *n_doubles);
int match_shard_merge(void *h, int G, const void *gathered_dev, double thresh, double *pairs_out, double *score_out, int *M_out);
void match_shard_destroy(void *h);
int match_shard_set_comm(void *h, void *comm);
int match_shard_match(void *h, double thresh, double *pairs_out, double *score_out, int *M_out);
int match_shard_test_stall(void *h, int release);

template <typename T> static int dmalloc(T **p, size_t count)
{
    void *q = nullptr;
    if (hipMalloc(&q, sizeof(T) * (count ? count : 1)) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", sizeof(T) * count); return PRE3_E_NOMEM; }
    // defined contents: a recycled allocation must not leak a previous context's data into fields the reference leaves empty
    // (a measured landmark that is not predicted has h = [] there; its h/H were whatever the allocator returned here)
    if (hipMemset(q, 0, sizeof(T) * (count ? count : 1)) != hipSuccess) { (void)hipFree(q); set_error("hipMemset failed"); return PRE3_E_HIP; }
    *p = (T *)q;
    return PRE3_OK;
}
static int dmalloc_bytes(void **p, size_t bytes)
{
    if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", bytes); return PRE3_E_NOMEM; }
    if (hipMemset(*p, 0, bytes ? bytes : 16) != hipSuccess) { (void)hipFree(*p); *p = nullptr; set_error("hipMemset failed"); return PRE3_E_HIP; }
    return PRE3_OK;
}

static int check_ctx(pre3_ctx *c)
{
    PRE3_CHECK(c != nullptr, PRE3_E_ARG, "null context");
    PRE3_HIP(hipSetDevice(c->device));
    if (c->hi_pending) {                 // PRE3_OPT_DEFER_HI: the previous step's HI update is completed by whatever call comes next
        c->hi_pending = false;
        PRE3_TRY(pre3_update_hi(c));
        c->last_n_hi = c->hi_from_host >= 0 ? c->hi_from_host : (c->hi_kernel ? c->mail_host[5] : 0);
    }
    // PRE3_OPT_PEND_HI: only pre3_step carries a pending HI down-date into its launches (pend_keep); for everybody else P is P before the call goes on
    if (c->pend_rows > 0 && !c->pend_keep) PRE3_TRY(pend_flush(c));
    return PRE3_OK;
}

// Poll the pinned mailbox until the kernel that was launched with sequence number `seq` has published.
// slot 8: k_ransac_select, slot 9: k_collect_hi.  Falls back to a stream sync if the word does not arrive.
static double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
// see pre3_internal.h
int stream_drain_on(hipStream_t st, void *comm, const char *what)
{
    if (comm == nullptr) { PRE3_HIP(hipStreamSynchronize(st)); return PRE3_OK; }
    const bool was_broken = comm_broken(comm);
    const double t0 = now_ms();
    for (long spin = 0; ; ++spin) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return PRE3_OK;
        if (e != hipErrorNotReady) PRE3_HIP(e);
        if ((spin & 15) == 15) {
            if (!was_broken) PRE3_TRY(comm_poll_error(comm));
            if (now_ms() - t0 > comm_timeout_ms(comm)) {
                if (!was_broken) return comm_give_up(comm, what);
                set_error("%s: the stream has not drained %d ms after its communicator was aborted", what, comm_timeout_ms(comm));
                return PRE3_E_COMM;
            }
            timespec ts = { 0, 20000 };
            nanosleep(&ts, nullptr);
        }
    }
}
int stream_drain(pre3_ctx *c, const char *what) { return stream_drain_on(c->stream, c->comm, what); }
static int wait_mail(pre3_ctx *c, int slot, int32_t seq)
{
    volatile int32_t *w = c->mail_host + slot;
    // With a communicator on the context the awaited kernel may sit behind a collective: a peer that stalls (or never entered) must not hang this
    // host for ever -- the wait has a wall-clock deadline (pre3_comm_set_timeout), and never ends in a bare hipStreamSynchronize.
    const bool coll = c->comm != nullptr;
    const double t0 = coll ? now_ms() : 0;
    static const int query_env = getenv("PRE3_WAIT_QUERY") ? atoi(getenv("PRE3_WAIT_QUERY")) : 0;      // 1: hipStreamQuery from the first 1024 spins on (rounds 1-4)
    double t_q = 0;
    for (long spin = 0; coll || spin < 2000000000L; ++spin) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == seq) return PRE3_OK;
        if ((spin & 1023) == 1023) {
            // Is the stream idle although the word has not come (the producing kernel was never launched)?  hipStreamQuery is not free of side effects:
            // to learn the state of the last launch the runtime appends a marker (a barrier packet with a completion signal) behind it, and the next
            // launch then starts ~6 us after its predecessor has ended -- the two "holes" of the step (behind k_cholp and behind the HI down-date: the
            // two places where the host polls a count) were these markers, not store drains.  So: only after 2 ms without the word.
            if (!query_env) {
                const double t = now_ms();
                if (t_q == 0) { t_q = t; continue; }
                if (t - t_q < 2.0) continue;
            }
            if (hipStreamQuery(c->stream) == hipSuccess) break;
            if (coll && (spin & 0x3ffff) == 0x3ffff) {
                PRE3_TRY(comm_poll_error(c->comm));     // a collective in front of the awaited kernel whose peer died
                if (now_ms() - t0 > comm_timeout_ms(c->comm)) return comm_give_up(c->comm, "a collective in front of the awaited kernel");
            }
        }
    }
    PRE3_TRY(stream_drain(c, __func__));
    PRE3_CHECK(__atomic_load_n(w, __ATOMIC_ACQUIRE) == seq, PRE3_E_STATE, "mailbox: the producing kernel has not been launched");
    return PRE3_OK;
}

static int fetch_stats(pre3_ctx *c)
{
    PRE3_HIP(hipMemcpyAsync(c->pinned_stats, c->stats, sizeof(int32_t) * 16, hipMemcpyDeviceToHost, c->stream));
    PRE3_TRY(stream_drain(c, __func__));
    PRE3_CHECK(c->pinned_stats[7] == 0, PRE3_E_HIP, "a device-side wait on another workgroup gave up (counter never arrived): results are invalid");
    PRE3_CHECK(c->pinned_stats[6] == 0, PRE3_E_NUMERIC, "innovation covariance S is not positive definite");
    return PRE3_OK;
}

// the staged scan [descriptors | positions] out of pinned host memory into the two device arrays (n16_pos == 0: one array)
// done.ctr != nullptr: the last workgroup to finish publishes done.seq in the pinned mailbox word done.mail -- the host may then overwrite the staging
// block (stage_wait).  An event recorded behind the launch would tell the same, but its record puts a barrier packet with a completion signal into the
// stream, and the NEXT launch then starts ~6 us after this one has ended (measured in the frame leg: three such holes per frame).
struct StageDone { unsigned int *ctr; int32_t *mail; int32_t seq; };
__global__ __launch_bounds__(256) void k_scan_pull(const int4 *__restrict__ src, int n16_desc, int4 *__restrict__ desc, int n16_pos, int4 *__restrict__ pos, StageDone done)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16_desc) desc[i] = src[i];
    else if (i < n16_desc + n16_pos) pos[i - n16_desc] = src[i];
    if (done.ctr != nullptr) {
        __syncthreads();                                 // (this workgroup's reads of the block have returned: their values are on their way out)
        if (threadIdx.x == 0 && atomicAdd(done.ctr, 1u) == gridDim.x - 1) {
            atomicExch(done.ctr, 0u);
            __threadfence_system();
            __hip_atomic_store(done.mail, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// the staging block `k` (0, 1: uploads; 2, 3: map management) has been pulled: one read of host memory (after 2 ms: a stream synchronisation)
int stage_wait(pre3_ctx *c, int k)
{
    volatile int32_t *w = c->mail_host + 16 + k;
    const int32_t seq = c->stage_seq[k];
    if (seq == 0) return PRE3_OK;
    double t0 = 0;
    for (long spin = 0; ; ++spin) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == seq) return PRE3_OK;
        if ((spin & 1023) == 1023) {
            const double t = now_ms();
            if (t0 == 0) t0 = t;
            else if (t - t0 > 2.0) break;
        }
    }
    PRE3_TRY(stream_drain(c, __func__));
    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seq) {
        c->stage_seq[k] = 0;                         // (the stream is idle: the block is free whatever became of that launch; the next pull starts a fresh count)
        (void)hipMemsetAsync(c->chol_arrive + 8 + k, 0, sizeof(unsigned int), c->stream);
        set_error("staging block %d: its pull has not run", k);
        return PRE3_E_STATE;
    }
    return PRE3_OK;
}
static StageDone stage_done(pre3_ctx *c, int k) { return StageDone{ c->chol_arrive + 8 + k, c->mail_dev + 16 + k, ++c->stage_seq[k] }; }

// bytes of a pinned (device-mapped) host block -> device memory, read over PCIe by the device itself on the context's stream: no DMA engine,
// whose copies start with ~10 us of latency each (and, once in a few hundred calls, with tens of milliseconds inside the runtime)
int launch_pull(pre3_ctx *c, const void *pinned_host, void *dst_dev, size_t bytes, int done_slot)
{
    void *src_dev = nullptr;
    PRE3_HIP(hipHostGetDevicePointer(&src_dev, const_cast<void *>(pinned_host), 0));
    const int n16 = (int)((bytes + 15) / 16);
    if (n16 == 0) return PRE3_OK;
    hipLaunchKernelGGL(k_scan_pull, dim3(ceil_div(n16, 256)), dim3(256), 0, c->stream, (const int4 *)src_dev, n16, (int4 *)dst_dev, 0, (int4 *)nullptr,
                       done_slot >= 0 ? stage_done(c, done_slot) : StageDone{ nullptr, nullptr, 0 });
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

}  // namespace pre3

using namespace pre3;

extern "C" {

const char *pre3_last_error(void) { return pre3::g_err; }
const char *pre3_version(void) { return "pre3-mi355x 0.1 (gfx950)"; }

int pre3_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The tail of the persistent launch (rescue stage + HI update inside k_cholp, CpTail; off by default): per landmark y = H J W' as bf16 planes (+ one zero
// slot), the row H J, crit's list, and W once more column-major.  Allocated on demand (pre3_set_option(PRE3_OPT_STEP_TAIL, 1) or PRE3_TAIL=1 at creation).
// PRE3_OPT_PEND_HI's buffers: W~ of up to two panels (f32 rows) and its planes in Wp's layout (same nst_total, so that the consumers' offsets hold for both)
static int pend_alloc(pre3_ctx *c)
{
    if (c->W_pend && c->Wp_pend && c->hf_xy) return PRE3_OK;
    PRE3_CHECK(c->dtype == PRE3_F32 && c->Wp != nullptr && c->rcap >= 2 * NB, PRE3_E_STATE, "PRE3_OPT_PEND_HI: fp32 contexts with the persistent factorisation only");
    int rc = PRE3_OK;
    auto A = [&](int r) { if (rc == PRE3_OK) rc = r; };
    if (!c->W_pend) { void *f = nullptr; A(dmalloc_bytes(&f, (size_t)2 * NB * c->ldw * sizeof(float))); c->W_pend = (float *)f; }
    if (!c->Wp_pend) A(dmalloc_bytes(&c->Wp_pend, (size_t)ceil_div(c->ldw, 128) * (c->rcap / 16) * (3 * 4 * 64) * 16));
    if (!c->hf_xy) { void *f = nullptr; A(dmalloc_bytes(&f, sizeof(unsigned) * 256)); c->hf_xy = (unsigned *)f; }
    return rc;
}

static int tail_alloc(pre3_ctx *c)
{
    if (c->tail_yp && c->tail_hb && c->tail_hib && c->tail_wt) return PRE3_OK;
    PRE3_CHECK(c->dtype == PRE3_F32 && c->Wp != nullptr, PRE3_E_STATE, "PRE3_OPT_STEP_TAIL: fp32 contexts with the persistent factorisation only");
    int rc = PRE3_OK;
    auto A = [&](int r) { if (rc == PRE3_OK) rc = r; };
    if (!c->tail_yp) A(dmalloc_bytes(&c->tail_yp, (size_t)(c->capN + 1) * 2 * 3 * c->rcap * 2));
    if (!c->tail_hb) { void *f = nullptr; A(dmalloc_bytes(&f, (size_t)c->capN * 2 * 16 * sizeof(float))); c->tail_hb = (float *)f; }
    if (!c->tail_hib) { void *f = nullptr; A(dmalloc_bytes(&f, sizeof(int32_t) * (64 + 64 * 16))); c->tail_hib = (int32_t *)f; }
    if (!c->tail_wt) { void *f = nullptr; A(dmalloc_bytes(&f, (size_t)c->ldw * c->rcap * sizeof(float))); c->tail_wt = (float *)f; }
    return rc;
}

int pre3_create(pre3_ctx **out, int device, int dtype, int max_landmarks, int max_hyp)
{
    PRE3_CHECK(out != nullptr, PRE3_E_ARG, "pre3_create: null output pointer");
    *out = nullptr;
    PRE3_CHECK(dtype == PRE3_F64 || dtype == PRE3_F32, PRE3_E_ARG, "pre3_create: dtype must be PRE3_F64 or PRE3_F32");
    PRE3_CHECK(max_landmarks >= 1 && max_hyp >= 1, PRE3_E_ARG, "pre3_create: capacities must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (libpre3 has no CPU fallback)"); return PRE3_E_NODEVICE; }
    PRE3_CHECK(device >= 0 && device < ndev, PRE3_E_NODEVICE, "pre3_create: device %d out of range (have %d)", device, ndev);
    PRE3_HIP(hipSetDevice(device));
    pre3_ctx *c = new (std::nothrow) pre3_ctx();
    PRE3_CHECK(c != nullptr, PRE3_E_NOMEM, "out of host memory");
    c->device = device; c->dtype = dtype; c->esz = dtype == PRE3_F64 ? 8 : 4;
    c->capN = max_landmarks; c->capn = 13 + 6 * max_landmarks; c->capm = max_landmarks; c->caph = max_hyp;
    c->ld = round_up(c->capn, TILE); c->ldw = c->ld + NB;
    { const char *e = getenv("PRE3_TAIL"); c->step_tail = e ? atoi(e) != 0 : false; }      // PRE3_OPT_STEP_TAIL: off by default (measured: DESIGN.md section 5d)
    c->rcap = round_up(2 * c->capm, NB) + NB;       // (+ one panel: the rescued landmarks' rows follow the LI update's inside the persistent launch, pre3_cholp.hip)
    c->mask_words_cap = ceil_div(c->capm, 32);
    int rc = PRE3_OK;
    auto A = [&](int r) { if (rc == PRE3_OK) rc = r; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return PRE3_E_HIP; }
    A(dmalloc(&c->x_kk, c->capn)); A(dmalloc(&c->x_km1, c->capn));
    A(dmalloc_bytes(&c->P, (size_t)c->ld * c->ld * c->esz));
    A(dmalloc(&c->lm.type, c->capN)); A(dmalloc(&c->lm.off, c->capN));
    A(dmalloc(&c->lm.h, 2 * (size_t)c->capN)); A(dmalloc(&c->lm.has_h, c->capN));
    A(dmalloc(&c->lm.Hc, 14 * (size_t)c->capN)); A(dmalloc(&c->lm.Hl, 12 * (size_t)c->capN));
    A(dmalloc(&c->lm.S, 4 * (size_t)c->capN)); A(dmalloc(&c->lm.has_S, c->capN));
    {
        // inbox: [meas | ic | hyp | z] is what the one H2D copy of a step ships (kept under 16 KB at N=500: larger copies
        // leave the runtime's shader path for the SDMA path, +20 us); the inlier flags [li | hi | li_meas | hi_meas] follow
        // in the same allocation and are cleared on the device (k_project_innovation in a step, a memset otherwise)
        c->off_meas = 0;
        c->off_ic = c->off_meas + sizeof(int32_t) * c->capm;
        c->off_hyp = (c->off_ic + sizeof(int32_t) * c->capN + 15) / 16 * 16;
        c->off_z = (c->off_hyp + sizeof(int32_t) * (size_t)c->caph * MAXK + 15) / 16 * 16;
        c->off_flags = c->off_z + sizeof(double) * 2 * c->capN;
        const size_t off_li = c->off_flags, off_hi = off_li + sizeof(int32_t) * c->capN;
        const size_t off_lim = off_hi + sizeof(int32_t) * c->capN, off_him = off_lim + sizeof(int32_t) * c->capm;
        c->flags_bytes = sizeof(int32_t) * (2 * (size_t)c->capN + 2 * (size_t)c->capm);
        c->inbox_bytes = c->off_flags + c->flags_bytes;
        A(dmalloc_bytes(&c->inbox_dev, c->inbox_bytes));
        if (rc == PRE3_OK && hipHostMalloc((void **)&c->inbox_host, c->inbox_bytes, hipHostMallocMapped) != hipSuccess) { set_error("hipHostMalloc failed"); rc = PRE3_E_NOMEM; }
        if (rc == PRE3_OK && hipHostGetDevicePointer((void **)&c->inbox_host_dev, c->inbox_host, 0) != hipSuccess) { set_error("hipHostGetDevicePointer failed"); rc = PRE3_E_HIP; }
        if (rc == PRE3_OK) {
            memset(c->inbox_host, 0, c->inbox_bytes);
            unsigned char *d = (unsigned char *)c->inbox_dev;
            c->meas = (int32_t *)(d + c->off_meas); c->lm.ic = (int32_t *)(d + c->off_ic);
            c->lm.li = (int32_t *)(d + off_li); c->lm.hi = (int32_t *)(d + off_hi); c->li_meas = (int32_t *)(d + off_lim); c->hi_meas = (int32_t *)(d + off_him);
            c->hyp = (int32_t *)(d + c->off_hyp); c->lm.z = (double *)(d + c->off_z);
        }
    }
    A(dmalloc(&c->row_col, (size_t)c->rcap * ELLW)); A(dmalloc_bytes(&c->row_val, (size_t)c->rcap * ELLW * c->esz));
    A(dmalloc(&c->row_nu, c->rcap));
    A(dmalloc_bytes(&c->HP, (size_t)c->rcap * c->ldw * c->esz));
    A(dmalloc_bytes(&c->W, (size_t)c->rcap * c->ldw * c->esz));
    A(dmalloc_bytes(&c->G, (size_t)c->rcap * c->rcap * c->esz));
    A(dmalloc_bytes(&c->Smat, (size_t)c->rcap * c->rcap * c->esz));
    A(dmalloc(&c->sel_rows, c->rcap)); A(dmalloc(&c->need, c->capm));
    if (rc == PRE3_OK) (void)hipMemset(c->need, 0, sizeof(int32_t) * c->capm);
    // supports and inlier masks in ONE allocation, the masks right behind the n_draw supports of the current round (ransac_prepare sets
    // c->masks): a sharded round all-reduces both with a single collective over one contiguous range
    A(dmalloc(&c->support, (size_t)c->caph + 4 + (size_t)c->caph * c->mask_words_cap));
    if (rc == PRE3_OK) c->masks = reinterpret_cast<uint32_t *>(c->support + c->caph + 4);
    A(dmalloc(&c->stats, 16));
    A(dmalloc(&c->pred_params, 128));
    {
        // K9 tile schedule: upper-triangle 64x64 tiles in 4x4 super-tile order, so that the tiles in flight at
        // any moment (tickets are handed out in this order) share W panels in L2 / Infinity Cache.
        const int nt = c->ld / 64, ns = ceil_div(nt, 4);
        std::vector<std::vector<int2>> lists(8);
        int sidx = 0;
        for (int SI = 0; SI < ns; ++SI)
            for (int SJ = SI; SJ < ns; ++SJ) {
                // whole super-tiles go to the currently shortest list (balanced to within one super-tile)
                int best = 0;
                for (int x = 1; x < 8; ++x) if (lists[x].size() < lists[best].size()) best = x;
                (void)sidx;
                for (int i = SI * 4; i < std::min(nt, SI * 4 + 4); ++i)
                    for (int j = SJ * 4; j < std::min(nt, SJ * 4 + 4); ++j)
                        if (j >= i) lists[best].push_back(make_int2(i, j));
            }
        for (;;) {       // even the lists out to within one tile (moves come from the tail: the last, partial super-tiles)
            int lo = 0, hi = 0;
            for (int x = 1; x < 8; ++x) { if (lists[x].size() < lists[lo].size()) lo = x; if (lists[x].size() > lists[hi].size()) hi = x; }
            if (lists[hi].size() <= lists[lo].size() + 1) break;
            lists[lo].push_back(lists[hi].back());
            lists[hi].pop_back();
        }
        size_t mx = 1;
        int cnts[8];
        for (int x = 0; x < 8; ++x) { mx = std::max(mx, lists[x].size()); cnts[x] = (int)lists[x].size(); }
        std::vector<int2> flat(mx * 8, make_int2(0, 0));
        int total = 0;
        for (int x = 0; x < 8; ++x) { for (size_t k2 = 0; k2 < lists[x].size(); ++k2) flat[x * mx + k2] = lists[x][k2]; total += cnts[x]; }
        c->tiles_stride = (int)mx;
        c->n_tiles = total;
        {
            std::vector<int2> inter;                  // interleave: block b takes the next tile of list b % 8
            size_t pos[8] = { 0 };
            while ((int)inter.size() < total)
                for (int x = 0; x < 8 && (int)inter.size() < total; ++x)
                    if (pos[x] < lists[x].size()) inter.push_back(lists[x][pos[x]++]);
            A(dmalloc_bytes(&c->tiles_flat, sizeof(int2) * (inter.size() ? inter.size() : 1)));
            if (rc == PRE3_OK && hipMemcpy(c->tiles_flat, inter.data(), sizeof(int2) * inter.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("tile table upload failed"); rc = PRE3_E_HIP; }
        }
        A(dmalloc(&c->tile_ctr, 8)); A(dmalloc(&c->tile_cnt, 8)); A(dmalloc(&c->chol_arrive, 16));      // ([8..11]: arrival counters of the staging pulls)
        if (rc == PRE3_OK) (void)hipMemset(c->chol_arrive, 0, sizeof(unsigned int) * 16);
        if (rc == PRE3_OK) { (void)hipMemset(c->tile_ctr, 0, sizeof(unsigned int) * 8); (void)hipMemcpy(c->tile_cnt, cnts, sizeof(cnts), hipMemcpyHostToDevice); }
        { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) c->num_cus = pr.multiProcessorCount; }
        A(dmalloc_bytes(&c->tiles, sizeof(int2) * flat.size()));
        if (rc == PRE3_OK && hipMemcpy(c->tiles, flat.data(), sizeof(int2) * flat.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("tile table upload failed"); rc = PRE3_E_HIP; }
    }
    if (dtype == PRE3_F32) {
        { const char *e = getenv("PRE3_K9_B3"); c->k9_b3 = e ? atoi(e) != 0 : true; }
        // k_downdate_b3: bf16 planes of W and the 128x128 tile list (4x4 super-tiles dealt to 8 lists, lists interleaved: block b runs
        // on XCD b % 8, so a super-tile's 8 column blocks of planes stay in one L2)
        A(dmalloc_bytes(&c->Wp, (size_t)(c->ld + 128) * c->rcap * 6));       // + one column block for the nu strip's planes
        A(dmalloc_bytes(&c->Sp, (size_t)(c->rcap / NB) * (c->rcap / NB) * 1536 * 16));
        // persistent factorisation (pre3_cholp.hip): flag words (zero: every launch brings its own epoch), one plane block per row for the hand-over to crit
        { void *f = nullptr; A(dmalloc_bytes(&f, cholp_flag_bytes(c->ldw / 32))); c->cholp_flags = (unsigned int *)f; }
        {   // the down-date consumers' schedule (pre3_cholp.hip)
            std::vector<int32_t> rec; std::vector<int2> t64;
            dd_build_groups(c->ld / 64, rec, t64, c->dd_tile_off);
            c->dd_n_groups = (int)c->dd_tile_off.size() - 1;
            { void *f = nullptr; A(dmalloc_bytes(&f, sizeof(int32_t) * rec.size())); c->dd_groups = (int32_t *)f; }
            A(dmalloc_bytes(&c->dd_tiles, sizeof(int2) * t64.size()));
            if (rc == PRE3_OK && (hipMemcpy(c->dd_groups, rec.data(), sizeof(int32_t) * rec.size(), hipMemcpyHostToDevice) != hipSuccess ||
                                  hipMemcpy(c->dd_tiles, t64.data(), sizeof(int2) * t64.size(), hipMemcpyHostToDevice) != hipSuccess)) { set_error("consumer table upload failed"); rc = PRE3_E_HIP; }
        }
        A(dmalloc_bytes(&c->cholp_tp, (size_t)(c->rcap / NB) * 1536 * 16));
        { void *f = nullptr; A(dmalloc_bytes(&f, sizeof(float) * 4 * (size_t)c->ld)); c->jn_q = (float *)f; }
        { void *f = nullptr; A(dmalloc_bytes(&f, sizeof(unsigned long long) * (2 * NB) * (2 * NB))); c->hf_sx = (unsigned long long *)f; }      // rows 3..6 of P before the Jnorm pass (GateRide, pre3_geom.hip)
        // (the buffers of the in-launch tail, PRE3_OPT_STEP_TAIL, are allocated when the option is switched on: tail_alloc -- at N = 2000 they are
        //  ~300 MB that the default path never touches)
        if (c->step_tail) A(tail_alloc(c));
        { const char *e = getenv("PRE3_PEND_HI"); if (e && atoi(e) != 0 && c->rcap >= 2 * NB && c->rcap / NB <= CP_MAX_NRB + 1) { A(pend_alloc(c)); c->pend_opt = rc == PRE3_OK; } }      // PRE3_OPT_PEND_HI
        if (rc == PRE3_OK) { cholp_context_count(c->device, +1); c->cholp_counted = true; }
        const int nt = c->ld / 128, ns = ceil_div(nt, 4);
        std::vector<std::vector<int2>> lists(8);
        for (int SI = 0; SI < ns; ++SI)
            for (int SJ = SI; SJ < ns; ++SJ) {
                int best = 0;
                for (int x = 1; x < 8; ++x) if (lists[x].size() < lists[best].size()) best = x;
                for (int i = SI * 4; i < std::min(nt, SI * 4 + 4); ++i)
                    for (int j = SJ * 4; j < std::min(nt, SJ * 4 + 4); ++j)
                        if (j >= i) lists[best].push_back(make_int2(i, j));
            }
        for (;;) {
            int lo = 0, hi = 0;
            for (int x = 1; x < 8; ++x) { if (lists[x].size() < lists[lo].size()) lo = x; if (lists[x].size() > lists[hi].size()) hi = x; }
            if (lists[hi].size() <= lists[lo].size() + 1) break;
            lists[lo].push_back(lists[hi].back());
            lists[hi].pop_back();
        }
        // whole rounds of large tiles; the left-over tiles (diagonal ones first: a quarter of each is below the diagonal) as 64x64
        std::vector<int2> inter;
        const int total = nt * (nt + 1) / 2;
        size_t pos[8] = { 0 };
        while ((int)inter.size() < total)
            for (int x = 0; x < 8 && (int)inter.size() < total; ++x)
                if (pos[x] < lists[x].size()) inter.push_back(lists[x][pos[x]++]);
        const int n_big = total <= c->num_cus ? total : total / c->num_cus * c->num_cus;
        int n_small_src = total - n_big;
        std::vector<int2> big, small;
        for (int pass = 0; pass < 2; ++pass)                         // pass 0: pick diagonal tiles for splitting, from the back of the order
            for (int k2 = (int)inter.size() - 1; k2 >= 0 && n_small_src > 0; --k2) {
                int2 &t = inter[k2];
                if (t.x < 0 || (pass == 0 && t.x != t.y)) continue;
                for (int a2 = 0; a2 < 2; ++a2)
                    for (int b2 = 0; b2 < 2; ++b2)
                        if (t.x != t.y || b2 >= a2) small.push_back(make_int2(2 * t.x + a2, 2 * t.y + b2));
                t.x = -1; --n_small_src;
            }
        for (const int2 &t : inter) if (t.x >= 0) big.push_back(make_int2((2 * t.x) | (1 << 16), 2 * t.y));
        inter = big;
        inter.insert(inter.end(), small.begin(), small.end());
        c->n_tiles128 = (int)inter.size();
        A(dmalloc_bytes(&c->tiles128, sizeof(int2) * inter.size()));
        if (rc == PRE3_OK && hipMemcpy(c->tiles128, inter.data(), sizeof(int2) * inter.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("tile table upload failed"); rc = PRE3_E_HIP; }
    }
    if (rc == PRE3_OK && hipHostMalloc((void **)&c->pinned_stats, sizeof(int32_t) * 16) != hipSuccess) { set_error("hipHostMalloc failed"); rc = PRE3_E_NOMEM; }
    if (rc == PRE3_OK && hipHostMalloc((void **)&c->mail_host, sizeof(int32_t) * 32, hipHostMallocMapped) != hipSuccess) { set_error("hipHostMalloc failed"); rc = PRE3_E_NOMEM; }
    if (rc == PRE3_OK) {
        memset(c->mail_host, 0, sizeof(int32_t) * 32);      // (words 16..19: the staging blocks' "pulled" sequence numbers, stage_wait)
        if (hipHostGetDevicePointer((void **)&c->mail_dev, c->mail_host, 0) != hipSuccess) { set_error("hipHostGetDevicePointer failed"); rc = PRE3_E_HIP; }
    }
    if (rc == PRE3_OK && (hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess)) { set_error("hipEventCreate failed"); rc = PRE3_E_HIP; }
    if (rc != PRE3_OK) { pre3_destroy(c); return rc; }
    (void)hipMemsetAsync(c->stats, 0, sizeof(int32_t) * 16, c->stream);
    (void)hipMemsetAsync(c->P, 0, (size_t)c->ld * c->ld * c->esz, c->stream);
    (void)stream_drain(c, __func__);
    *out = c;
    return PRE3_OK;
}

int pre3_destroy(pre3_ctx *c)
{
    if (!c) return PRE3_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)stream_drain(c, __func__);
    // an abort started by a deadline may still be inside RCCL, and the collective it unblocks may still touch this context's buffers: joined (within the
    // deadline) and the stream drained again before anything is freed
    if (c->comm && !comm_abort_wait(c->comm, comm_abort_ms(c->comm))) set_error("pre3_destroy: the communicator's abort has not returned: buffers are freed under it");
    else if (c->comm && comm_broken(c->comm) && c->stream) (void)stream_drain(c, __func__);
    if (c->cholp_counted) { cholp_context_count(c->device, -1); c->cholp_counted = false; }
    ic_rank_free(c);
    if (c->comm && c->comm_owned) (void)pre3_comm_destroy((pre3_comm *)c->comm);
    c->comm = nullptr;
    void *bufs[] = { c->x_kk, c->x_km1, c->P, c->lm.type, c->lm.off, c->lm.h, c->lm.has_h, c->lm.Hc, c->lm.Hl, c->lm.S, c->lm.has_S,
                     c->inbox_dev, c->row_col, c->row_val, c->row_nu, c->HP, c->W, c->G, c->Smat, c->Rdense,
                     c->sel_rows, c->support, c->stats, c->pred_params, c->tiles, c->tile_ctr, c->tile_cnt, c->tiles_flat, c->P_alt, c->x_alt, c->map_col, c->map_val, c->map_desc, c->map_src0, c->map_conv, c->map_feat, c->map_flags, c->bank, c->bank_alt, c->scan_desc, c->scan_pos, c->ic_pred, c->ic_counts, c->ic_arg, c->ic_newk2, c->ic_best, c->ic_second, c->bank_src, c->chol_arrive, c->ic_pb, c->ic_ps, c->ic_pa, c->Wp, c->Sp, c->tiles128, c->need, c->cholp_flags, c->cholp_tp, c->dd_groups, c->dd_tiles, c->tail_yp, c->tail_hb, c->tail_hib, c->tail_wt, c->jn_q, c->W_pend, c->Wp_pend, c->hf_xy, c->hf_sx };
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (int k2 = 0; k2 < 2; ++k2) { if (c->map_stage[k2]) (void)hipHostFree(c->map_stage[k2]); if (c->map_stage_ev[k2]) (void)hipEventDestroy(c->map_stage_ev[k2]); }
    for (int k2 = 0; k2 < 2; ++k2) { if (c->up_stage[k2]) (void)hipHostFree(c->up_stage[k2]); if (c->up_stage_ev[k2]) (void)hipEventDestroy(c->up_stage_ev[k2]); }
    if (c->pinned_stats) (void)hipHostFree(c->pinned_stats);
    if (c->inbox_host) (void)hipHostFree(c->inbox_host);
    if (c->mail_host) (void)hipHostFree(c->mail_host);
    if (c->ic_result_host) (void)hipHostFree(c->ic_result_host);
    for (hipEvent_t e : c->kt.ev) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PRE3_OK;
}

int pre3_set_option(pre3_ctx *c, int option, int value)
{
    PRE3_TRY(check_ctx(c));
    switch (option) {
    case PRE3_OPT_DEFER_HI: c->defer_hi = value != 0; return PRE3_OK;
    case PRE3_OPT_K9_BF16X3: c->k9_b3 = value != 0 && c->dtype == PRE3_F32 && c->Wp != nullptr; return PRE3_OK;
    case PRE3_OPT_CHOL_PERSIST: c->chol_persist = value != 0; return PRE3_OK;
    case PRE3_OPT_K9_OVERLAP: c->k9_overlap = value != 0; return PRE3_OK;
    case PRE3_OPT_STEP_TAIL:
        // (fp64 contexts have no persistent launch: the option is accepted and stays without effect, pre3_get_option reports 0)
        if (value != 0 && c->dtype == PRE3_F32 && c->Wp != nullptr) { const int rc = tail_alloc(c); if (rc != PRE3_OK) { c->step_tail = false; return rc; } }
        c->step_tail = value != 0;
        return PRE3_OK;
    case PRE3_OPT_PEND_HI:
        // (contexts that cannot take the persistent launch at their capacity -- fp64, more than 16 panels of rows -- accept the option and stay without it: the
        //  pending rows would only ever be flushed, and their planes mirror Wp's size)
        if (value != 0 && c->dtype == PRE3_F32 && c->Wp != nullptr && c->rcap / NB <= CP_MAX_NRB + 1) { const int rc = pend_alloc(c); if (rc != PRE3_OK) { c->pend_opt = false; return rc; } }
        c->pend_opt = value != 0 && c->W_pend != nullptr;
        return PRE3_OK;
    default: set_error("pre3_set_option: unknown option %d", option); return PRE3_E_ARG;
    }
}

int pre3_get_option(pre3_ctx *c, int option, int *value_out)
{
    PRE3_CHECK(c != nullptr && value_out != nullptr, PRE3_E_ARG, "pre3_get_option: null argument");
    switch (option) {
    case PRE3_OPT_DEFER_HI: *value_out = c->defer_hi ? 1 : 0; return PRE3_OK;
    case PRE3_OPT_K9_BF16X3: *value_out = c->k9_b3 ? 1 : 0; return PRE3_OK;
    case PRE3_OPT_CHOL_PERSIST: *value_out = cholp_usable(c, 1) ? 1 : 0; return PRE3_OK;
    case PRE3_OPT_IC_RANKED: *value_out = c->ic_last_ranked ? 1 : 0; return PRE3_OK;
    case PRE3_OPT_IC_ROUTE: *value_out = c->ic_route; return PRE3_OK;
    case PRE3_OPT_K9_OVERLAP: *value_out = c->k9_overlap ? 1 : 0; return PRE3_OK;
    case PRE3_OPT_STEP_TAIL: *value_out = (c->step_tail && c->tail_yp != nullptr) ? 1 : 0; return PRE3_OK;
    case PRE3_OPT_PEND_HI: *value_out = c->pend_opt ? 1 : 0; return PRE3_OK;
    default: set_error("pre3_get_option: unknown option %d", option); return PRE3_E_ARG;
    }
}

int pre3_sync(pre3_ctx *c)
{
    PRE3_TRY(check_ctx(c));
    PRE3_TRY(stream_drain(c, __func__));
    return PRE3_OK;
}

int pre3_set_cam(pre3_ctx *c, const pre3_cam *cam)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(cam != nullptr && cam->f > 0, PRE3_E_ARG, "pre3_set_cam: invalid camera");
    c->cam = *cam; c->have_cam = true;
    return PRE3_OK;
}

int pre3_set_map(pre3_ctx *c, int N, const int32_t *lm_type)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(N >= 0 && N <= c->capN, PRE3_E_ARG, "pre3_set_map: N=%d exceeds capacity %d", N, c->capN);
    PRE3_CHECK(N == 0 || lm_type != nullptr, PRE3_E_ARG, "pre3_set_map: null type list");
    std::vector<int32_t> off(N ? N : 1);
    int n = 13;
    for (int i = 0; i < N; ++i) {
        PRE3_CHECK(lm_type[i] == PRE3_INVDEPTH || lm_type[i] == PRE3_CARTESIAN, PRE3_E_ARG, "pre3_set_map: landmark %d has unknown type %d", i, lm_type[i]);
        off[i] = n; n += lm_type[i] == PRE3_INVDEPTH ? 6 : 3;
    }
    PRE3_CHECK(n <= c->capn, PRE3_E_ARG, "pre3_set_map: state size %d exceeds capacity %d", n, c->capn);
    PRE3_TRY(stream_drain(c, __func__));
    if (N) {
        PRE3_HIP(hipMemcpy(c->lm.type, lm_type, sizeof(int32_t) * N, hipMemcpyHostToDevice));
        PRE3_HIP(hipMemcpy(c->lm.off, off.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
    }
    c->N = N; c->n = n;
    c->lm_type_host.assign(lm_type, lm_type + N);
    // the P buffer keeps its capacity-sized leading dimension; entries beyond n stay zero
    PRE3_HIP(hipMemset(c->lm.has_h, 0, sizeof(int32_t) * c->capN)); PRE3_HIP(hipMemset(c->lm.has_S, 0, sizeof(int32_t) * c->capN));
    // update_features_info.m:30-44 empties h, H, S, z: a landmark that is measured but never predicted must read zeros, not a previous map's values
    PRE3_HIP(hipMemset(c->lm.h, 0, sizeof(double) * 2 * c->capN)); PRE3_HIP(hipMemset(c->lm.Hc, 0, sizeof(double) * 14 * c->capN));
    PRE3_HIP(hipMemset(c->lm.Hl, 0, sizeof(double) * 12 * c->capN)); PRE3_HIP(hipMemset(c->lm.S, 0, sizeof(double) * 4 * c->capN));
    PRE3_HIP(hipMemset(c->inbox_dev, 0, c->inbox_bytes)); PRE3_HIP(hipMemset(c->lm.li, 0, sizeof(int32_t) * c->capN));
    PRE3_HIP(hipMemset(c->lm.hi, 0, sizeof(int32_t) * c->capN));
    c->m = 0; c->meas_host.clear(); c->measurements_set = false; c->projected = false; c->innovated = false;
    c->x_valid[0] = c->x_valid[1] = false; c->p_which = -1;
    return PRE3_OK;
}

int pre3_state_size(pre3_ctx *c) { return c ? c->n : PRE3_E_ARG; }

int pre3_set_state(pre3_ctx *c, int which, int n, const double *x, const double *P)
{
    {
        // a fresh state also clears what an earlier state left in the device's error words (a factorisation that was not positive definite, a
        // wait that gave up): the deferred work of the old state is dropped with them
        const int rc0 = check_ctx(c);
        if (c && rc0 != PRE3_OK && rc0 != PRE3_E_NUMERIC && rc0 != PRE3_E_HIP) return rc0;
        if (c) {
            PRE3_HIP(hipSetDevice(c->device));
            (void)stream_drain(c, __func__);
            (void)hipMemsetAsync(c->stats + 6, 0, sizeof(int32_t) * 2, c->stream);
            if (c->mail_host) { c->mail_host[6] = 0; c->mail_host[7] = 0; }
            c->jn_pending = false; c->hi_pending = false; c->tail_done = false; c->hi_fused = false; c->pend_rows = 0; c->hi_pend_launched = false;
        }
    }
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(which == PRE3_X_K_K || which == PRE3_X_K_KM1, PRE3_E_ARG, "pre3_set_state: bad selector");
    PRE3_CHECK(n == c->n, PRE3_E_ARG, "pre3_set_state: n=%d but the map defines n=%d", n, c->n);
    PRE3_CHECK(x && P, PRE3_E_ARG, "pre3_set_state: null pointer");
    PRE3_TRY(stream_drain(c, __func__));
    PRE3_HIP(hipMemcpy(which == PRE3_X_K_K ? c->x_kk : c->x_km1, x, sizeof(double) * n, hipMemcpyHostToDevice));
    const int ld = c->ld;
    PRE3_HIP(hipMemset(c->P, 0, (size_t)ld * ld * c->esz));
    if (c->dtype == PRE3_F64) {
        PRE3_HIP(hipMemcpy2D(c->P, (size_t)ld * 8, P, (size_t)n * 8, (size_t)n * 8, n, hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp((size_t)n * n);
        for (size_t i = 0; i < (size_t)n * n; ++i) tmp[i] = (float)P[i];
        PRE3_HIP(hipMemcpy2D(c->P, (size_t)ld * 4, tmp.data(), (size_t)n * 4, (size_t)n * 4, n, hipMemcpyHostToDevice));
    }
    c->x_valid[which] = true; c->p_which = which; c->hp_all_valid = false;
    return PRE3_OK;
}

int pre3_get_state(pre3_ctx *c, int which, int n, double *x, double *P)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(which == PRE3_X_K_K || which == PRE3_X_K_KM1, PRE3_E_ARG, "pre3_get_state: bad selector");
    PRE3_CHECK(n == c->n, PRE3_E_ARG, "pre3_get_state: n=%d but the map defines n=%d", n, c->n);
    PRE3_CHECK(c->x_valid[which], PRE3_E_STATE, "pre3_get_state: that estimate has not been computed");
    PRE3_TRY(fetch_stats(c));
    if (x) PRE3_HIP(hipMemcpy(x, which == PRE3_X_K_K ? c->x_kk : c->x_km1, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (P) {
        PRE3_CHECK(c->p_which == which, PRE3_E_STATE, "pre3_get_state: the covariance buffer currently holds the other estimate (it is updated in place)");
        const int ld = c->ld;
        if (c->dtype == PRE3_F64) {
            PRE3_HIP(hipMemcpy2D(P, (size_t)n * 8, c->P, (size_t)ld * 8, (size_t)n * 8, n, hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp((size_t)n * n);
            PRE3_HIP(hipMemcpy2D(tmp.data(), (size_t)n * 4, c->P, (size_t)ld * 4, (size_t)n * 4, n, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < (size_t)n * n; ++i) P[i] = (double)tmp[i];
        }
    }
    return PRE3_OK;
}

int pre3_predict(pre3_ctx *c, const double u[7])
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(u != nullptr, PRE3_E_ARG, "pre3_predict: null u");
    PRE3_CHECK(c->x_valid[PRE3_X_K_K] && c->p_which == PRE3_X_K_K, PRE3_E_STATE, "pre3_predict: needs (x_k_k, p_k_k) on the device");
    PRE3_TRY(launch_predict_impl(c, u));
    c->x_valid[PRE3_X_K_KM1] = true; c->p_which = PRE3_X_K_KM1; c->hp_all_valid = false;
    return PRE3_OK;
}

int pre3_project(pre3_ctx *c, int which, int clear_first)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(which == PRE3_X_K_K || which == PRE3_X_K_KM1, PRE3_E_ARG, "pre3_project: bad selector");
    PRE3_CHECK(c->have_cam, PRE3_E_STATE, "pre3_project: camera not set");
    PRE3_CHECK(c->x_valid[which], PRE3_E_STATE, "pre3_project: that estimate is not on the device");
    if (c->N == 0) return PRE3_OK;
    PRE3_TRY(launch_project(c, which, clear_first));
    c->projected = true;
    return PRE3_OK;
}

int pre3_innovation(pre3_ctx *c)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->projected, PRE3_E_STATE, "pre3_innovation: call pre3_project first");
    if (c->N == 0) return PRE3_OK;
    PRE3_TRY(launch_innovation(c, 0, 0.0));
    c->innovated = true;
    return PRE3_OK;
}

int pre3_get_landmark_fields(pre3_ctx *c, double *h, int32_t *has_h, double *Hc, double *Hl, double *S)
{
    PRE3_TRY(check_ctx(c));
    PRE3_TRY(stream_drain(c, __func__));
    int N = c->N;
    if (N == 0) return PRE3_OK;
    if (h) PRE3_HIP(hipMemcpy(h, c->lm.h, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
    if (has_h) PRE3_HIP(hipMemcpy(has_h, c->lm.has_h, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    if (Hc) PRE3_HIP(hipMemcpy(Hc, c->lm.Hc, sizeof(double) * 14 * N, hipMemcpyDeviceToHost));
    if (Hl) PRE3_HIP(hipMemcpy(Hl, c->lm.Hl, sizeof(double) * 12 * N, hipMemcpyDeviceToHost));
    if (S) PRE3_HIP(hipMemcpy(S, c->lm.S, sizeof(double) * 4 * N, hipMemcpyDeviceToHost));
    return PRE3_OK;
}

// Fill the pinned inbox and ship it with ONE async copy: [meas | ic | (hyp) | z].  hyp (n_hyp_ints ints) optional.
// pull == false: the inbox is filled and the context updated, but no pull is launched: the caller's next launch carries it (pre3_step: k_predict);
// *nbytes_out = what that pull must copy
static int install_measurements(pre3_ctx *c, int m, const int32_t *meas_idx, const double *z /* 2m, null: z already on device */,
                                const int32_t *hyp, int n_hyp_ints, bool flags_clear = false, bool pull = true, size_t *nbytes_out = nullptr)
{
    PRE3_CHECK(m >= 0 && m <= c->capm, PRE3_E_ARG, "measurements: m=%d exceeds capacity %d", m, c->capm);
    for (int j = 0; j < m; ++j) {
        PRE3_CHECK(meas_idx[j] >= 0 && meas_idx[j] < c->N, PRE3_E_ARG, "measurements: landmark index %d out of range", meas_idx[j]);
        PRE3_CHECK(j == 0 || meas_idx[j] > meas_idx[j - 1], PRE3_E_ARG, "measurements: landmark indices must be strictly ascending");
    }
    // the previous pull out of the pinned inbox must have completed before the host overwrites it (it has, a whole step ago: the pull
    // kernel publishes its sequence number in mailbox word 10, so this is one read of host memory -- no event, whose record would put
    // a barrier packet into the stream -- and, unlike a stream sync, it does not wait for the kernels queued since)
    if (c->inbox_pending) { PRE3_TRY(wait_mail(c, 10, c->seq_inbox)); c->inbox_pending = false; }
    c->m = m; c->meas_host.assign(meas_idx, meas_idx + m);
    c->select_pending = false;
    int32_t *hm = (int32_t *)(c->inbox_host + c->off_meas), *hic = (int32_t *)(c->inbox_host + c->off_ic);
    double *hz = (double *)(c->inbox_host + c->off_z);
    memset(hic, 0, sizeof(int32_t) * c->N);
    for (int j = 0; j < m; ++j) { hm[j] = meas_idx[j]; hic[meas_idx[j]] = 1; }
    if (z) {
        memset(hz, 0, sizeof(double) * 2 * c->N);
        for (int j = 0; j < m; ++j) { hz[2 * meas_idx[j]] = z[2 * j]; hz[2 * meas_idx[j] + 1] = z[2 * j + 1]; }
    }
    if (hyp) memcpy(c->inbox_host + c->off_hyp, hyp, sizeof(int32_t) * n_hyp_ints);
    // The inbox crosses PCIe by a KERNEL that reads the pinned, device-mapped host buffer (14 KB at N=500): a hipMemcpyAsync between
    // kernels is a blit with barrier packets on both sides and opened two ~10 us holes in the stream around a 3 us copy.
    const size_t nbytes = z ? c->off_z + sizeof(double) * 2 * c->N : c->off_hyp + (hyp ? sizeof(int32_t) * n_hyp_ints : 0);
    // (the inlier flags behind the inbox are cleared by the pull's own workgroup: a hipMemsetAsync is a fill kernel between barrier packets)
    int32_t *const flags = (int32_t *)((unsigned char *)c->inbox_dev + c->off_flags);
    if (pull) { PRE3_TRY(launch_inbox_pull(c, c->inbox_host_dev, c->inbox_dev, (nbytes + 15) / 16, ++c->seq_inbox, 10, flags_clear ? nullptr : flags, flags_clear ? 0 : (int)(c->flags_bytes / 4))); c->inbox_pending = true; }
    if (nbytes_out) *nbytes_out = nbytes;
    if (!flags_clear && !pull) PRE3_HIP(hipMemsetAsync(flags, 0, c->flags_bytes, c->stream));
    c->li_from_host = c->hi_from_host = -1; c->li_kernel = c->hi_kernel = false;
    c->hp_all_valid = false;
    c->measurements_set = true;
    return PRE3_OK;
}

int pre3_set_measurements(pre3_ctx *c, int m, const int32_t *meas_idx, const double *z)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(m == 0 || (meas_idx && z), PRE3_E_ARG, "pre3_set_measurements: null pointer");
    return install_measurements(c, m, meas_idx, z, nullptr, 0);
}

int pre3_window_gate(pre3_ctx *c, int M, const int32_t *k1, const double *zc, int strict_reference, int32_t *accept_out)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->projected, PRE3_E_STATE, "pre3_window_gate: call pre3_project / pre3_innovation first");
    PRE3_CHECK(M >= 0 && (M == 0 || (k1 && zc)), PRE3_E_ARG, "pre3_window_gate: bad arguments");
    int N = c->N;
    std::vector<int32_t> has_h(N ? N : 1), pred;
    PRE3_TRY(stream_drain(c, __func__));
    if (N) PRE3_HIP(hipMemcpy(has_h.data(), c->lm.has_h, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) if (has_h[i]) pred.push_back(i);
    PRE3_CHECK(M <= (int)pred.size(), PRE3_E_ARG, "pre3_window_gate: %d candidates but only %zu predicted landmarks", M, pred.size());
    for (int q = 0; q < M; ++q) PRE3_CHECK(k1[q] >= 0 && k1[q] < (int)pred.size(), PRE3_E_ARG, "pre3_window_gate: k1[%d]=%d out of range", q, k1[q]);
    std::vector<int32_t> acc(M ? M : 1, 0);
    if (M) {
        // temporaries are released on every exit path (an early PRE3_TRY / PRE3_HIP return used to leak them)
        struct Tmp { void *p[4] = { nullptr, nullptr, nullptr, nullptr }; ~Tmp() { for (void *q : p) if (q) (void)hipFree(q); } } tmp;
        int32_t *d_pred = nullptr, *d_k1 = nullptr, *d_acc = nullptr; double *d_zc = nullptr;
        PRE3_TRY(dmalloc(&d_pred, pred.size())); tmp.p[0] = d_pred;
        PRE3_TRY(dmalloc(&d_k1, M)); tmp.p[1] = d_k1;
        PRE3_TRY(dmalloc(&d_acc, M)); tmp.p[2] = d_acc;
        PRE3_TRY(dmalloc(&d_zc, 2 * (size_t)M)); tmp.p[3] = d_zc;
        PRE3_HIP(hipMemcpy(d_pred, pred.data(), sizeof(int32_t) * pred.size(), hipMemcpyHostToDevice));
        PRE3_HIP(hipMemcpy(d_k1, k1, sizeof(int32_t) * M, hipMemcpyHostToDevice));
        PRE3_HIP(hipMemcpy(d_zc, zc, sizeof(double) * 2 * M, hipMemcpyHostToDevice));
        PRE3_HIP(hipMemsetAsync(c->lm.ic, 0, sizeof(int32_t) * N, c->stream));
        PRE3_TRY(launch_window_gate(c, M, d_pred, d_k1, d_zc, strict_reference, d_acc));
        PRE3_TRY(stream_drain(c, __func__));
        PRE3_HIP(hipMemcpy(acc.data(), d_acc, sizeof(int32_t) * M, hipMemcpyDeviceToHost));
    }
    // accepted candidates become the measurement list (ascending landmark order; one match per landmark)
    std::vector<int32_t> flag(N ? N : 1, 0);
    for (int q = 0; q < M; ++q) if (acc[q]) flag[pred[k1[q]]] = 1;
    std::vector<int32_t> meas;
    for (int i = 0; i < N; ++i) if (flag[i]) meas.push_back(i);
    if (accept_out) for (int q = 0; q < M; ++q) accept_out[q] = acc[q];
    // z was written on the device by the gate kernel; keep it (z == nullptr)
    return install_measurements(c, (int)meas.size(), meas.data(), nullptr, nullptr, 0);
}

// ---- IC search on the device ---------------------------------------------------------
static int ensure_ic_buffers(pre3_ctx *c)
{
    if (c->bank) return PRE3_OK;
    const size_t N = (size_t)c->capN;
    PRE3_TRY(dmalloc(&c->bank, N * DESC_DIM)); PRE3_TRY(dmalloc(&c->bank_alt, N * DESC_DIM));
    // result block, fetched with ONE copy: [counts(4) | meas(capN) | pairs(3 capN) | z(2 capN doubles)]
    PRE3_TRY(dmalloc(&c->ic_pred, N)); PRE3_TRY(dmalloc(&c->ic_counts, 4 + 4 * N + 4 * N + 4)); PRE3_TRY(dmalloc(&c->ic_arg, N));
    c->ic_pairs = c->ic_counts + 4 + N;
    PRE3_HIP(hipMemset(c->ic_counts, 0, sizeof(int32_t) * (4 + 8 * N)));
    PRE3_TRY(dmalloc(&c->ic_newk2, N)); PRE3_TRY(dmalloc(&c->ic_best, N)); PRE3_TRY(dmalloc(&c->ic_second, N)); PRE3_TRY(dmalloc(&c->bank_src, N));
    PRE3_HIP(hipMemset(c->bank, 0, sizeof(double) * N * DESC_DIM));
    const size_t blk_bytes = (sizeof(int32_t) * (4 + 4 * N) + sizeof(double) * 2 * N + 15) & ~(size_t)15;
    PRE3_HIP(hipHostMalloc(&c->ic_result_host, blk_bytes, hipHostMallocMapped));
    PRE3_HIP(hipHostGetDevicePointer(&c->ic_result_host_dev, c->ic_result_host, 0));
    return PRE3_OK;
}

// k_rank_pack's test of its input, on the host: every value finite with |x| <= 2^60 and no non-zero |x| < 2^-40 (an or-reduction of two
// compares per value), made on the way into the staging block: one pass over the caller's array instead of a copy and a second read
// (AVX2 where the host has it: 36 us for copy + check of 600 keypoints -> see DESIGN.md section 11)
__attribute__((target("avx2"))) static bool copy_desc_checked_avx2(double *__restrict__ dst, const double *__restrict__ src, size_t count)
{
    const __m256d absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffLL)), hi = _mm256_set1_pd(0x1p60), lo = _mm256_set1_pd(0x1p-40), zero = _mm256_setzero_pd();
    __m256d bad = zero;
    size_t i = 0;
    for (; i + 8 <= count; i += 8) {
        const __m256d v0 = _mm256_loadu_pd(src + i), v1 = _mm256_loadu_pd(src + i + 4);
        _mm256_storeu_pd(dst + i, v0); _mm256_storeu_pd(dst + i + 4, v1);
        const __m256d a0 = _mm256_and_pd(v0, absmask), a1 = _mm256_and_pd(v1, absmask);
        // !(ax <= 2^60) (true for NaN too)  |  (ax != 0 && ax < 2^-40)
        bad = _mm256_or_pd(bad, _mm256_cmp_pd(a0, hi, _CMP_NLE_UQ)); bad = _mm256_or_pd(bad, _mm256_cmp_pd(a1, hi, _CMP_NLE_UQ));
        bad = _mm256_or_pd(bad, _mm256_and_pd(_mm256_cmp_pd(a0, zero, _CMP_NEQ_OQ), _mm256_cmp_pd(a0, lo, _CMP_LT_OQ)));
        bad = _mm256_or_pd(bad, _mm256_and_pd(_mm256_cmp_pd(a1, zero, _CMP_NEQ_OQ), _mm256_cmp_pd(a1, lo, _CMP_LT_OQ)));
    }
    int b = _mm256_movemask_pd(bad);
    for (; i < count; ++i) {
        const double v = src[i], ax = fabs(v);
        dst[i] = v;
        b |= (int)!(ax <= 0x1p60);
        b |= (int)(ax != 0.0) & (int)(ax < 0x1p-40);
    }
    return b == 0;
}
static bool copy_desc_checked(double *__restrict__ dst, const double *__restrict__ src, size_t count)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return copy_desc_checked_avx2(dst, src, count);
    int bad = 0;
    for (size_t i = 0; i < count; ++i) {
        const double v = src[i], ax = fabs(v);
        dst[i] = v;
        bad |= (int)!(ax <= 0x1p60);
        bad |= (int)(ax != 0.0) & (int)(ax < 0x1p-40);
    }
    return bad == 0;
}

// A pinned block of the context for `bytes` of host data on their way to the device: two blocks, used alternately, grown on demand; a block is
// written again only after the pull of its previous contents has run (its sequence number in the mailbox: stage_wait).  The caller fills *host, enqueues the pull from *dev on the
// context's stream and calls stage_release.
static int stage_acquire(pre3_ctx *c, size_t bytes, void **host, void **dev, int *slot)
{
    if (c->up_stage_bytes < bytes) {
        for (int k = 0; k < 2; ++k) {
            if (c->up_stage_used[k]) PRE3_TRY(stage_wait(c, k));
            if (c->up_stage[k]) (void)hipHostFree(c->up_stage[k]);
            c->up_stage[k] = nullptr; c->up_stage_used[k] = false;
        }
        c->up_stage_bytes = 0;
        const size_t cap = (bytes + 65535) & ~(size_t)65535;
        for (int k = 0; k < 2; ++k) {
            PRE3_HIP(hipHostMalloc(&c->up_stage[k], cap, hipHostMallocMapped));
        }
        c->up_stage_bytes = cap;
    }
    const int k = c->up_stage_next; c->up_stage_next ^= 1;
    if (c->up_stage_used[k]) PRE3_TRY(stage_wait(c, k));
    *host = c->up_stage[k]; *slot = k;
    PRE3_HIP(hipHostGetDevicePointer(dev, c->up_stage[k], 0));
    return PRE3_OK;
}
static int stage_release(pre3_ctx *c, int slot)        // (the pull launched with stage_done(c, slot) announces itself)
{
    c->up_stage_used[slot] = true;
    return PRE3_OK;
}

int pre3_set_descriptors(pre3_ctx *c, int first, int count, const double *desc)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(first >= 0 && count >= 0 && first + count <= c->N && (count == 0 || desc), PRE3_E_ARG, "pre3_set_descriptors: range [%d, %d) outside the map (N=%d)", first, first + count, c->N);
    PRE3_TRY(ensure_ic_buffers(c));
    bool ok = true;
    if (count) {
        // staged and pulled on the context's stream like the scan (pre3_set_scan): map management calls this once per frame for the new
        // landmark -- it used to cost a stream synchronisation, a blocking copy and a device-side bounds check with a read-back
        const size_t nd = (size_t)count * DESC_DIM;
        void *st = nullptr, *st_dev = nullptr; int slot = 0;
        PRE3_TRY(stage_acquire(c, sizeof(double) * nd, &st, &st_dev, &slot));
        ok = copy_desc_checked(static_cast<double *>(st), desc, nd);
        hipLaunchKernelGGL(k_scan_pull, dim3(ceil_div((int)(nd / 2), 256)), dim3(256), 0, c->stream, (const int4 *)st_dev, (int)(nd / 2),
                           (int4 *)(c->bank + (size_t)first * DESC_DIM), 0, (int4 *)nullptr, stage_done(c, slot));
        PRE3_HIP(hipGetLastError());
        PRE3_TRY(stage_release(c, slot));
    }
    c->bank_set = true;
    if (!ok) c->bank_ok = false;            // (sticky until the whole bank is rewritten: a bad descriptor may stay in the map)
    else if (first == 0 && count >= c->N) c->bank_ok = true;
    return PRE3_OK;
}

int pre3_get_descriptors(pre3_ctx *c, int first, int count, double *desc)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(first >= 0 && count >= 0 && first + count <= c->N && (count == 0 || desc), PRE3_E_ARG, "pre3_get_descriptors: range outside the map");
    PRE3_CHECK(c->bank_set, PRE3_E_STATE, "pre3_get_descriptors: no descriptors have been set");
    PRE3_TRY(stream_drain(c, __func__));
    if (count) PRE3_HIP(hipMemcpy(desc, c->bank + (size_t)first * DESC_DIM, sizeof(double) * (size_t)count * DESC_DIM, hipMemcpyDeviceToHost));
    return PRE3_OK;
}

int pre3_set_scan(pre3_ctx *c, int K2, const double *descriptor_raw, const double *scale_orient_pos_raw)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(K2 >= 0 && (K2 == 0 || (descriptor_raw && scale_orient_pos_raw)), PRE3_E_ARG, "pre3_set_scan: bad arguments");
    if (K2 > c->scan_cap) {
        PRE3_TRY(stream_drain(c, __func__));             // (the buffers being replaced may still be read by queued kernels)
        if (c->scan_desc) (void)hipFree(c->scan_desc);
        if (c->scan_pos) (void)hipFree(c->scan_pos);
        if (c->ic_pb) (void)hipFree(c->ic_pb);
        if (c->ic_ps) (void)hipFree(c->ic_ps);
        if (c->ic_pa) (void)hipFree(c->ic_pa);
        c->scan_desc = c->scan_pos = nullptr; c->ic_pb = c->ic_ps = nullptr; c->ic_pa = nullptr; c->scan_cap = 0; c->ic_pcap = 0;
        const int cap = round_up(K2, 256);
        PRE3_TRY(dmalloc(&c->scan_desc, (size_t)cap * DESC_DIM)); PRE3_TRY(dmalloc(&c->scan_pos, (size_t)cap * 4));
        const size_t np = std::max((size_t)(cap / 64) * c->capN, (size_t)64 * c->capN);     // [scan_cap/64][capN] for the 64-wide tiles; the fused route: [capN][64 column tiles of 32]
        c->ic_pcap = np;
        PRE3_TRY(dmalloc(&c->ic_pb, np)); PRE3_TRY(dmalloc(&c->ic_ps, np)); PRE3_TRY(dmalloc(&c->ic_pa, np));
        c->scan_cap = cap;
    }
    bool in_bounds = true;
    if (K2) {
        // The frame's scan crosses PCIe from a pinned block of the context's own, enqueued on its stream: the call neither waits for the queued
        // work nor ends in a read-back.  The bounds the ranked route needs of the descriptors (k_rank_pack's test: finite, |x| <= 2^60, no
        // non-zero |x| < 2^-40) are checked here, on the way into that block.
        static const int trace = getenv("PRE3_SCAN_TRACE") ? atoi(getenv("PRE3_SCAN_TRACE")) : 0;
        const auto t0 = std::chrono::steady_clock::now();
        void *st_v = nullptr, *st_dev = nullptr; int k = 0;
        PRE3_TRY(stage_acquire(c, sizeof(double) * (size_t)K2 * (DESC_DIM + 4), &st_v, &st_dev, &k));
        const auto t1 = std::chrono::steady_clock::now();
        double *st = static_cast<double *>(st_v);
        const size_t nd = (size_t)K2 * DESC_DIM;
        in_bounds = copy_desc_checked(st, descriptor_raw, nd);
        const auto t2 = std::chrono::steady_clock::now();
        memcpy(st + nd, scale_orient_pos_raw, sizeof(double) * (size_t)K2 * 4);
        const auto t3 = std::chrono::steady_clock::now();
        // the device reads the block over PCIe itself (16 bytes per lane, every request in flight at once: ~20 us for 600 keypoints) instead of
        // two DMA-engine copies with their start-up latencies
        const int n16_desc = (int)(nd / 2), n16_pos = K2 * 2;
        hipLaunchKernelGGL(k_scan_pull, dim3(ceil_div(n16_desc + n16_pos, 256)), dim3(256), 0, c->stream, (const int4 *)st_dev, n16_desc, (int4 *)c->scan_desc, n16_pos, (int4 *)c->scan_pos, stage_done(c, k));
        PRE3_HIP(hipGetLastError());
        PRE3_TRY(stage_release(c, k));
        if (trace) { const auto t4 = std::chrono::steady_clock::now(); auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "[pre3 set_scan, us] event wait %.1f | descriptors copied + checked %.1f | positions %.1f | enqueue %.1f\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4)); }
    }
    c->scan_K2 = K2;
    return ic_rank_set_scan(c, in_bounds);
}

int pre3_ic_search(pre3_ctx *c, double thresh, int strict_reference, int32_t *n_matches_out, int32_t *m_out, int32_t *meas_idx_out,
                   double *z_out, int32_t *pairs_out)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->bank_set, PRE3_E_STATE, "pre3_ic_search: call pre3_set_descriptors first");
    PRE3_CHECK(c->scan_K2 >= 0 && (c->scan_K2 == 0 || c->scan_desc), PRE3_E_STATE, "pre3_ic_search: call pre3_set_scan first");
    PRE3_CHECK(c->x_valid[PRE3_X_K_KM1] && c->p_which == PRE3_X_K_KM1, PRE3_E_STATE, "pre3_ic_search: needs the predicted estimate (call pre3_predict first)");
    PRE3_CHECK(m_out != nullptr, PRE3_E_ARG, "pre3_ic_search: null m_out");
    const int N = c->N;
    if (N == 0) {
        // matching_sift_based.m:115 `if isempty(des1) return`: a SLAM run starts with an empty map -- nothing to match, no measurements
        if (n_matches_out) *n_matches_out = 0;
        *m_out = 0;
        c->projected = true; c->innovated = true;
        return install_measurements(c, 0, nullptr, nullptr, nullptr, 0);
    }
    // search_IC_matches.m:31-44: h, H and S for every landmark at the prediction
    PRE3_CHECK(c->have_cam, PRE3_E_STATE, "pre3_ic_search: camera not set");
    const bool fused = ic_search_fused_applies(c);
    const IcMatchRide ride = fused ? ic_match_ride(c) : IcMatchRide{};          // the fused route's matcher tiles ride in the projection's launch
    if (N) PRE3_TRY(launch_project_innovation(c, PRE3_X_K_KM1, 1, 0, 0.0, true, true, ride.n_blocks ? &ride : nullptr));      // (also clears individually_compatible of every landmark)
    c->projected = true; c->innovated = true;
    // the result block [counts | meas | pairs | z] is written into mapped pinned memory by the device itself and announced through the mailbox:
    // no DMA-engine copy, no stream synchronisation (the host polls one word)
    const size_t capN = (size_t)c->capN, blk_bytes = sizeof(int32_t) * (4 + 4 * capN) + sizeof(double) * 2 * capN;
    if (fused) {
        // the reference's real sizes: exact matcher + (stack, merge, gate, refresh, result block) as two launches (pre3_match.hip)
        c->ic_last_ranked = false;
        PRE3_TRY(launch_ic_search_fused(c, thresh, strict_reference, ++c->seq_ic, 12, ride.n_blocks > 0));
    } else {
        PRE3_TRY(launch_ic_search(c, thresh, strict_reference));
        PRE3_TRY(launch_inbox_pull(c, c->ic_counts, c->ic_result_host_dev, (blk_bytes + 15) / 16, ++c->seq_ic, 12));
    }
    PRE3_TRY(wait_mail(c, 12, c->seq_ic));
    const int32_t *blk_p = static_cast<const int32_t *>(c->ic_result_host);
    std::vector<int32_t> blk(blk_p, blk_p + (4 + 4 * capN + 4 * capN));
    const int32_t *counts = blk.data();
    const int n_match = c->scan_K2 > 0 ? counts[1] : 0, m = c->scan_K2 > 0 ? counts[2] : 0;
    PRE3_CHECK(m >= 0 && m <= N && n_match >= 0 && n_match <= N, PRE3_E_STATE, "pre3_ic_search: inconsistent counts (%d matches, %d accepted)", n_match, m);
    std::vector<int32_t> meas(blk.begin() + 4, blk.begin() + 4 + m);
    if (n_matches_out) *n_matches_out = n_match;
    if (pairs_out) for (int i = 0; i < 3 * n_match; ++i) pairs_out[i] = blk[4 + capN + i];
    *m_out = m;
    if (meas_idx_out) for (int j = 0; j < m; ++j) meas_idx_out[j] = meas[j];
    if (z_out) { const double *zz = (const double *)(blk.data() + 4 + 4 * capN); for (int j = 0; j < 2 * m; ++j) z_out[j] = zz[j]; }
    return install_measurements(c, (int)meas.size(), meas.data(), nullptr, nullptr, 0);
}

// ---- RANSAC ---------------------------------------------------------------------------------------
// zero_words > 0 (the sliced forms): that many words of c->support (supports, masks [, the missing-slice word]) are cleared on the way
static int ransac_prepare(pre3_ctx *c, int n_draw, int k, const int32_t *hyp, int lo = 0, int hi = -1, bool slice_form = false, size_t zero_words = 0)
{
    PRE3_CHECK(c->measurements_set && c->projected, PRE3_E_STATE, "ransac: needs pre3_project and measurements");
    PRE3_CHECK(c->p_which == PRE3_X_K_KM1, PRE3_E_STATE, "ransac: needs the predicted estimate (call pre3_predict or set x_k_km1)");
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph, PRE3_E_ARG, "ransac: n_draw=%d exceeds capacity %d", n_draw, c->caph);
    PRE3_CHECK(k >= 1 && k <= MAXK, PRE3_E_ARG, "ransac: k=%d unsupported (1..%d)", k, MAXK);
    PRE3_CHECK(hyp != nullptr, PRE3_E_ARG, "ransac: null hypothesis table");
    PRE3_CHECK(c->m >= k, PRE3_E_ARG, "ransac: %d measurements but k=%d", c->m, k);
    for (int i = 0; i < n_draw * k; ++i) PRE3_CHECK(hyp[i] >= 0 && hyp[i] < c->m, PRE3_E_ARG, "ransac: hyp[%d]=%d not a position in the IC list (m=%d)", i, hyp[i], c->m);
    if (hi < 0) hi = n_draw;
    const bool sliced = lo > 0 || hi < n_draw || slice_form;
    const bool need_pull = hyp != (const int32_t *)(c->inbox_host + c->off_hyp);       // not already shipped with the measurements
    if (need_pull) {
        if (c->inbox_pending) { PRE3_TRY(wait_mail(c, 10, c->seq_inbox)); c->inbox_pending = false; }
        memcpy(c->inbox_host + c->off_hyp, hyp, sizeof(int32_t) * n_draw * k);
    }
    if (sliced) {
        // one launch: the clear, the pull and the marks of the measurements this slice's hypotheses draw (k_slice_prepare)
        ++c->need_tag;
        PRE3_TRY(launch_slice_prepare(c, (const unsigned char *)c->inbox_host_dev + c->off_hyp, need_pull ? (sizeof(int32_t) * n_draw * k + 15) / 16 : 0,
                                      need_pull ? ++c->seq_inbox : c->seq_inbox, (int)zero_words, k, lo, hi, c->need_tag));
        if (need_pull) c->inbox_pending = true;
    } else {
        if (zero_words) PRE3_HIP(hipMemsetAsync(c->support, 0, sizeof(int32_t) * zero_words, c->stream));
        // (need_pull: the caller's table crosses PCIe as a rider of the H*P launch below -- its first reader is the scorer behind that launch)
    }
    c->masks = reinterpret_cast<uint32_t *>(c->support + round_up(n_draw, 4));
    c->scored_n_draw = n_draw; c->scored_k = k;         // the mask offset depends on n_draw: select / export / import must use the same
    int r = 2 * c->m, r_pad = round_up(r, NB);
    static const int inline_g_env = getenv("PRE3_INLINE_G") ? atoi(getenv("PRE3_INLINE_G")) : 1;
    const bool inline_g = inline_g_env != 0;
    if (sliced) {
        // a rank's slice of a sharded round: H*P and H*P*H' only for the measurements its hypotheses draw (the scorer of hypothesis h
        // reads the 2k rows of its own landmarks and the entries of G among them, nothing else) -- the part of the round that
        // shrinks with the number of ranks.  The LI update must not gather from these partial products: hp_all_valid stays false.
        PRE3_TRY(launch_ell_HP_build(c, c->HP, c->need, c->need_tag));
        if (!inline_g) PRE3_TRY(launch_ell_G_hyp(c, k, lo, hi, r_pad));
        c->g_valid = !inline_g;
        c->hp_all_valid = false;
        return PRE3_OK;
    }
    if (need_pull) {
        static const int ride_env = getenv("PRE3_HYP_RIDE") ? atoi(getenv("PRE3_HYP_RIDE")) : 1;      // 0: the pull as a launch of its own (rounds 2-4)
        const size_t n16 = (sizeof(int32_t) * n_draw * k + 15) / 16;
        if (ride_env) {
            const InboxRide ib{ (const int4 *)((const unsigned char *)c->inbox_host_dev + c->off_hyp), (int4 *)c->hyp, (int)n16, c->mail_dev, ++c->seq_inbox, 10, nullptr, 0 };
            c->inbox_pending = true;
            PRE3_TRY(launch_ell_HP_build(c, c->HP, nullptr, 0, &ib));
        } else {
            PRE3_TRY(launch_inbox_pull(c, (const unsigned char *)c->inbox_host_dev + c->off_hyp, c->hyp, n16, ++c->seq_inbox)); c->inbox_pending = true;
            PRE3_TRY(launch_ell_HP_build(c, c->HP));
        }
    } else
    PRE3_TRY(launch_ell_HP_build(c, c->HP));
    // H*P*H' of all measured rows is no longer built (6.6 us of launch in front of the scoring, PRE3_INLINE_G=0 brings it back): the scorer
    // computes the (2k)^2 entries among its hypothesis' rows and the LI gather the entries of S it needs, both with k_ell_G's sum
    if (!inline_g) PRE3_TRY(launch_ell_G(c, r, c->HP, c->G, r_pad, 0, nullptr, true));      // lower triangle: the scorer and the LI gather read (max, min)
    c->g_valid = !inline_g;
    c->hp_all_valid = true;
    return PRE3_OK;
}

int pre3_ransac_score(pre3_ctx *c, int n_draw, int k, const int32_t *hyp, double threshold, int hyp_begin, int hyp_end,
                      void **support_dev, void **mask_dev, int *mask_words)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(hyp_begin >= 0 && hyp_begin <= hyp_end && hyp_end <= n_draw, PRE3_E_ARG, "ransac: bad hypothesis range [%d,%d) of %d", hyp_begin, hyp_end, n_draw);
    int words = ceil_div(c->m, 32);
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph, PRE3_E_ARG, "ransac: n_draw=%d exceeds capacity %d", n_draw, c->caph);
    PRE3_TRY(ransac_prepare(c, n_draw, k, hyp, hyp_begin, hyp_end, false, (size_t)round_up(n_draw, 4) + (size_t)n_draw * words));   // (supports + masks cleared on the way)
    PRE3_TRY(launch_ransac_score_impl(c, k, threshold, hyp_begin, hyp_end, round_up(2 * c->m, NB), c->support, c->masks, words));
    if (support_dev) *support_dev = c->support;
    if (mask_dev) *mask_dev = c->masks;
    if (mask_words) *mask_words = words;
    PRE3_TRY(stream_drain(c, __func__));     // the caller's collective runs on another stream
    return PRE3_OK;
}

// after the selection stage has been enqueued (its own kernel, or the tail of the scoring launch)
static int ransac_results(pre3_ctx *c, int n_draw, int32_t *support, int32_t *li_mask, int32_t stats[4])
{
    c->li_from_host = -1; c->li_kernel = true;
    if (support || li_mask) {
        // (behind a collective the synchronisation comes second: the selection's mailbox word first, under the communicator's deadline)
        if (c->shard_round) PRE3_TRY(wait_mail(c, 8, c->seq_select));
        PRE3_TRY(stream_drain(c, __func__));
        if (support) PRE3_HIP(hipMemcpy(support, c->support, sizeof(int32_t) * n_draw, hipMemcpyDeviceToHost));
        if (li_mask && c->m) PRE3_HIP(hipMemcpy(li_mask, c->li_meas, sizeof(int32_t) * c->m, hipMemcpyDeviceToHost));
    }
    if (stats) {
        PRE3_TRY(wait_mail(c, 8, c->seq_select));
        PRE3_CHECK(!c->shard_round || c->mail_host[11] == 0, PRE3_E_COMM, "sharded RANSAC: a rank failed before the collective of this round (its slice is missing from the sums)");
        for (int i = 0; i < 4; ++i) stats[i] = c->mail_host[i];
    }
    return PRE3_OK;
}

int pre3_ransac_select(pre3_ctx *c, int n_draw, int k, int early_exit, int32_t *support, int32_t *li_mask, int32_t stats[4])
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph, PRE3_E_ARG, "ransac: n_draw out of range");
    PRE3_CHECK(n_draw == c->scored_n_draw && k == c->scored_k, PRE3_E_STATE, "pre3_ransac_select: n_draw=%d, k=%d differs from the scored round (n_draw=%d, k=%d): the mask buffer is laid out for that round", n_draw, k, c->scored_n_draw, c->scored_k);
    int words = ceil_div(c->m, 32);
    PRE3_TRY(launch_ransac_select_impl(c, n_draw, k, early_exit, c->support, c->masks, words));
    return ransac_results(c, n_draw, support, li_mask, stats);
}

int pre3_ransac_export(pre3_ctx *c, int n_draw, void *support_dst_dev, void *mask_dst_dev)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph, PRE3_E_ARG, "ransac: n_draw out of range");
    PRE3_CHECK(n_draw == c->scored_n_draw, PRE3_E_STATE, "pre3_ransac_export: n_draw=%d differs from the scored round (n_draw=%d): the mask buffer is laid out for that round", n_draw, c->scored_n_draw);
    int words = ceil_div(c->m, 32);
    if (support_dst_dev) PRE3_HIP(hipMemcpyAsync(support_dst_dev, c->support, sizeof(int32_t) * n_draw, hipMemcpyDeviceToDevice, c->stream));
    if (mask_dst_dev) PRE3_HIP(hipMemcpyAsync(mask_dst_dev, c->masks, sizeof(uint32_t) * (size_t)n_draw * words, hipMemcpyDeviceToDevice, c->stream));
    PRE3_TRY(stream_drain(c, __func__));
    return PRE3_OK;
}

int pre3_ransac_import(pre3_ctx *c, int n_draw, const void *support_src_dev, const void *mask_src_dev)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph, PRE3_E_ARG, "ransac: n_draw out of range");
    PRE3_CHECK(n_draw == c->scored_n_draw, PRE3_E_STATE, "pre3_ransac_import: n_draw=%d differs from the scored round (n_draw=%d): the mask buffer is laid out for that round", n_draw, c->scored_n_draw);
    int words = ceil_div(c->m, 32);
    if (support_src_dev) PRE3_HIP(hipMemcpyAsync(c->support, support_src_dev, sizeof(int32_t) * n_draw, hipMemcpyDeviceToDevice, c->stream));
    if (mask_src_dev) PRE3_HIP(hipMemcpyAsync(c->masks, mask_src_dev, sizeof(uint32_t) * (size_t)n_draw * words, hipMemcpyDeviceToDevice, c->stream));
    PRE3_TRY(stream_drain(c, __func__));
    return PRE3_OK;
}

// ---- RCCL communicator on the context (pre3_comm.hip) ------------------------------------------------
int pre3_set_comm(pre3_ctx *c, pre3_comm *comm)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(comm == nullptr || comm_device(comm) == c->device, PRE3_E_ARG, "pre3_set_comm: the communicator lives on device %d, the context on %d", comm ? comm_device(comm) : -1, c->device);
    if (c->comm && c->comm != (void *)comm) {
        // nothing queued on the stream may still use the handle being replaced (the caller may destroy it next); after a deadline: its abort has come back
        PRE3_TRY(stream_drain(c, __func__));
        PRE3_CHECK(comm_abort_wait(c->comm, comm_abort_ms(c->comm)), PRE3_E_COMM, "pre3_set_comm: the abort of the previous communicator has not returned yet");
        if (c->comm_owned) (void)pre3_comm_destroy((pre3_comm *)c->comm);
    }
    c->comm = comm; c->comm_owned = false;
    return PRE3_OK;
}

int pre3_comm_init(pre3_ctx *c, const void *id, int rank, int world)
{
    PRE3_TRY(check_ctx(c));
    pre3_comm *cm = nullptr;
    PRE3_TRY(pre3_comm_create(&cm, c->device, id, rank, world));
    const int rc = pre3_set_comm(c, cm);
    if (rc != PRE3_OK) { (void)pre3_comm_destroy(cm); return rc; }
    c->comm_owned = true;
    return PRE3_OK;
}

// One sharded RANSAC round with everything on the context's stream: [H*P | H*P*H' of this rank's measurements] -> scoring of hypotheses
// [lo, hi) -> ncclAllReduce(sum) of [supports | masks], in place (the slices are disjoint and the buffer is cleared first: the integer sum
// is the union) -> selection.  The host waits once, on the selection's mailbox word.
// ---- test hook (pre3_test_hooks.h; inert without PRE3_TEST_HOOKS=1): a kernel that keeps the stream busy until the host releases it (or ~20 s have passed)
static bool test_hooks_on() { const char *e = getenv("PRE3_TEST_HOOKS"); return e && atoi(e) == 1; }
__global__ void k_test_stall(volatile int32_t *flag)
{
    for (long spin = 0; spin < 6000000L; ++spin) {          // (~20 s: the kernel lets go by itself)
        if (__hip_atomic_load(const_cast<int32_t *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
        __builtin_amdgcn_s_sleep(100);
    }
}
int pre3_test_stall(pre3_ctx *c, int release)
{
    PRE3_CHECK(test_hooks_on(), PRE3_E_STATE, "pre3_test_stall: test hooks are off (PRE3_TEST_HOOKS=1 enables them)");
    PRE3_CHECK(c != nullptr, PRE3_E_ARG, "null context");
    PRE3_HIP(hipSetDevice(c->device));
    if (release) { __atomic_store_n(c->mail_host + 15, 1, __ATOMIC_RELEASE); return PRE3_OK; }
    __atomic_store_n(c->mail_host + 15, 0, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(k_test_stall, dim3(1), dim3(1), 0, c->stream, (volatile int32_t *)(c->mail_dev + 15));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

int pre3_ransac_sharded(pre3_ctx *c, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit, int32_t *support, int32_t *li_mask,
                        int32_t stats[4])
{
    // What may differ between the ranks must not decide whether a rank enters the collective: only the arguments every rank passes alike (the
    // communicator, n_draw) return early.  Everything rank-local -- the deferred work of the previous step (check_ctx), the measurements, the
    // table, a failed launch -- is folded into rc_local: the rank then still enters ncclAllReduce, with its slice zero and the missing-slice
    // word set, and every rank fails the round with PRE3_E_COMM instead of waiting for a partner that has returned.
    PRE3_CHECK(c != nullptr, PRE3_E_ARG, "null context");
    PRE3_HIP(hipSetDevice(c->device));
    PRE3_CHECK(c->comm != nullptr, PRE3_E_STATE, "pre3_ransac_sharded: no communicator (pre3_comm_init / pre3_set_comm)");
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph, PRE3_E_ARG, "ransac: n_draw=%d exceeds capacity %d", n_draw, c->caph);
    int rc_local = check_ctx(c);
    int rank = 0, world = 1;
    comm_rank_world(c->comm, &rank, &world);
    const int base = n_draw / world, rem = n_draw % world;
    const int lo = rank * base + std::min(rank, rem), hi = lo + base + (rank < rem ? 1 : 0);
    const int words = ceil_div(c->m, 32);
    // the element count of the all-reduce comes from the contexts' capacities, which the ranks share (replicas), not from this rank's
    // measurement count: supports | masks laid out for the capacity's mask words | the missing-slice word
    const size_t count = (size_t)round_up(n_draw, 4) + (size_t)n_draw * c->mask_words_cap;
    if (rc_local == PRE3_OK) rc_local = ransac_prepare(c, n_draw, k, hyp, lo, hi, true, count + 1);
    if (rc_local != PRE3_OK) (void)hipMemsetAsync(c->support, 0, sizeof(int32_t) * (count + 1), c->stream);      // (a failure in front of the prepare launch: the buffer must still be clear)
    if (rc_local == PRE3_OK && hi > lo) rc_local = launch_ransac_score_impl(c, k, threshold, lo, hi, round_up(2 * c->m, NB), c->support, c->masks, words);
    if (rc_local != PRE3_OK) {
        (void)hipMemsetAsync(c->support, 0, sizeof(int32_t) * count, c->stream);                      // whatever part of the slice got written does not count
        (void)hipMemsetAsync(c->support + count, 1, sizeof(int32_t), c->stream);                       // (0x01010101: non-zero is all that matters)
    }
    const int rc_coll = comm_all_reduce_i32(c->comm, c->support, count + 1, c->stream);
    if (rc_local != PRE3_OK) return rc_local;
    PRE3_TRY(rc_coll);
    PRE3_TRY(launch_ransac_select_impl(c, n_draw, k, early_exit, c->support, c->masks, words, (int)count));
    c->shard_round = true;
    return ransac_results(c, n_draw, support, li_mask, stats);
}

int pre3_ransac(pre3_ctx *c, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit, int32_t *support, int32_t *li_mask,
                int32_t stats[4])
{
    PRE3_TRY(check_ctx(c));
    c->shard_round = false; c->select_pending = false;
    PRE3_TRY(ransac_prepare(c, n_draw, k, hyp));
    int words = ceil_div(c->m, 32);
    // Scoring, then the selection stage (the reference's loop replayed on the supports) as a launch of its own.  The selection can also ride
    // in the scoring launch's last workgroup (PRE3_SELECT_FUSE=1, round 1's form), but measured at N=500 / 200 hypotheses that launch then
    // takes 21.2 us against 10.0 + 6.7 us for the two (tools/score_split.py): every workgroup pays a device-scope release (an L2 write-back)
    // and a ticket before it may finish, and the last one starts the selection behind an L2 invalidate.
    static const int fuse_env = getenv("PRE3_SELECT_FUSE") ? atoi(getenv("PRE3_SELECT_FUSE")) : 0;
    if (fuse_env) PRE3_TRY(launch_ransac_score_impl(c, k, threshold, 0, n_draw, round_up(2 * c->m, NB), c->support, c->masks, words, n_draw, early_exit));
    else {
        PRE3_TRY(launch_ransac_score_impl(c, k, threshold, 0, n_draw, round_up(2 * c->m, NB), c->support, c->masks, words, 0, 0));
        // pre3_step: the selection rides in the LI gather's launch (k_select_gather), which pre3_update_li sends next
        if (c->defer_select && !support && !li_mask && !stats && select_gather_usable(c)) {
            c->select_pending = true; c->sel_n_draw = n_draw; c->sel_k = k; c->sel_early_exit = early_exit;
            c->li_from_host = -1; c->li_kernel = true;
            return PRE3_OK;
        }
        PRE3_TRY(launch_ransac_select_impl(c, n_draw, k, early_exit, c->support, c->masks, words));
    }
    return ransac_results(c, n_draw, support, li_mask, stats);
}

// ---- updates --------------------------------------------------------------------------------------
static int update_selected(pre3_ctx *c, int which_prior, int nsel, const int32_t *sel_dev, bool gathered = false, bool first_done = false)
{
    PRE3_CHECK(c->p_which == which_prior, PRE3_E_STATE, "update: the covariance buffer does not hold the required prior");
    int r = 2 * nsel;
    // rows of the predicted-state update that RANSAC already multiplied out: gather instead of recomputing
    const bool reuse = r > 0 && which_prior == PRE3_X_K_KM1 && c->hp_all_valid && sel_dev != nullptr;
    bool hp_built = false;
    if (reuse) { if (!gathered) PRE3_TRY(launch_gather_li(c, nsel, nsel, sel_dev, round_up(2 * c->m, NB))); }
    else if (r > 0) {
        // rows built on the fly inside the H*P launch (one launch instead of k_build_rows + k_ell_HP; PRE3_FUSE_ROWS=0: the two)
        static const int fuse_rows = getenv("PRE3_FUSE_ROWS") ? atoi(getenv("PRE3_FUSE_ROWS")) : 1;
        if (fuse_rows && round_up(r, NB) <= c->rcap) { PRE3_TRY(launch_ell_HP_build_sel(c, nsel, sel_dev, c->W)); hp_built = true; }
        else PRE3_TRY(launch_build_rows_impl(c, nsel, sel_dev, round_up(r, NB)));
    }
    PRE3_TRY(run_update(c, which_prior, r, false, nullptr, reuse, first_done && reuse, hp_built));
    c->hp_all_valid = false;                 // P changed
    c->x_valid[PRE3_X_K_K] = true; c->p_which = PRE3_X_K_K;
    return PRE3_OK;
}

int pre3_update_li(pre3_ctx *c)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->measurements_set && c->projected, PRE3_E_STATE, "pre3_update_li: needs projection and measurements");
    int n_li = 0;       // no RANSAC / flags for this measurement set: no low-innovation inliers, update is the identity
    bool gathered = false, first_done = false;
    if (c->li_from_host >= 0) n_li = c->li_from_host;
    else if (c->li_kernel) {
        const bool fused_sel = c->select_pending && c->p_which == PRE3_X_K_KM1 && c->hp_all_valid && c->m > 0;
        if (c->select_pending && !fused_sel) PRE3_TRY(launch_ransac_select_impl(c, c->sel_n_draw, c->sel_k, c->sel_early_exit, c->support, c->masks, ceil_div(c->m, 32)));
        c->select_pending = false;
        // the gather of the LI rows does not need the count on the host: issue it first, with the grid sized for all
        // measurements, so that the GPU has work while the host polls the mailbox and launches the factorisation
        if (c->p_which == PRE3_X_K_KM1 && c->hp_all_valid && c->m > 0) {
            if (fused_sel) PRE3_TRY(launch_select_gather(c, c->sel_n_draw, c->sel_k, c->sel_early_exit, ceil_div(c->m, 32)));
            else PRE3_TRY(launch_gather_li(c, -1, c->m, c->sel_rows, round_up(2 * c->m, NB)));
            gathered = true;
            // ... and so does the first panel of the factorisation (row count read on the device, grid sized for all measurements)
            static const int spec_env = getenv("PRE3_CHOL_SPEC0") ? atoi(getenv("PRE3_CHOL_SPEC0")) : 1;
            if (spec_env && round_up(2 * c->m, NB) <= c->rcap) {
                // fp32: the whole factorisation + solve is ONE launch that reads the row count on the device (pre3_cholp.hip)
                if (cholp_usable(c, round_up(2 * c->m, NB) / NB)) {
                    // pre3_step: the rescue stage and the HI update ride in the same launch (mono_slam.m:184-187 as panel nrb of this factorisation)
                    CholpTailReq req{ c->tail_chi2, c->seq_collect + 1 };
                    c->tail_launched = false;
                    PRE3_TRY(launch_cholp(c, -1, round_up(2 * c->m, NB) / NB, -1, PRE3_X_K_KM1, c->tail_want ? &req : nullptr));
                    if (c->tail_launched) ++c->seq_collect;
                    c->cholp_done = true;
                }
                else PRE3_TRY(launch_chol_first_spec(c, c->m));
                first_done = true;
            }
        }
        PRE3_TRY(wait_mail(c, 8, c->seq_select)); n_li = c->mail_host[4];
        PRE3_CHECK(!c->shard_round || c->mail_host[11] == 0, PRE3_E_COMM, "sharded RANSAC: a rank failed before the collective of the round this update follows");
    }
    // (a launch that found no rows on the device returned at once: the tail has not run either)
    c->tail_done = first_done && c->cholp_done && c->tail_launched && n_li > 0;
    c->tail_launched = false;
    return update_selected(c, PRE3_X_K_KM1, n_li, c->sel_rows, gathered, first_done);
}

int pre3_rescue(pre3_ctx *c, double chi2, int32_t *hi_mask)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->p_which == PRE3_X_K_K && c->x_valid[PRE3_X_K_K], PRE3_E_STATE, "pre3_rescue: needs (x_k_k, p_k_k), i.e. after the LI update");
    if (c->N) {
        if (c->rescue_projected) PRE3_TRY(launch_innovation(c, 1, chi2));          // h / H at x_k_k came with the K9 launch
        else PRE3_TRY(launch_project_innovation(c, PRE3_X_K_K, 0, 1, chi2));
    }
    c->rescue_projected = false;
    c->hi_from_host = -1; c->hi_kernel = true;
    if (hi_mask) {
        PRE3_TRY(stream_drain(c, __func__));
        if (c->m) PRE3_HIP(hipMemcpy(hi_mask, c->hi_meas, sizeof(int32_t) * c->m, hipMemcpyDeviceToHost));
    }
    return PRE3_OK;
}

int pre3_update_hi(pre3_ctx *c)
{
    PRE3_TRY(check_ctx(c));
    int n_hi = 0;
    const bool tail_done = c->tail_done, was_fused = c->hi_fused;
    c->tail_done = false; c->hi_fused = false;                      // (before anything can return)
    if (c->hi_from_host >= 0) n_hi = c->hi_from_host;
    else if (c->hi_kernel) {
        PRE3_TRY(wait_mail(c, 9, c->seq_collect)); n_hi = c->mail_host[5];
        // the collection stage also brings the device's error words: what went wrong in this step's launches fails THIS call -- once: the words
        // are cleared with the report, so that a context that installs a fresh state (pre3_set_state) works again
        const int e_wait = c->mail_host[7], e_npd = c->mail_host[6];
        if (e_wait != 0 || e_npd != 0) {
            c->mail_host[6] = 0; c->mail_host[7] = 0;
            (void)hipMemsetAsync(c->stats + 6, 0, sizeof(int32_t) * 2, c->stream);
        }
        PRE3_CHECK(e_wait == 0, PRE3_E_HIP, "a device-side wait on another workgroup gave up (counter never arrived): results are invalid");
        PRE3_CHECK(e_npd == 0, PRE3_E_NUMERIC, "innovation covariance S is not positive definite");
    }
    if (tail_done) {
        // The persistent launch of the LI update has run the rescue stage and (up to 32 landmarks) the HI update as well (pre3_cholp.hip, CpTail):
        // P holds P - W'W - W~'W~ and update.m:42-46 of BOTH updates is one pending rows / columns 3..6 pass (params[16..] = params[96..] = J2 J1,
        // or J1 alone when nothing was updated).  More than 32: that pass now (J1), then the general path.
        c->hp_all_valid = false;
        if (c->hi_from_host < 0 && n_hi <= 32) {
            if (c->leave_jn_to_predict) c->jn_pending = true;
            else PRE3_TRY(launch_jnorm(c, 0));
            return PRE3_OK;
        }
        PRE3_TRY(launch_jnorm(c, 0));
        return update_selected(c, PRE3_X_K_K, n_hi, c->sel_rows);
    }
    if (was_fused) {
        // pre3_step sent the collection and the update out as one device-driven pair of launches (k_hi_fused + its down-date): up to 64
        // landmarks (two panels) are done, only the Jnorm pass of update.m:42-46 is left; more than that take the general path now
        const bool pend_launched = c->hi_pend_launched;
        c->hi_pend_launched = false;
        if (c->hi_from_host < 0 && n_hi <= hi_fused_max(c)) {
            if (n_hi > 0) {
                c->hp_all_valid = false;
                // PRE3_OPT_PEND_HI: k_hi_fused's down-date was not launched -- from here on P stands for P - W~'W~ (2 n_hi rows) until somebody takes it
                if (pend_launched) {
                    c->pend_rows = 2 * n_hi;
                    // (two panels of pending rows cost the next H*P launch ~19 us more, one panel ~5: sending the two-panel ones out at once -- PRE3_PEND_MAX_ROWS=64 --
                    //  measured 5922 against 5954 steps/s: the launch they then need costs as much)
                    static const int pend_max = getenv("PRE3_PEND_MAX_ROWS") ? atoi(getenv("PRE3_PEND_MAX_ROWS")) : 2 * NB;
                    if (c->pend_rows > pend_max) PRE3_TRY(pend_flush(c));
                }
                if (c->leave_jn_to_predict) c->jn_pending = true;
                else PRE3_TRY(launch_jnorm(c, 0));              // (flushes the pending rows first: the pass reads P)
            }
            return PRE3_OK;
        }
        // (more than k_hi_fused takes: it has written nothing -- no W~, no x-update --, the general path follows)
    }
    return update_selected(c, PRE3_X_K_K, n_hi, c->sel_rows);
}

int pre3_update_all(pre3_ctx *c)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->measurements_set && c->projected, PRE3_E_STATE, "pre3_update_all: needs projection and measurements");
    return update_selected(c, PRE3_X_K_KM1, c->m, nullptr);
}

int pre3_get_flags(pre3_ctx *c, int32_t *li, int32_t *hi)
{
    PRE3_TRY(check_ctx(c));
    PRE3_TRY(stream_drain(c, __func__));
    if (c->m == 0) return PRE3_OK;
    if (li) PRE3_HIP(hipMemcpy(li, c->li_meas, sizeof(int32_t) * c->m, hipMemcpyDeviceToHost));
    if (hi) PRE3_HIP(hipMemcpy(hi, c->hi_meas, sizeof(int32_t) * c->m, hipMemcpyDeviceToHost));
    return PRE3_OK;
}

int pre3_set_flags(pre3_ctx *c, const int32_t *li, const int32_t *hi)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->measurements_set, PRE3_E_STATE, "pre3_set_flags: no measurements");
    PRE3_TRY(stream_drain(c, __func__));
    int m = c->m, N = c->N;
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t *src = pass == 0 ? li : hi;
        if (!src) continue;
        std::vector<int32_t> lmflag(N ? N : 1, 0), sel;
        for (int j = 0; j < m; ++j) { lmflag[c->meas_host[j]] = src[j] ? 1 : 0; if (src[j]) sel.push_back(j); }
        PRE3_HIP(hipMemcpy(pass == 0 ? c->lm.li : c->lm.hi, lmflag.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
        if (m) PRE3_HIP(hipMemcpy(pass == 0 ? c->li_meas : c->hi_meas, src, sizeof(int32_t) * m, hipMemcpyHostToDevice));
        if (!sel.empty()) PRE3_HIP(hipMemcpy(c->sel_rows, sel.data(), sizeof(int32_t) * sel.size(), hipMemcpyHostToDevice));
        int32_t cnt = (int32_t)sel.size();
        PRE3_HIP(hipMemcpy(c->stats + (pass == 0 ? 4 : 5), &cnt, sizeof(int32_t), hipMemcpyHostToDevice));
        if (pass == 0) c->li_from_host = cnt; else c->hi_from_host = cnt;
    }
    return PRE3_OK;
}

// mono_slam.m:178-187 behind the prediction and the IC search: RANSAC, LI update, rescue, HI update, every launch sized on the device.
// hyp: the draw table -- the inbox's own copy when it was shipped with the measurements (pre3_step), the caller's otherwise.
static int step_back(pre3_ctx *c, int m, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit, double chi2, int32_t stats[8])
{
    int32_t st[8] = { -1, 0, 0, 0, 0, 0, 0, 0 };
    bool ran = false;
    if (m >= k && m > 0) {
        // mono_slam.m:178; the statistics are read after pre3_update_li's poll of the same mailbox
        c->defer_select = true;                                     // the selection stage rides in the LI gather's launch (pre3_update_li below)
        const int rc_r = pre3_ransac(c, n_draw, k, hyp, threshold, early_exit, nullptr, nullptr, nullptr);
        c->defer_select = false;
        if (c->ride_innovation) {                                   // the H*P launch did not go out (error before it): S_i on its own, flags cleared
            c->ride_innovation = false;
            PRE3_TRY(launch_innovation(c, 0, 0.0, true));
        }
        PRE3_TRY(rc_r);
        ran = true;
    }
    static const int ride_rescue = getenv("PRE3_RIDE_RESCUE") ? atoi(getenv("PRE3_RIDE_RESCUE")) : 1;      // 0: projection + gate as one launch of their own (A/B)
    c->ride_rescue_projection = ride_rescue != 0;                   // the rescue's projection rides in the LI update's K9 launch
    {
        c->tail_want = c->step_tail && hi_fused_usable(c); c->tail_chi2 = chi2;      // ... or, with the whole rescue stage and the HI update, in the persistent launch itself
        // ... or projection AND chi2 gate in the Jnorm pass's launch, when the persistent launch's consumers leave rows 3..6 of P behind (GateRide)
        c->want_gate_ride = ride_rescue != 0 && !c->tail_want && hi_fused_usable(c); c->rescue_chi2 = chi2; c->rescue_gated = false;
        const int rc_li = pre3_update_li(c);                        // mono_slam.m:181
        c->tail_want = false; c->want_gate_ride = false;
        c->ride_rescue_projection = false;                          // (also on failure: a later K9 launch must not carry the riders)
        if (rc_li != PRE3_OK) { c->rescue_projected = false; c->rescue_gated = false; c->tail_done = false; return rc_li; }
    }
    if (ran) for (int i = 0; i < 4; ++i) st[i] = c->mail_host[i];
    if (c->tail_done) {
        // mono_slam.m:184 + :187 went out with the LI update's launch: the count arrives with mailbox word 9 (pre3_update_hi)
        c->rescue_projected = false; c->proj_with_jnorm = false;
        c->hi_from_host = -1; c->hi_kernel = true; c->hi_fused = false;
    } else if (hi_fused_usable(c)) {
        // mono_slam.m:184 + :187 without the host in between: the chi2 gate, then the collection and the HI update of up to 32 landmarks as ONE
        // launch that reads the count on the device, and its down-date behind it (pre3_update.hip, k_hi_fused)
        PRE3_CHECK(c->p_which == PRE3_X_K_K && c->x_valid[PRE3_X_K_K], PRE3_E_STATE, "pre3_step: the LI update did not leave (x_k_k, p_k_k)");
        if (c->rescue_gated) { /* the gate rode with the Jnorm pass */ }
        else if (c->rescue_projected) PRE3_TRY(launch_innovation(c, 1, chi2, false, false));
        else PRE3_TRY(launch_project_innovation(c, PRE3_X_K_K, 0, 1, chi2, false));
        c->rescue_projected = false; c->rescue_gated = false;
        c->hi_from_host = -1; c->hi_kernel = true;
        PRE3_TRY(launch_hi_fused(c, ++c->seq_collect));
        c->hi_fused = true;
    } else
    PRE3_TRY(pre3_rescue(c, chi2, nullptr));                        // mono_slam.m:184
    if (c->defer_hi) c->hi_pending = true;                          // mono_slam.m:187, completed at the next call on this context
    else PRE3_TRY(pre3_update_hi(c));                               // mono_slam.m:187
    st[4] = c->li_from_host >= 0 ? c->li_from_host : (c->li_kernel ? c->mail_host[4] : 0);
    st[5] = c->defer_hi ? c->last_n_hi : (c->hi_from_host >= 0 ? c->hi_from_host : (c->hi_kernel ? c->mail_host[5] : 0));
    st[7] = c->defer_hi ? 1 : 0;          // 1: st[5] is the HI count of the PREVIOUS step (this step's is still on the device)
    if (stats) for (int i = 0; i < 8; ++i) stats[i] = st[i];
    return PRE3_OK;
}

int pre3_step(pre3_ctx *c, const double u[7], int m, const int32_t *meas_idx, const double *z, int n_draw, int k, const int32_t *hyp,
              double threshold, int early_exit, double chi2, int32_t stats[8])
{
    // the previous step's deferred HI update is completed here; its rows/cols 3..6 <- Jn pass (update.m:42-46) is left to the prediction's
    // launch below (one launch less per step; PRE3_FUSE_JN=0: as its own launch).  Any return before that launch flushes it.
    static const int fuse_jn_env = getenv("PRE3_FUSE_JN") ? atoi(getenv("PRE3_FUSE_JN")) : 1;
    if (c) c->leave_jn_to_predict = fuse_jn_env && c->hi_pending;
    // PRE3_OPT_PEND_HI: this call's own launches take a pending HI down-date along (prediction, H*P + S_i, the LI update's consumers); whatever of it
    // cannot -- and every call made from in here that reads P some other way -- flushes it first (pend_flush in the launchers)
    struct PendKeep { pre3_ctx *c; ~PendKeep() { if (c) c->pend_keep = false; } } pend_keep{ c };
    if (c) c->pend_keep = c->pend_opt && c->dtype == PRE3_F32 && m >= k && m > 0 && c->N > 0;
    {
        const int rc0 = check_ctx(c);
        if (c) c->leave_jn_to_predict = false;
        if (rc0 != PRE3_OK) { if (c && c->jn_pending) { c->jn_pending = false; (void)launch_jnorm(c, 0); } return rc0; }
    }
    struct JnFlush { pre3_ctx *c; ~JnFlush() { if (c->jn_pending) { c->jn_pending = false; (void)launch_jnorm(c, 0); } } } jn_flush{ c };
    static const bool trace = getenv("PRE3_STEP_TRACE") != nullptr;     // host-side stage clock (debug): where the host spends a step
    static double acc[8], t_prev_end = 0; static int nacc = 0;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
    double t0 = trace ? now() : 0, t1 = 0, t5 = 0;
    PRE3_CHECK(u != nullptr, PRE3_E_ARG, "pre3_step: null u");
    PRE3_CHECK(c->have_cam, PRE3_E_STATE, "pre3_step: camera not set");
    PRE3_CHECK(c->x_valid[PRE3_X_K_K] && c->p_which == PRE3_X_K_K, PRE3_E_STATE, "pre3_step: needs (x_k_k, p_k_k) on the device");
    PRE3_CHECK(m == 0 || (meas_idx && z), PRE3_E_ARG, "pre3_step: null measurement pointers");
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph && k >= 1 && k <= MAXK && hyp, PRE3_E_ARG, "pre3_step: bad hypothesis table");
    // matching_sift_based.m:131-134 outcome (+ the draws) into the pinned inbox; it crosses PCIe in one extra block of the prediction's
    // launch (nothing in that launch reads it), so the copy costs neither a launch nor stream time.  Flags cleared by k_innovation.
    size_t inbox_bytes = 0;
    PRE3_TRY(install_measurements(c, m, meas_idx, z, hyp, n_draw * k, c->N > 0, false, &inbox_bytes));
    // mono_slam.m:153 + search_IC_matches.m:31-32: prediction, with the projection of every landmark at x_k_km1 riding in the
    // same launch; then search_IC_matches.m:33-44 (S_i), which also clears the previous frame's inlier flags
    {
        static const int ride_proj = getenv("PRE3_RIDE_PROJ") ? atoi(getenv("PRE3_RIDE_PROJ")) : 1;      // 0: the projection as its own launch (A/B)
        const int rc_p = launch_predict_impl(c, u, ride_proj != 0, (inbox_bytes + 15) / 16, ++c->seq_inbox);
        c->inbox_pending = rc_p == PRE3_OK;
        if (rc_p != PRE3_OK) { c->measurements_set = false; return rc_p; }
        if (!ride_proj && c->N) PRE3_TRY(launch_project(c, PRE3_X_K_KM1, 1));
    }
    c->x_valid[PRE3_X_K_KM1] = true; c->p_which = PRE3_X_K_KM1; c->hp_all_valid = false;
    c->projected = true;
    // S_i (which also clears last frame's inlier flags) rides in the H*P launch of the RANSAC stage when there is one
    static const int ride_env = getenv("PRE3_RIDE_INNOV") ? atoi(getenv("PRE3_RIDE_INNOV")) : 1;
    c->ride_innovation = ride_env && c->N > 0 && m >= k && m > 0;
    if (c->N && !c->ride_innovation) PRE3_TRY(launch_innovation(c, 0, 0.0, true));
    c->innovated = true;
    if (trace) t1 = now();
    const int rc_back = step_back(c, m, n_draw, k, (const int32_t *)(c->inbox_host + c->off_hyp), threshold, early_exit, chi2, stats);
    if (trace) {
        t5 = now();
        acc[0] += t1 - t0; acc[1] += t5 - t1;
        if (t_prev_end > 0) acc[5] += t0 - t_prev_end;
        t_prev_end = t5;
        if (++nacc == 100) {
            fprintf(stderr, "[pre3 step trace, us] predict+project+innov launches %.1f | ransac .. HI update (polls + launches) %.1f | caller between steps %.1f\n",
                    acc[0] / nacc, acc[1] / nacc, acc[5] / nacc);
            nacc = 0; for (double &a2 : acc) a2 = 0;
        }
    }
    return rc_back;
}

/* mono_slam.m:153-162 + :199 -- the 'PURE_EKF' branch (config_file.m:21): prediction, projection + Jacobians + S_i of every landmark, then ONE
 * update with every individually compatible measurement (ekf_update_all.m:46-62), as one call: the projection and the inbox ride in the
 * prediction's launch, S_i and the flag clearing in the H*P launch -- four launches fewer than the call-by-call sequence
 * (pre3_predict, pre3_project, pre3_innovation, pre3_set_measurements, pre3_update_all), the same arithmetic. */
int pre3_step_all(pre3_ctx *c, const double u[7], int m, const int32_t *meas_idx, const double *z)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(u != nullptr, PRE3_E_ARG, "pre3_step_all: null u");
    PRE3_CHECK(c->have_cam, PRE3_E_STATE, "pre3_step_all: camera not set");
    PRE3_CHECK(c->x_valid[PRE3_X_K_K] && c->p_which == PRE3_X_K_K, PRE3_E_STATE, "pre3_step_all: needs (x_k_k, p_k_k) on the device");
    PRE3_CHECK(m == 0 || (meas_idx && z), PRE3_E_ARG, "pre3_step_all: null measurement pointers");
    size_t inbox_bytes = 0;
    PRE3_TRY(install_measurements(c, m, meas_idx, z, nullptr, 0, c->N > 0 && m > 0, false, &inbox_bytes));
    {
        const int rc_p = launch_predict_impl(c, u, true, (inbox_bytes + 15) / 16, ++c->seq_inbox);
        c->inbox_pending = rc_p == PRE3_OK;
        if (rc_p != PRE3_OK) { c->measurements_set = false; return rc_p; }
    }
    c->x_valid[PRE3_X_K_KM1] = true; c->p_which = PRE3_X_K_KM1; c->hp_all_valid = false;
    c->projected = true;
    c->ride_innovation = c->N > 0 && m > 0;          // S_i (and the clearing of last frame's flags) in the update's H*P launch
    if (c->N && !c->ride_innovation) PRE3_TRY(launch_innovation(c, 0, 0.0, true));
    c->innovated = true;
    const int rc_u = update_selected(c, PRE3_X_K_KM1, c->m, nullptr);
    if (c->ride_innovation) {                        // (the H*P launch did not go out: S_i on its own)
        c->ride_innovation = false;
        if (rc_u == PRE3_OK) PRE3_TRY(launch_innovation(c, 0, 0.0, true));
    }
    return rc_u;
}

/* The same behind a prediction and an IC search the caller has already run (mono_slam.m:153 ekf_prediction, :159 search_IC_matches +
 * matching_sift_based, e.g. pre3_predict + pre3_ic_search): the installed measurements are used. */
int pre3_step_predicted(pre3_ctx *c, int n_draw, int k, const int32_t *hyp, double threshold, int early_exit, double chi2, int32_t stats[8])
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(c->x_valid[PRE3_X_K_KM1] && c->p_which == PRE3_X_K_KM1, PRE3_E_STATE, "pre3_step_predicted: needs the predicted estimate (pre3_predict)");
    PRE3_CHECK(c->measurements_set && c->projected && c->innovated, PRE3_E_STATE, "pre3_step_predicted: needs projection, S_i and measurements (pre3_ic_search, or pre3_project + pre3_innovation + pre3_set_measurements)");
    PRE3_CHECK(n_draw >= 1 && n_draw <= c->caph && k >= 1 && k <= MAXK && hyp, PRE3_E_ARG, "pre3_step_predicted: bad hypothesis table");
    c->ride_innovation = false;
    return step_back(c, c->m, n_draw, k, hyp, threshold, early_exit, chi2, stats);
}

// ---- stateless update.m drop-in ---------------------------------------------------------------------
int pre3_update_ell(int device, int dtype, int n, int r, const double *x, const double *P, int width, const int32_t *nnz, const int32_t *col,
                    const double *val, const double *R, const double *z, const double *h, double *x_out, double *P_out, double *K_out)
{
    PRE3_CHECK(n >= 13 && r >= 0 && x && P && x_out && P_out, PRE3_E_ARG, "pre3_update_ell: bad arguments");
    PRE3_CHECK(r == 0 || (nnz && col && val && z && h && width >= 1), PRE3_E_ARG, "pre3_update_ell: null row data");
    if (r == 0) {            // update.m:50-55
        if (x_out != x) memcpy(x_out, x, sizeof(double) * n);
        if (P_out != P) memcpy(P_out, P, sizeof(double) * (size_t)n * n);
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_error("no HIP device available (libpre3 has no CPU fallback)"); return PRE3_E_NODEVICE; }
        return PRE3_OK;
    }
    for (int a = 0; a < r; ++a) {
        PRE3_CHECK(nnz[a] >= 0 && nnz[a] <= width && nnz[a] <= ELLW, PRE3_E_ARG, "pre3_update_ell: row %d has %d non-zeros (max %d)", a, nnz[a], ELLW);
        for (int t = 0; t < nnz[a]; ++t) PRE3_CHECK(col[a * width + t] >= 0 && col[a * width + t] < n, PRE3_E_ARG, "pre3_update_ell: column index out of range in row %d", a);
    }
    // a throw-away context sized for this call: capacity in "landmarks" such that 13+6*cap >= n and 2*cap >= r
    int capL = std::max((n - 13 + 5) / 6, (r + 1) / 2);
    if (capL < 1) capL = 1;
    pre3_ctx *c = nullptr;
    PRE3_TRY(pre3_create(&c, device, dtype, capL, 1));
    int rc = PRE3_OK;
    do {
        c->n = n; c->N = 0;
        c->x_valid[PRE3_X_K_KM1] = true;
        if ((rc = pre3_set_state(c, PRE3_X_K_KM1, n, x, P)) != PRE3_OK) break;
        int r_pad = round_up(r, NB);
        std::vector<int32_t> hc((size_t)r_pad * ELLW, 0);
        std::vector<double> hv((size_t)r_pad * ELLW, 0.0), nu(r_pad, 0.0);
        for (int a = 0; a < r; ++a) {
            for (int t = 0; t < nnz[a]; ++t) { hc[(size_t)a * ELLW + t] = col[a * width + t]; hv[(size_t)a * ELLW + t] = val[a * width + t]; }
            nu[a] = z[a] - h[a];
        }
        if (hipMemcpy(c->row_col, hc.data(), sizeof(int32_t) * hc.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = PRE3_E_HIP; set_error("copy failed"); break; }
        if (dtype == PRE3_F64) {
            if (hipMemcpy(c->row_val, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = PRE3_E_HIP; break; }
        } else {
            std::vector<float> hf(hv.begin(), hv.end());
            if (hipMemcpy(c->row_val, hf.data(), sizeof(float) * hf.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = PRE3_E_HIP; break; }
        }
        if (hipMemcpy(c->row_nu, nu.data(), sizeof(double) * r_pad, hipMemcpyHostToDevice) != hipSuccess) { rc = PRE3_E_HIP; break; }
        if (R) {
            if ((rc = dmalloc_bytes(&c->Rdense, (size_t)r * r * c->esz)) != PRE3_OK) break;
            if (dtype == PRE3_F64) { if (hipMemcpy(c->Rdense, R, sizeof(double) * r * r, hipMemcpyHostToDevice) != hipSuccess) { rc = PRE3_E_HIP; break; } }
            else { std::vector<float> rf(R, R + (size_t)r * r); if (hipMemcpy(c->Rdense, rf.data(), sizeof(float) * rf.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = PRE3_E_HIP; break; } }
        }
        void *Kt = nullptr;
        if (K_out) { if ((rc = dmalloc_bytes(&Kt, (size_t)r_pad * c->ldw * c->esz)) != PRE3_OK) break; }
        rc = run_update(c, PRE3_X_K_KM1, r, R != nullptr, Kt);
        if (rc == PRE3_OK) { c->x_valid[PRE3_X_K_K] = true; c->p_which = PRE3_X_K_K; rc = pre3_get_state(c, PRE3_X_K_K, n, x_out, P_out); }
        if (rc == PRE3_OK && K_out) {
            // Kt (r x ldw, T) -> K_out (n x r column-major) : K(i,a) = Kt[a][i]
            if (dtype == PRE3_F64) {
                if (hipMemcpy2D(K_out, sizeof(double) * n, Kt, sizeof(double) * c->ldw, sizeof(double) * n, r, hipMemcpyDeviceToHost) != hipSuccess) rc = PRE3_E_HIP;
            } else {
                std::vector<float> kf((size_t)r * n);
                if (hipMemcpy2D(kf.data(), sizeof(float) * n, Kt, sizeof(float) * c->ldw, sizeof(float) * n, r, hipMemcpyDeviceToHost) != hipSuccess) rc = PRE3_E_HIP;
                else for (size_t i = 0; i < kf.size(); ++i) K_out[i] = kf[i];
            }
        }
        if (Kt) (void)hipFree(Kt);
    } while (0);
    pre3_destroy(c);
    return rc;
}

// ---- matcher ------------------------------------------------------------------------------------------
// Stateless drop-in for compute_hypothesis_support_fast.m:27 (caller: ransac_hypotheses.m:72).
int pre3_hypothesis_support(int device, int n, const double *xi, const pre3_cam *cam, const double *state_vector_pattern,
                            int n_id, const double *z_id, int n_euc, const double *z_euc, double threshold,
                            int32_t *support_out, int32_t *positions_li_inliers_id, int32_t *positions_li_inliers_euc)
{
    PRE3_CHECK(n >= 13 && xi && cam && state_vector_pattern && support_out, PRE3_E_ARG, "pre3_hypothesis_support: bad arguments");
    PRE3_CHECK(n_id >= 0 && n_euc >= 0 && (n_id == 0 || z_id) && (n_euc == 0 || z_euc), PRE3_E_ARG, "pre3_hypothesis_support: bad measurement arrays");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_error("no HIP device available (libpre3 has no CPU fallback)"); return PRE3_E_NODEVICE; }
    PRE3_CHECK(device >= 0 && device < nd, PRE3_E_ARG, "pre3_hypothesis_support: device %d of %d", device, nd);
    // xi(logical(state_vector_pattern(:,c))): the selected entries in state order, column c of the n x 4 column-major pattern
    std::vector<int32_t> idx[4];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < n; ++i) if (state_vector_pattern[(size_t)c * n + i] != 0.0) idx[c].push_back(i);
    // reshape(ri,3,n_id) etc. (:40-43, :84) fail in MATLAB on a count mismatch; so does this
    PRE3_CHECK((int)idx[0].size() == 3 * n_id && (int)idx[1].size() == 2 * n_id && (int)idx[2].size() == n_id, PRE3_E_ARG,
               "pre3_hypothesis_support: pattern selects %zu/%zu/%zu entries for %d inverse-depth measurements (needs 3/2/1 each)",
               idx[0].size(), idx[1].size(), idx[2].size(), n_id);
    PRE3_CHECK((int)idx[3].size() == 3 * n_euc, PRE3_E_ARG, "pre3_hypothesis_support: pattern selects %zu entries for %d cartesian measurements", idx[3].size(), n_euc);
    if (n_id + n_euc == 0) { *support_out = 0; return PRE3_OK; }                       // :29, both branches empty
    PRE3_HIP(hipSetDevice(device));
    std::vector<int32_t> out(1 + (size_t)n_id + n_euc);
    PRE3_TRY(run_hypothesis_support(n, xi, *cam, n_id, idx[0].data(), idx[1].data(), idx[2].data(), z_id, n_euc, idx[3].data(), z_euc, threshold, out.data()));
    *support_out = out[0];
    if (positions_li_inliers_id) for (int j = 0; j < n_id; ++j) positions_li_inliers_id[j] = out[1 + j];
    if (positions_li_inliers_euc) for (int j = 0; j < n_euc; ++j) positions_li_inliers_euc[j] = out[1 + n_id + j];
    return PRE3_OK;
}

// Stateless drop-in for `[X_km1_k, P_km1_k] = predict_state_and_covariance(X_k, P_k, type, SD_A, SD_alpha)`
// (predict_state_and_covariance.m:27, caller @ekf_filter/ekf_prediction.m:29) with the odometry increment made explicit.
struct DenseCtx { pre3_ctx *c = nullptr; int device = 0, dtype = 0, capL = 0; };
static DenseCtx g_dense[4];
static std::mutex g_dense_mu;

int pre3_predict_dense(int device, int dtype, int n, const double *x, const double *P, const double u[7], double *x_out, double *P_out)
{
    PRE3_CHECK(n >= 13 && (n - 13) % 3 == 0 && x && P && u && x_out && P_out, PRE3_E_ARG, "pre3_predict_dense: bad arguments (n = 13 + 6 N_id + 3 N_euc)");
    int capL = std::max((n - 13 + 5) / 6, 1);
    // ekf_prediction.m reaches this entry every frame: the context (P at capacity, the update's work buffers, pinned inbox / mailbox -- more
    // than a GB allocated and cleared at n = 12013) is kept between calls, one per (device, dtype), grown when a larger state arrives and
    // released by pre3_release_scratch().  Calls are serialised on it.
    std::lock_guard<std::mutex> lk(g_dense_mu);
    DenseCtx *slot = nullptr;
    for (DenseCtx &d : g_dense) if (d.c && d.device == device && d.dtype == dtype) slot = &d;
    if (slot && slot->capL < capL) { pre3_destroy(slot->c); slot->c = nullptr; slot = nullptr; }
    if (!slot) {
        for (DenseCtx &d : g_dense) if (!d.c) { slot = &d; break; }
        if (!slot) { slot = &g_dense[0]; pre3_destroy(slot->c); slot->c = nullptr; }
        const int cap = capL + capL / 8;                                     // a little headroom: the map grows by a few landmarks per frame
        PRE3_TRY(pre3_create(&slot->c, device, dtype, cap, 1));
        slot->device = device; slot->dtype = dtype; slot->capL = cap;
        // (this context never factorises: it must not count against the two persistent-factorisation contexts a device serves, pre3_cholp.hip)
        slot->c->chol_persist = false;
        if (slot->c->cholp_counted) { cholp_context_count(device, -1); slot->c->cholp_counted = false; }
    }
    pre3_ctx *c = slot->c;
    int rc = PRE3_OK;
    do {
        c->n = n; c->N = 0;                       // the prediction touches the 13 camera entries and rows/columns 1:13 only: no landmark table needed
        if ((rc = pre3_set_state(c, PRE3_X_K_K, n, x, P)) != PRE3_OK) break;
        if ((rc = pre3_predict(c, u)) != PRE3_OK) break;
        rc = pre3_get_state(c, PRE3_X_K_KM1, n, x_out, P_out);
    } while (0);
    if (rc != PRE3_OK) { pre3_destroy(slot->c); slot->c = nullptr; }         // never reuse a context that failed half-way
    return rc;
}

int pre3_release_scratch(void)
{
    release_scratch();
    std::lock_guard<std::mutex> lk(g_dense_mu);
    for (DenseCtx &d : g_dense) if (d.c) { pre3_destroy(d.c); d.c = nullptr; }
    return PRE3_OK;
}

int pre3_siftmatch_merge(int cls, int G, int K1, const double *best, const double *second, const int32_t *arg, double thresh_d,
                         double *pairs_out, double *score_out, int *M_out)
{
    PRE3_CHECK(G >= 1 && K1 >= 0 && M_out, PRE3_E_ARG, "pre3_siftmatch_merge: bad arguments");
    PRE3_CHECK(K1 == 0 || (best && second && arg && pairs_out), PRE3_E_ARG, "pre3_siftmatch_merge: null pointer");
    const float thresh = (float)thresh_d;      // the gateway passes its double into a float parameter (siftmatch.c:208-216)
    int M = 0;
    for (int k1 = 0; k1 < K1; ++k1) {
        double B = 0, S2 = 0; int K = -1;
        for (int g = 0; g < G; ++g) {
            double ob = best[(size_t)g * K1 + k1], os = second[(size_t)g * K1 + k1]; int ok = arg[(size_t)g * K1 + k1];
            if (ok < 0) continue;
            if (K < 0) { B = ob; S2 = os; K = ok; continue; }
            if (ob < B || (ob == B && ok < K)) { S2 = os < B ? os : B; B = ob; K = ok; }
            else { S2 = ob < S2 ? ob : S2; }
        }
        if (K < 0) continue;
        // Lowe's ratio test in float (siftmatch.c:122); integer classes convert int -> float
        float fb, fs;
        if (cls == 1) { fb = (float)B; fs = (float)S2; }
        else if (cls >= 2) { fb = (float)(int)B; fs = (float)(int)S2; }
        else { fb = (float)B; fs = (float)S2; }
        if (thresh * fb <= fs) {
            pairs_out[2 * M] = k1 + 1; pairs_out[2 * M + 1] = K + 1;
            if (score_out) score_out[M] = B;
            ++M;
        }
    }
    *M_out = M;
    return PRE3_OK;
}

int pre3_siftmatch_partial(int device, int cls, int ND, int K1, const void *L1, int K2_local, const void *L2_local, int k2_offset,
                           double *best, double *second, int32_t *arg)
{
    return match_partial(device, cls, ND, K1, L1, K2_local, L2_local, k2_offset, best, second, arg);
}

static int siftmatch_any(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2, double thresh, double *pairs_out,
                         double *score_out, int *M_out)
{
    PRE3_CHECK(M_out != nullptr, PRE3_E_ARG, "siftmatch: null M_out");
    *M_out = 0;
    PRE3_CHECK(ND > 0 && K1 >= 0 && K2 >= 0, PRE3_E_ARG, "siftmatch: bad sizes");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_error("no HIP device available (libpre3 has no CPU fallback)"); return PRE3_E_NODEVICE; }
    if (K1 == 0) return PRE3_OK;
    std::vector<double> b(K1), s(K1); std::vector<int32_t> a(K1);
    PRE3_TRY(match_partial(device, cls, ND, K1, L1, K2, L2, 0, b.data(), s.data(), a.data()));
    return pre3_siftmatch_merge(cls, 1, K1, b.data(), s.data(), a.data(), thresh, pairs_out, score_out, M_out);
}

int pre3_siftmatch_f64(int device, int ND, int K1, const double *L1, int K2, const double *L2, double thresh, double *pairs_out, double *score_out, int *M_out)
{ return siftmatch_any(device, 0, ND, K1, L1, K2, L2, thresh, pairs_out, score_out, M_out); }
int pre3_siftmatch_f32(int device, int ND, int K1, const float *L1, int K2, const float *L2, double thresh, double *pairs_out, double *score_out, int *M_out)
{ return siftmatch_any(device, 1, ND, K1, L1, K2, L2, thresh, pairs_out, score_out, M_out); }
int pre3_siftmatch_u8(int device, int ND, int K1, const uint8_t *L1, int K2, const uint8_t *L2, double thresh, double *pairs_out, double *score_out, int *M_out)
{ return siftmatch_any(device, 2, ND, K1, L1, K2, L2, thresh, pairs_out, score_out, M_out); }
int pre3_siftmatch_i8(int device, int ND, int K1, const int8_t *L1, int K2, const int8_t *L2, double thresh, double *pairs_out, double *score_out, int *M_out)
{ return siftmatch_any(device, 3, ND, K1, L1, K2, L2, thresh, pairs_out, score_out, M_out); }

int pre3_knn_f64(int device, int D, int N, const double *data, int M, const double *query, int k, double *ids_out, double *dist_out)
{
    PRE3_CHECK(data && (M == 0 || (query && ids_out && dist_out)), PRE3_E_ARG, "pre3_knn_f64: null pointer");
    return knn_run(device, D, N, data, M, query, k, ids_out, dist_out);
}

// ---- measurement hooks ---------------------------------------------------------------------------------
int pre3_timer_start(pre3_ctx *c)
{
    PRE3_TRY(check_ctx(c));
    PRE3_HIP(hipEventRecord(c->t0, c->stream));
    return PRE3_OK;
}

int pre3_timer_stop(pre3_ctx *c, double *ms_out)
{
    PRE3_TRY(check_ctx(c));
    PRE3_HIP(hipEventRecord(c->t1, c->stream));
    PRE3_HIP(hipEventSynchronize(c->t1));
    float ms = 0;
    PRE3_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    if (ms_out) *ms_out = ms;
    return PRE3_OK;
}

int pre3_kernel_timing(pre3_ctx *c, int enable)
{
    PRE3_TRY(check_ctx(c));
    c->kt.enabled = enable != 0; c->kt.every = enable > 1 ? enable : 1; c->kt.seen = 0; c->kt.used = 0; c->kt.flops = 0; c->kt.bytes = 0;
    c->kt.fused = 0; c->kt.fact_flops = 0; c->kt.pending = false;
    return PRE3_OK;
}

int pre3_kernel_timing_read(pre3_ctx *c, int *launches_out, double *total_ms_out, double *flops_out, double *bytes_out)
{
    PRE3_TRY(check_ctx(c));
    PRE3_TRY(stream_drain(c, __func__));
    double tot = 0;
    for (int i = 0; i + 1 < c->kt.used; i += 2) { float ms = 0; PRE3_HIP(hipEventElapsedTime(&ms, c->kt.ev[i], c->kt.ev[i + 1])); tot += ms; }
    if (launches_out) *launches_out = c->kt.used / 2;
    if (total_ms_out) *total_ms_out = tot;
    if (flops_out) *flops_out = c->kt.flops;
    if (bytes_out) *bytes_out = c->kt.bytes;
    c->kt.used = 0; c->kt.flops = 0; c->kt.bytes = 0;
    return PRE3_OK;
}

int pre3_kernel_timing_info(pre3_ctx *c, int *fused_launches_out, double *fact_flops_out)
{
    PRE3_TRY(check_ctx(c));
    if (fused_launches_out) *fused_launches_out = c->kt.fused;
    if (fact_flops_out) *fact_flops_out = c->kt.fact_flops;
    c->kt.fused = 0; c->kt.fact_flops = 0;
    return PRE3_OK;
}

int pre3_bench_downdate(pre3_ctx *c, int r, int reps, double *ms_per_launch_out)
{
    PRE3_TRY(check_ctx(c));
    PRE3_CHECK(r >= 1 && round_up(r, NB) <= c->rcap && reps >= 1, PRE3_E_ARG, "pre3_bench_downdate: bad r/reps");
    PRE3_CHECK(c->n > 0, PRE3_E_STATE, "pre3_bench_downdate: no map/state set");
    int r_pad = round_up(r, NB);
    PRE3_TRY(launch_fill_w(c, r_pad));
    bool was = c->kt.enabled; c->kt.enabled = false;
    c->dd_done = 0; c->x_done = false;
    PRE3_TRY(launch_downdate(c, r, c->W));                         // (splits W into its bf16 planes on the way)
    PRE3_HIP(hipEventRecord(c->t0, c->stream));
    for (int i = 0; i < reps; ++i) {
        if (c->k9_b3 && c->dtype == PRE3_F32 && c->Wp != nullptr) c->split_rows = r_pad;      // the planes are there: time the down-date alone, as it runs behind a factorisation
        PRE3_TRY(launch_downdate(c, r, c->W));
    }
    PRE3_HIP(hipEventRecord(c->t1, c->stream));
    PRE3_HIP(hipEventSynchronize(c->t1));
    float ms = 0; PRE3_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    if (ms_per_launch_out) *ms_per_launch_out = ms / reps;
    c->kt.enabled = was;
    return PRE3_OK;
}

// ---- device-resident database shard of the sharded matcher (uint8 class; DESIGN.md "multi-GPU", C2)
int pre3_match_shard_create_cls(pre3_match_shard **out, int device, int cls, int ND, int K1, const void *L1, int K2_local, const void *L2_local, int k2_offset)
{
    PRE3_CHECK(out != nullptr, PRE3_E_ARG, "pre3_match_shard_create: null output");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_error("no HIP device available (libpre3 has no CPU fallback)"); return PRE3_E_NODEVICE; }
    void *h = match_shard_create(device, cls, ND, K1, L1, K2_local, L2_local, k2_offset);
    if (!h) return PRE3_E_ARG;
    *out = (pre3_match_shard *)h;
    return PRE3_OK;
}
int pre3_match_shard_create(pre3_match_shard **out, int device, int ND, int K1, const uint8_t *L1, int K2_local, const uint8_t *L2_local, int k2_offset)
{
    return pre3_match_shard_create_cls(out, device, 2, ND, K1, L1, K2_local, L2_local, k2_offset);
}
int pre3_match_shard_run(pre3_match_shard *s, void **partial_dev, int *n_doubles) { return s ? match_shard_run(s, partial_dev, n_doubles) : PRE3_E_ARG; }
int pre3_match_shard_merge(pre3_match_shard *s, int G, const void *gathered_dev, double thresh, double *pairs_out, double *score_out, int *M_out)
{
    return s ? match_shard_merge(s, G, gathered_dev, thresh, pairs_out, score_out, M_out) : PRE3_E_ARG;
}
int pre3_match_shard_set_comm(pre3_match_shard *s, pre3_comm *comm) { return s ? match_shard_set_comm(s, comm) : PRE3_E_ARG; }
int pre3_match_shard_match(pre3_match_shard *s, double thresh, double *pairs_out, double *score_out, int *M_out)
{
    return s ? match_shard_match(s, thresh, pairs_out, score_out, M_out) : PRE3_E_ARG;
}
int pre3_match_shard_destroy(pre3_match_shard *s) { if (s) match_shard_destroy(s); return PRE3_OK; }
int pre3_match_shard_test_stall(pre3_match_shard *s, int release)
{
    PRE3_CHECK(test_hooks_on(), PRE3_E_STATE, "pre3_match_shard_test_stall: test hooks are off (PRE3_TEST_HOOKS=1 enables them)");
    return s ? match_shard_test_stall(s, release) : PRE3_E_ARG;
}

// matcher roofline/bench probe (inputs resident in HBM); not part of the reference-shaped API
PRE3_API void *pre3_match_bench_create(int device, int ND, int K1, const uint8_t *L1, int K2, const uint8_t *L2)
{
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return nullptr; }
    return match_bench_create(2, ND, K1, L1, K2, L2);
}
// the same probe for any class (0 double, 1 float, 2 uint8); pre3_match_bench_info: [route (0 exact kernels, 1 int8 MFMA, 2 bf16 rank +
// exact re-evaluation), queries scanned in full, candidates re-evaluated] of the last run
PRE3_API void *pre3_match_bench_create_cls(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2)
{
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return nullptr; }
    return match_bench_create(cls, ND, K1, L1, K2, L2);
}
PRE3_API int pre3_match_bench_info(void *h, int32_t info[3]) { return h ? match_bench_info(h, info) : PRE3_E_ARG; }
PRE3_API int pre3_match_bench_run(void *h, int reps, double *ms_per) { return h ? match_bench_run(h, reps, ms_per) : PRE3_E_ARG; }
PRE3_API int pre3_match_bench_fetch(void *h, double *best, double *second, int32_t *arg) { return h ? match_bench_fetch(h, best, second, arg) : PRE3_E_ARG; }
PRE3_API void pre3_match_bench_destroy(void *h) { if (h) match_bench_destroy(h); }

}  // extern "C"
