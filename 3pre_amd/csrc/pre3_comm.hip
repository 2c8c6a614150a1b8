// RCCL communicator of libpre3: the collectives of the two stages that shard are enqueued by the library itself, on the
// stream its kernels run on -- scoring kernel, ncclAllReduce, selection kernel back to back; distance kernel, ncclAllGather, merge kernel
// back to back -- so that a sharded round has ONE host wait (the pinned mailbox), not a synchronisation either side of the collective.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy a host program such as PyTorch has already loaded if there is one): libpre3
// neither needs the library to load nor pages its code objects in for single-GPU use.  One process per GPU; the 128-byte ncclUniqueId made
// by pre3_comm_unique_id on rank 0 is handed to the other ranks by the host program (torch.distributed, MPI, a file: 3pre_amd/comm.py).
#include "pre3_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>
#include <atomic>
#include <chrono>
#include <thread>

namespace pre3 {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char where[256] = { 0 };
};

static Rccl g_rccl;
static std::mutex g_rccl_mu;

static const Rccl *rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return &g_rccl;
    // the copy already in the process first (a second RCCL beside the host program's would double every per-process resource)
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    const char *override_path = getenv("PRE3_RCCL_LIB");
    if (override_path && *override_path) h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0)); if (h) break; }
    if (!h) { set_error("RCCL not found (dlopen librccl.so.1: %s); set PRE3_RCCL_LIB", dlerror()); return nullptr; }
    Rccl r; r.h = h;
#define PRE3_SYM(field, name) do { *(void **)(&r.field) = dlsym(h, name); if (!r.field) { set_error("RCCL: symbol %s missing", name); dlclose(h); return nullptr; } } while (0)
    PRE3_SYM(GetVersion, "ncclGetVersion"); PRE3_SYM(GetUniqueId, "ncclGetUniqueId"); PRE3_SYM(CommInitRank, "ncclCommInitRank");
    PRE3_SYM(CommDestroy, "ncclCommDestroy"); PRE3_SYM(CommAbort, "ncclCommAbort"); PRE3_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
    PRE3_SYM(AllReduce, "ncclAllReduce"); PRE3_SYM(AllGather, "ncclAllGather"); PRE3_SYM(GetErrorString, "ncclGetErrorString");
#undef PRE3_SYM
    Dl_info di;
    if (dladdr((void *)r.AllReduce, &di) && di.dli_fname) snprintf(r.where, sizeof(r.where), "%s", di.dli_fname);
    g_rccl = r;
    return &g_rccl;
}

#define PRE3_NCCL(R, expr) do { ncclResult_t nr_ = (expr); if (nr_ != ncclSuccess) { set_error("RCCL: %s failed: %s", #expr, (R)->GetErrorString(nr_)); return PRE3_E_COMM; } } while (0)

struct Comm {
    ncclComm_t comm = nullptr; int device = 0, rank = 0, world = 1; bool broken = false; int timeout_ms = 10000;
    // comm_give_up's abort runs on a thread of its own (it may wait for work that has not started).  The thread is kept: pre3_comm_destroy, pre3_destroy
    // and the shard's destroy join it -- within the deadline -- before the library's buffers or code can go away under it.
    std::thread aborter; std::atomic<bool> abort_done{ true };
};

int comm_all_reduce_i32(void *h, void *buf, size_t count, hipStream_t st)
{
    Comm *cm = (Comm *)h; const Rccl *R = rccl();
    PRE3_CHECK(cm && R && !cm->broken, PRE3_E_COMM, "communicator unusable");
    PRE3_NCCL(R, R->AllReduce(buf, buf, count, ncclInt32, ncclSum, cm->comm, st));
    return PRE3_OK;
}
int comm_all_gather_f64(void *h, const void *src, void *dst, size_t count, hipStream_t st)
{
    Comm *cm = (Comm *)h; const Rccl *R = rccl();
    PRE3_CHECK(cm && R && !cm->broken, PRE3_E_COMM, "communicator unusable");
    PRE3_NCCL(R, R->AllGather(src, dst, count, ncclFloat64, cm->comm, st));
    return PRE3_OK;
}
// 0: healthy; else the communicator has reported an asynchronous error (a peer died, a transport failed): it is aborted, so that the
// collective the stream is stuck in returns, and every later call on it fails
int comm_poll_error(void *h)
{
    Comm *cm = (Comm *)h; const Rccl *R = rccl();
    if (!cm || !R || cm->broken) return cm && cm->broken ? PRE3_E_COMM : PRE3_OK;
    ncclResult_t ae = ncclSuccess;
    if (R->CommGetAsyncError(cm->comm, &ae) != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress)) {
        set_error("RCCL: asynchronous communicator error: %s", R->GetErrorString(ae));
        cm->broken = true;
        (void)R->CommAbort(cm->comm); cm->comm = nullptr;
        return PRE3_E_COMM;
    }
    return PRE3_OK;
}
// the deadline of a host wait behind a collective (pre3_comm_set_timeout; default 10 s)
int comm_timeout_ms(void *h) { Comm *cm = (Comm *)h; return cm ? cm->timeout_ms : 0; }
// A host wait behind a collective has run out of time (a peer that stalls, or never entered): the communicator is aborted -- the collective the
// stream is stuck in then returns -- and marked broken; the caller returns PRE3_E_COMM.  The context stays usable with a fresh communicator.
int comm_give_up(void *h, const char *what)
{
    Comm *cm = (Comm *)h; const Rccl *R = rccl();
    set_error("%s did not complete within %d ms: the communicator is aborted (pre3_comm_set_timeout sets the deadline)", what, cm ? cm->timeout_ms : 0);
    if (cm && !cm->broken) {
        cm->broken = true;
        // ncclCommAbort makes a collective that is spinning on its peers return -- but it may itself wait for work that has not started yet (a
        // collective queued behind other kernels of the stream): it runs on a thread of its own, so that the caller has its PRE3_E_COMM at the
        // deadline whatever the abort has to wait for
        if (R && cm->comm) {
            ncclComm_t dead = cm->comm;
            const int dev = cm->device;
            auto abort_fn = R->CommAbort;
            if (cm->aborter.joinable()) cm->aborter.join();          // (an earlier abort of this handle: it has long returned)
            cm->abort_done.store(false);
            std::atomic<bool> *done = &cm->abort_done;
            cm->aborter = std::thread([dead, dev, abort_fn, done] { (void)hipSetDevice(dev); (void)abort_fn(dead); done->store(true); });
        }
        cm->comm = nullptr;
    }
    return PRE3_E_COMM;
}
// how long a join of the abort thread may take: ncclCommAbort tears the communicator down (proxy threads, IPC handles) once the collective has let go --
// local work, but measured at up to a second; the hang-proofing deadline (timeout_ms, possibly a few hundred ms) is too short a bound for it
int comm_abort_ms(void *h) { Comm *cm = (Comm *)h; return cm ? (cm->timeout_ms > 10000 ? cm->timeout_ms : 10000) : 0; }
bool comm_broken(void *h) { Comm *cm = (Comm *)h; return cm && cm->broken; }
// the abort thread, if one is running, has finished within `ms` (and is joined); false: it is still inside ncclCommAbort
bool comm_abort_wait(void *h, int ms)
{
    Comm *cm = (Comm *)h;
    if (!cm || !cm->aborter.joinable()) return true;
    const auto t0 = std::chrono::steady_clock::now();
    while (!cm->abort_done.load()) {
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > ms) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    cm->aborter.join();
    return true;
}
void comm_rank_world(void *h, int *rank, int *world) { Comm *cm = (Comm *)h; *rank = cm ? cm->rank : 0; *world = cm ? cm->world : 1; }
int comm_device(void *h) { return ((Comm *)h)->device; }

} // namespace pre3

using namespace pre3;

extern "C" {

int pre3_comm_unique_id(void *id_out)
{
    PRE3_CHECK(id_out != nullptr, PRE3_E_ARG, "pre3_comm_unique_id: null output");
    const Rccl *R = rccl();
    if (!R) return PRE3_E_COMM;
    static_assert(sizeof(ncclUniqueId) == PRE3_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    PRE3_NCCL(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return PRE3_OK;
}

int pre3_comm_create(pre3_comm **out, int device, const void *id, int rank, int world)
{
    PRE3_CHECK(out && id && world >= 1 && rank >= 0 && rank < world, PRE3_E_ARG, "pre3_comm_create: bad arguments (rank %d of %d)", rank, world);
    *out = nullptr;
    const Rccl *R = rccl();
    if (!R) return PRE3_E_COMM;
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return PRE3_E_NODEVICE; }
    ncclUniqueId uid; memcpy(&uid, id, sizeof(uid));
    Comm *cm = new Comm();
    cm->device = device; cm->rank = rank; cm->world = world;
    ncclResult_t nr = R->CommInitRank(&cm->comm, world, uid, rank);
    if (nr != ncclSuccess) { set_error("RCCL: ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, R->GetErrorString(nr)); delete cm; return PRE3_E_COMM; }
    *out = (pre3_comm *)cm;
    return PRE3_OK;
}

int pre3_comm_destroy(pre3_comm *h)
{
    Comm *cm = (Comm *)h;
    if (!cm) return PRE3_OK;
    const Rccl *R = rccl();
    if (R && cm->comm) { (void)hipSetDevice(cm->device); (void)(cm->broken ? R->CommAbort(cm->comm) : R->CommDestroy(cm->comm)); }
    // an abort started by a deadline may still be inside RCCL: it is joined here, within the same deadline.  If it does not come back the handle is
    // leaked on purpose (its thread keeps using it) and the call says so: the process must not unload librccl / libpre3 under that thread.
    if (!comm_abort_wait(cm, comm_abort_ms(cm))) {
        cm->aborter.detach();
        set_error("pre3_comm_destroy: the abort of the communicator has not returned within %d ms: handle leaked", comm_abort_ms(cm));
        return PRE3_E_COMM;
    }
    delete cm;
    return PRE3_OK;
}

int pre3_comm_set_timeout(pre3_comm *h, int milliseconds)
{
    Comm *cm = (Comm *)h;
    PRE3_CHECK(cm != nullptr && milliseconds >= 1, PRE3_E_ARG, "pre3_comm_set_timeout: null communicator or a non-positive deadline");
    cm->timeout_ms = milliseconds;
    return PRE3_OK;
}

int pre3_comm_info(pre3_comm *h, int *rank, int *world, int *rccl_version, char *lib_path, int lib_path_len)
{
    Comm *cm = (Comm *)h;
    PRE3_CHECK(cm != nullptr, PRE3_E_ARG, "pre3_comm_info: null communicator");
    const Rccl *R = rccl();
    if (!R) return PRE3_E_COMM;
    if (rank) *rank = cm->rank;
    if (world) *world = cm->world;
    if (rccl_version) { int v = 0; (void)R->GetVersion(&v); *rccl_version = v; }
    if (lib_path && lib_path_len > 0) snprintf(lib_path, (size_t)lib_path_len, "%s", R->where);
    return PRE3_OK;
}

} // extern "C"
