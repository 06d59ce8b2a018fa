// lsm_comm.hip — multi-GPU side of the C ABI (include/lsm.h, "multi-GPU" section): slab decomposition of the last
// dimension, exchange of LSM_GHOST full padded planes with the neighbouring ranks after every stage, Δt all-reduce
// (the loop bodies served are src/timestepping.jl:128-137,143-164,170-202 and src/levelsetterms.jl:22-28),
// and the overlap planes of a slab-decomposed NarrowBandMeshField (mask as whole planes of bytes, values sparse).
//
// Two transports behind the same entry points:
//   RCCL   one process per GPU.  librccl is opened at run time (dlopen): the library has no link-time dependency on it
//          and says so loudly when it is missing.  Messages travel as grouped ncclSend/ncclRecv on a stream of the
//          communicator's own, ordered against the handle's stream by events, so that the interior update of a stage
//          overlaps the exchange of its boundary planes.
//   LOCAL  every rank is a handle of ONE process (one host thread per rank, or one thread driving all ranks stage by
//          stage).  The rank that posts last enqueues the whole group's copies (peer copies between the ranks'
//          buffers); a rank's stream then waits for the copies into its own buffers AND for its neighbours' copies
//          out of its buffers — whatever it launches next may overwrite them.
//
// Every exchange is the same four messages: [send up, recv from down, send down, recv from up] — contiguous device
// buffers (plane ranges in place, or packed values).
//
// Failure model.  No blocking call here hangs for ever: every wait of the LOCAL transport and the host wait of
// lsm_allreduce_dt end with LSM_ERR_COMM when a peer has left the group (lsm_comm_detach / lsm_destroy), when any rank
// has called lsm_comm_abort, or after LSM_COMM_TIMEOUT_MS (default 60000) without progress — a call that is already
// complete still succeeds.  The LOCAL group owns the per-rank exchange streams and events (ref-counted, indexed by
// rank): they outlive a rank's handle, so a rank that has finished its last exchange may be destroyed while a slower
// neighbour is still ordering its stream behind that exchange's events, and no rank ever dereferences another rank's
// handle.  A communicator that has failed stays failed: every later call on it returns LSM_ERR_COMM.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "lsm_handle.h"

namespace {

// ---- librccl, resolved at run time -------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("librccl could not be opened: ") + dlerror(); return; }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.lib, n);
            if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    });
    return r;
}

// one exchange = four contiguous device buffers (byte counts; a NULL / 0 message is skipped on both sides)
struct Msg {
    void* send_up = nullptr; size_t n_send_up = 0;
    void* recv_dn = nullptr; size_t n_recv_dn = 0;
    void* send_dn = nullptr; size_t n_send_dn = 0;
    void* recv_up = nullptr; size_t n_recv_up = 0;
};

// ---- LOCAL transport: state shared by the handles of one in-process group --------------------------------------------
// what the group keeps of rank r: everything a PEER touches
struct LocalRank {
    bool present = false;                     // false once the rank has detached
    int device = 0;
    int up = -1, dn = -1;
    hipStream_t stream = nullptr;             // the rank's exchange stream
    hipEvent_t ev_ready[2] = {nullptr, nullptr};
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    Msg msg;                                  // the exchange being posted
};
struct LocalGroup {
    int world = 0;
    std::vector<LocalRank> ranks;
    std::mutex mu;
    std::condition_variable cv;
    unsigned long long posted_seq = 0;        // exchanges whose copies have been enqueued (by the last rank to post)
    int nposted = 0;
    // Δt all-reduce
    std::vector<double> red;
    int nred = 0;
    unsigned long long red_gen = 0;
    double red_result = 0.0;
    int refs = 0;
    bool failed = false;                      // a rank left, aborted or timed out: every unfinished wait returns LSM_ERR_COMM
    std::string why;
};

std::chrono::milliseconds comm_timeout(const LsmHandle* h) {
    const int ms = h ? h->tune.comm_timeout_ms : lsm::lsm_tuning_env().comm_timeout_ms;
    return std::chrono::milliseconds(ms > 0 ? ms : 60000);
}

}  // namespace

struct LsmComm {
    int transport;             // LSM_COMM_RCCL / LSM_COMM_LOCAL
    int rank, world;
    int up, dn;                // neighbour ranks (-1: none — a physical boundary)
    bool wrap_up, wrap_dn;     // the neighbour lies across the periodic wrap (period n-1: the duplicate end node is skipped)
    // LOCAL: the stream and the events are the group's (LocalGroup::ranks[rank]) and are destroyed with the group
    hipStream_t stream;        // the exchange runs here
    hipEvent_t ev_ready[2];    // handle's stream: the buffers of the posted exchange are final (parity of the exchange)
    hipEvent_t ev_done[2];     // exchange stream: this rank's receive buffers are filled (LOCAL: and its copies have read the neighbours' buffers)
    unsigned long long seq;    // exchanges started by this rank
    bool pending;              // an exchange was started and not yet waited for
    bool overlap;              // stages update the interface planes first and overlap the exchange with the interior
    std::atomic<bool> failed;  // a call failed half-way, a peer is gone, or lsm_comm_abort was called: LSM_ERR_COMM from now on
    ncclComm_t nccl;
    std::atomic<bool> nccl_aborted;   // ncclCommAbort has run (once): `nccl` is gone — never used again, never destroyed
    double* d_dt;              // device scalar of the Δt all-reduce
    double* h_dt;              // pinned host scalar
    LocalGroup* grp;
    // slab-decomposed NarrowBandMeshField (lsm_band_overlap_*): W planes of each neighbour held as planes of this slab
    int band_W;
    unsigned* d_idx[4];        // band nodes (byte offsets within the plane range) of [send up, recv dn, send dn, recv up]
    size_t idx_cap[4];
    unsigned n_idx[4];
    char* d_pack[4];           // packed values of those nodes
    size_t pack_cap[4];
    unsigned* d_nz_counts;     // scratch of the ordered compaction: per-chunk counts / offsets, and the four totals
    size_t nz_cap;
    unsigned* d_nz_total;      // 4 totals
    unsigned* h_nz_total;      // pinned
};

namespace {

#define COMM_HIP(h, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return lsm_fail(h, LSM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline size_t esize(const LsmHandle* h) { return h->dtype == LSM_DTYPE_F32 ? sizeof(float) : sizeof(double); }

// geometry of the plane exchange: elements per padded plane, local plane range -> pointer
struct Planes {
    long long sl;      // elements per padded plane of the last dimension
    int nloc, G;
    size_t es;
    char* at(void* field, int k) const { return (char*)field + es * (size_t)((long long)(k + G) * sl); }   // local plane k, -G <= k < nloc+G
    size_t bytes(int nplanes) const { return es * (size_t)nplanes * (size_t)sl; }
};
Planes planes_of(const LsmHandle* h, size_t es) {
    const int N = h->grid.ndim;
    return Planes{h->lay.stride[N - 1], h->nloc[N - 1], LSM_GHOST, es};
}
// first plane this rank sends up / down (src/boundaryconditions.jl:107-119: nodes 1 and n coincide on a periodic
// dimension, so across the wrap the sender skips its duplicate end node)
inline int send_dn_from(const LsmComm* c) { return c->wrap_dn ? 1 : 0; }

int comm_failed(LsmHandle* h, LsmComm* c, const char* what) {
    std::string why = "the communicator has failed";
    if (c->grp) { std::lock_guard<std::mutex> g(c->grp->mu); if (!c->grp->why.empty()) why = c->grp->why; }
    return lsm_fail(h, LSM_ERR_COMM, std::string(what) + ": " + why);
}
// mark the communicator (and its LOCAL group) failed and wake every waiter
void mark_failed(LsmComm* c, const std::string& why) {
    c->failed.store(true);
    if (c->grp) {
        { std::lock_guard<std::mutex> g(c->grp->mu); if (!c->grp->failed) { c->grp->failed = true; c->grp->why = why; } }
        c->grp->cv.notify_all();
    }
}

// RCCL: release this rank's queued send / receive / all-reduce kernels (a peer that never posts would keep them, and every
// host wait behind them, for ever).  Runs at most once per communicator; any thread.  After it `nccl` is never touched again.
void rccl_abort(LsmComm* c) {
    if (c->transport != LSM_COMM_RCCL || !c->nccl) return;
    if (c->nccl_aborted.exchange(true)) return;
    if (rccl().CommAbort) (void)rccl().CommAbort(c->nccl);
}

void free_local_group(LocalGroup* g) {
    for (auto& r : g->ranks) {
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
        for (auto e : r.ev_ready) if (e) (void)hipEventDestroy(e);
        for (auto e : r.ev_done) if (e) (void)hipEventDestroy(e);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    delete g;
}

int make_comm(LsmHandle* h, int transport, int rank, int world, LsmComm** out) {
    const int N = h->grid.ndim;
    if (h->comm) return lsm_fail(h, LSM_ERR_INVALID, "a communicator is already attached to this handle");
    if (world < 1 || rank < 0 || rank >= world) return lsm_fail(h, LSM_ERR_INVALID, "lsm_comm_attach: bad rank / world");
    const bool face_dn = h->bc[N - 1][0].kind == LSM_BC_NONE, face_up = h->bc[N - 1][1].kind == LSM_BC_NONE;
    if (world == 1 && (face_dn || face_up))
        return lsm_fail(h, LSM_ERR_INVALID, "lsm_comm_attach: a one-rank group cannot have slab interfaces (LSM_BC_NONE faces)");
    if (world > 1) {
        // interior faces must be slab interfaces; the outer faces of the end ranks are interfaces iff the dimension is periodic
        if ((rank > 0 && !face_dn) || (rank < world - 1 && !face_up))
            return lsm_fail(h, LSM_ERR_INVALID, "lsm_comm_attach: the faces towards neighbouring ranks must be LSM_BC_NONE");
        if (h->nloc[N - 1] < LSM_GHOST + 1)
            return lsm_fail(h, LSM_ERR_INVALID, "lsm_comm_attach: a slab needs at least LSM_GHOST + 1 planes");
    }
    LsmComm* c = new LsmComm();
    c->transport = transport; c->rank = rank; c->world = world;
    c->dn = face_dn ? (rank > 0 ? rank - 1 : world - 1) : -1;
    c->up = face_up ? (rank < world - 1 ? rank + 1 : 0) : -1;
    c->wrap_dn = face_dn && rank == 0;
    c->wrap_up = face_up && rank == world - 1;
    c->overlap = h->tune.slab_overlap != 0;
    c->seq = 0; c->pending = false; c->failed.store(false); c->nccl = nullptr; c->nccl_aborted.store(false); c->d_dt = nullptr; c->h_dt = nullptr; c->grp = nullptr;
    c->stream = nullptr;
    for (auto& e : c->ev_ready) e = nullptr;
    for (auto& e : c->ev_done) e = nullptr;
    c->band_W = 0;
    for (int k = 0; k < 4; ++k) { c->d_idx[k] = nullptr; c->idx_cap[k] = 0; c->n_idx[k] = 0; c->d_pack[k] = nullptr; c->pack_cap[k] = 0; }
    c->d_nz_counts = nullptr; c->nz_cap = 0; c->d_nz_total = nullptr; c->h_nz_total = nullptr;
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_dt, sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_dt, sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) {
        if (c->d_dt) (void)hipFree(c->d_dt);
        delete c;
        return lsm_fail(h, LSM_ERR_HIP, std::string("lsm_comm_attach: ") + hipGetErrorString(e));
    }
    *out = c;
    return LSM_OK;
}
// the exchange stream and its events: owned by the communicator (RCCL) or by the group's rank entry (LOCAL)
hipError_t make_stream_events(int device, hipStream_t* s, hipEvent_t (&ready)[2], hipEvent_t (&done)[2]) {
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ready[i], hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    return e;
}

void free_comm(LsmHandle* h, LsmComm* c) {
    (void)hipSetDevice(h->device);
    LocalGroup* g = c->grp;
    if (g) {
        // leave the group: peers that still wait for this rank get LSM_ERR_COMM; the stream and the events stay with the group
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->ranks[c->rank].present = false;
            if (!g->failed) { g->failed = true; g->why = "rank " + std::to_string(c->rank) + " has left the group"; }
        }
        g->cv.notify_all();
        // this rank's copies, and the neighbours' copies out of this rank's buffers, must have finished before the caller frees them
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (int nb : {c->dn, c->up})
            if (nb >= 0 && g->ranks[nb].stream) { (void)hipSetDevice(g->ranks[nb].device); (void)hipStreamSynchronize(g->ranks[nb].stream); }
        (void)hipSetDevice(h->device);
    } else {
        // a failed communicator may still have an unmatched send / receive / all-reduce kernel queued: abort first, or the
        // synchronisations below (and ncclCommDestroy) would wait for a peer that never comes
        if (c->failed.load()) rccl_abort(c);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(h->stream);            // the Δt all-reduce runs on the handle's stream
        if (c->nccl && !c->nccl_aborted.load()) (void)rccl().CommDestroy(c->nccl);
        for (auto e : c->ev_ready) if (e) (void)hipEventDestroy(e);
        for (auto e : c->ev_done) if (e) (void)hipEventDestroy(e);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    if (c->d_dt) (void)hipFree(c->d_dt);
    if (c->h_dt) (void)hipHostFree(c->h_dt);
    for (int k = 0; k < 4; ++k) { if (c->d_idx[k]) (void)hipFree(c->d_idx[k]); if (c->d_pack[k]) (void)hipFree(c->d_pack[k]); }
    if (c->d_nz_counts) (void)hipFree(c->d_nz_counts);
    if (c->d_nz_total) (void)hipFree(c->d_nz_total);
    if (c->h_nz_total) (void)hipHostFree(c->h_nz_total);
    if (g) {
        bool last;
        { std::lock_guard<std::mutex> lk(g->mu); last = --g->refs == 0; }
        if (last) free_local_group(g);
    }
    delete c;
}

// LOCAL: enqueue the copies of exchange `seq` for every rank of the group (called, under the group's mutex, by the rank
// that posted last).  Rank r PULLS its receive buffers from its neighbours' posted send buffers on its own exchange
// stream, behind the neighbours' and its own "buffers final" events.
int local_enqueue(LsmHandle* h, LocalGroup* g, unsigned long long seq) {
    const int par = (int)(seq & 1);
    struct DeviceGuard {      // the posting thread leaves on the device it came with, whatever happens below
        int cur = -1;
        DeviceGuard() { (void)hipGetDevice(&cur); }
        ~DeviceGuard() { if (cur >= 0) (void)hipSetDevice(cur); }
    } guard;
    for (int r = 0; r < g->world; ++r) {
        LocalRank& me = g->ranks[r];
        COMM_HIP(h, hipSetDevice(me.device));
        COMM_HIP(h, hipStreamWaitEvent(me.stream, me.ev_ready[par], 0));
        if (me.dn >= 0) {
            LocalRank& n = g->ranks[me.dn];
            if (n.msg.n_send_up != me.msg.n_recv_dn) return lsm_fail(h, LSM_ERR_COMM, "exchange: message sizes of neighbouring ranks differ");
            COMM_HIP(h, hipStreamWaitEvent(me.stream, n.ev_ready[par], 0));
            if (me.msg.n_recv_dn) COMM_HIP(h, hipMemcpyAsync(me.msg.recv_dn, n.msg.send_up, me.msg.n_recv_dn, hipMemcpyDefault, me.stream));
        }
        if (me.up >= 0) {
            LocalRank& n = g->ranks[me.up];
            if (n.msg.n_send_dn != me.msg.n_recv_up) return lsm_fail(h, LSM_ERR_COMM, "exchange: message sizes of neighbouring ranks differ");
            COMM_HIP(h, hipStreamWaitEvent(me.stream, n.ev_ready[par], 0));
            if (me.msg.n_recv_up) COMM_HIP(h, hipMemcpyAsync(me.msg.recv_up, n.msg.send_dn, me.msg.n_recv_up, hipMemcpyDefault, me.stream));
        }
        COMM_HIP(h, hipEventRecord(me.ev_done[par], me.stream));
    }
    return LSM_OK;
}

// wait (group mutex held) until done() — which wins even over a failed group — or the group fails / the timeout passes
template <class P>
int group_wait(LsmHandle* h, LsmComm* c, std::unique_lock<std::mutex>& lk, P done, const char* what) {
    LocalGroup* g = c->grp;
    const auto deadline = std::chrono::steady_clock::now() + comm_timeout(h);
    while (!done()) {
        if (g->failed) { c->failed.store(true); return lsm_fail(h, LSM_ERR_COMM, std::string(what) + ": " + g->why); }
        if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout && !done() && !g->failed) {
            g->failed = true;
            g->why = "rank " + std::to_string(c->rank) + " timed out waiting for the other ranks (LSM_COMM_TIMEOUT_MS)";
            g->cv.notify_all();
        }
    }
    return LSM_OK;
}

// start one exchange of four messages on the communicator
int exchange_start(LsmHandle* h, const Msg& m, ncclDataType_t ty, size_t tysize, const char* what) {
    LsmComm* c = h->comm;
    if (c->failed.load()) return comm_failed(h, c, what);
    if (c->pending) return lsm_fail(h, LSM_ERR_INVALID, std::string(what) + ": the previous exchange has not been waited for (lsm_halo_wait)");
    const unsigned long long seq = ++c->seq;
    const int par = (int)(seq & 1);
    COMM_HIP(h, hipSetDevice(h->device));
    COMM_HIP(h, hipEventRecord(c->ev_ready[par], h->stream));
    c->pending = true;
    if (c->transport == LSM_COMM_RCCL) {
        if (c->up < 0 && c->dn < 0) { COMM_HIP(h, hipEventRecord(c->ev_done[par], h->stream)); return LSM_OK; }
        Rccl& r = rccl();
        COMM_HIP(h, hipStreamWaitEvent(c->stream, c->ev_ready[par], 0));
        // op order [send up, recv dn, send dn, recv up]: the messages of one pair of ranks match in posting order, which
        // matters when up == dn (two ranks on a periodic ring).  A failure inside the group closes it, drops the exchange
        // and fails the communicator: nothing later waits on an event that was never recorded.
        ncclResult_t nr = r.GroupStart();
        if (nr == ncclSuccess) {
            if (c->up >= 0 && m.n_send_up && nr == ncclSuccess) nr = r.Send(m.send_up, m.n_send_up / tysize, ty, c->up, c->nccl, c->stream);
            if (c->dn >= 0 && m.n_recv_dn && nr == ncclSuccess) nr = r.Recv(m.recv_dn, m.n_recv_dn / tysize, ty, c->dn, c->nccl, c->stream);
            if (c->dn >= 0 && m.n_send_dn && nr == ncclSuccess) nr = r.Send(m.send_dn, m.n_send_dn / tysize, ty, c->dn, c->nccl, c->stream);
            if (c->up >= 0 && m.n_recv_up && nr == ncclSuccess) nr = r.Recv(m.recv_up, m.n_recv_up / tysize, ty, c->up, c->nccl, c->stream);
            const ncclResult_t ne = r.GroupEnd();
            if (nr == ncclSuccess) nr = ne;
        }
        if (nr != ncclSuccess) {
            c->pending = false;
            c->failed.store(true);
            return lsm_fail(h, LSM_ERR_COMM, std::string(what) + ": RCCL: " + r.GetErrorString(nr));
        }
        COMM_HIP(h, hipEventRecord(c->ev_done[par], c->stream));
        return LSM_OK;
    }
    // LOCAL: post; the last rank to post enqueues the copies of the whole group
    LocalGroup* g = c->grp;
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->failed) { c->pending = false; c->failed.store(true); return lsm_fail(h, LSM_ERR_COMM, std::string(what) + ": " + g->why); }
    g->ranks[c->rank].msg = m;
    if (++g->nposted == g->world) {
        g->nposted = 0;
        const int rc = local_enqueue(h, g, seq);
        if (rc) { g->failed = true; g->why = "enqueueing the copies of an exchange failed"; c->pending = false; c->failed.store(true); }
        else g->posted_seq = seq;
        lk.unlock();
        g->cv.notify_all();
        return rc;
    }
    return LSM_OK;
}

// The handle's stream waits for the messages received (and, LOCAL, for the neighbours' reads of this rank's buffers).
// LOCAL blocks the calling thread until every rank of the group has posted the exchange.
int exchange_wait(LsmHandle* h, const char* what) {
    LsmComm* c = h->comm;
    if (!c->pending) return c->failed.load() ? comm_failed(h, c, what) : LSM_OK;
    c->pending = false;
    const int par = (int)(c->seq & 1);
    COMM_HIP(h, hipSetDevice(h->device));
    if (c->transport == LSM_COMM_LOCAL) {
        LocalGroup* g = c->grp;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            const int rc = group_wait(h, c, lk, [&] { return g->posted_seq >= c->seq; }, what);
            if (rc) return rc;
        }
        // the neighbours' events belong to the group: valid even if the neighbour has been destroyed since
        if (c->dn >= 0) COMM_HIP(h, hipStreamWaitEvent(h->stream, g->ranks[c->dn].ev_done[par], 0));
        if (c->up >= 0) COMM_HIP(h, hipStreamWaitEvent(h->stream, g->ranks[c->up].ev_done[par], 0));
    } else if (c->failed.load()) {
        return comm_failed(h, c, what);
    }
    COMM_HIP(h, hipStreamWaitEvent(h->stream, c->ev_done[par], 0));
    return LSM_OK;
}

// host wait for the handle's stream that gives up when the communicator fails or the timeout passes
int stream_wait(LsmHandle* h, LsmComm* c, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    const auto deadline = t0 + comm_timeout(h);
    for (;;) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) return LSM_OK;
        if (e != hipErrorNotReady) return lsm_fail(h, LSM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        const auto now = std::chrono::steady_clock::now();
        if (c->failed.load() || now > deadline) {
            const bool timed_out = !c->failed.load();
            c->failed.store(true);
            rccl_abort(c);                            // releases the kernels waiting for the peer
            return lsm_fail(h, LSM_ERR_COMM, std::string(what) + (timed_out ? ": timed out waiting for the other ranks (LSM_COMM_TIMEOUT_MS)"
                                                                           : ": the communicator was aborted"));
        }
        if (now - t0 > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// ---- ordered compaction of a byte mask range: offsets (ascending) of the non-zero bytes -------------------------------
constexpr int NZ_CHUNK = 4096;      // bytes per workgroup (256 threads x 16)
typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ unsigned nz_bytes16(const unsigned char* p, size_t off, size_t n, unsigned& bits) {
    bits = 0;
    if (off + 16 <= n) {
        const unsigned long long a = *reinterpret_cast<const u64_unaligned*>(p + off), b = *reinterpret_cast<const u64_unaligned*>(p + off + 8);
        for (int k = 0; k < 8; ++k) {
            bits |= ((a >> (8 * k)) & 0xff) ? (1u << k) : 0u;
            bits |= ((b >> (8 * k)) & 0xff) ? (1u << (8 + k)) : 0u;
        }
    } else {
        for (int k = 0; k < 16; ++k)
            if (off + k < n && p[off + k]) bits |= 1u << k;
    }
    return (unsigned)__builtin_popcount(bits);
}
__device__ __forceinline__ unsigned block_sum256(unsigned v, unsigned* sh) {      // sum over a 256-thread block; sh: 4 words
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}
__global__ void __launch_bounds__(256) nz_count_kernel(const unsigned char* p, size_t n, unsigned* counts) {
    __shared__ unsigned sh[4];
    unsigned bits;
    const unsigned c = nz_bytes16(p, (size_t)blockIdx.x * NZ_CHUNK + threadIdx.x * 16, n, bits);
    const unsigned t = block_sum256(c, sh);
    if (threadIdx.x == 0) counts[blockIdx.x] = t;
}
// exclusive scan of the per-chunk counts in place (one workgroup), total -> *total
__global__ void __launch_bounds__(1024) nz_scan_kernel(unsigned* counts, unsigned nchunks, unsigned* total) {
    __shared__ unsigned sh[1024];
    const unsigned per = (nchunks + 1023u) / 1024u, b = threadIdx.x * per;
    unsigned s = 0;
    for (unsigned k = 0; k < per && b + k < nchunks; ++k) s += counts[b + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (unsigned off = 1; off < 1024; off <<= 1) {
        const unsigned v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = sh[threadIdx.x] - s;
    for (unsigned k = 0; k < per && b + k < nchunks; ++k) { const unsigned c = counts[b + k]; counts[b + k] = run; run += c; }
    if (threadIdx.x == 1023) *total = sh[1023];
}
__global__ void __launch_bounds__(256) nz_write_kernel(const unsigned char* p, size_t n, const unsigned* offsets, unsigned* out, size_t cap) {
    __shared__ unsigned wsum[4];
    unsigned bits;
    const size_t off = (size_t)blockIdx.x * NZ_CHUNK + threadIdx.x * 16;
    const unsigned c = nz_bytes16(p, off, n, bits);
    unsigned inc = c;                                   // inclusive scan over the wave, then over the 4 waves
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(inc, o, 64); if ((int)lane >= o) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = offsets[blockIdx.x];
    for (unsigned w = 0; w < wave; ++w) base += wsum[w];
    unsigned k = base + inc - c;
    for (int j = 0; j < 16; ++j)
        if ((bits >> j) & 1u) { if (k < cap) out[k] = (unsigned)(off + j); ++k; }
}
// values of the listed nodes of a plane range <-> a packed buffer (es = 4 or 8 bytes per value)
__global__ void __launch_bounds__(256) pack_kernel(const char* base, const unsigned* idx, unsigned n, int es, char* out) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (es == 8) reinterpret_cast<double*>(out)[i] = reinterpret_cast<const double*>(base)[idx[i]];
    else reinterpret_cast<float*>(out)[i] = reinterpret_cast<const float*>(base)[idx[i]];
}
__global__ void __launch_bounds__(256) unpack_kernel(char* base, const unsigned* idx, unsigned n, int es, const char* in) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (es == 8) reinterpret_cast<double*>(base)[idx[i]] = reinterpret_cast<const double*>(in)[i];
    else reinterpret_cast<float*>(base)[idx[i]] = reinterpret_cast<const float*>(in)[i];
}

// the four plane ranges of the band overlap: [send up, recv dn, send dn, recv up] as first local plane (W planes each);
// -1 = no neighbour on that side
void band_ranges(const LsmHandle* h, const LsmComm* c, int first[4]) {
    const int W = c->band_W, nloc = h->nloc[h->grid.ndim - 1];
    const int wlo = c->dn >= 0 ? W : 0, whi = c->up >= 0 ? W : 0;
    first[0] = c->up >= 0 ? nloc - whi - W : -1;      // my top owned planes -> up's lower overlap
    first[1] = c->dn >= 0 ? 0 : -1;                    // my lower overlap <- dn's top owned planes
    first[2] = c->dn >= 0 ? wlo : -1;                  // my bottom owned planes -> dn's upper overlap
    first[3] = c->up >= 0 ? nloc - whi : -1;           // my upper overlap <- up's bottom owned planes
}

}  // namespace

bool lsm_comm_overlap(const LsmHandle* h) { return h->comm && h->comm->overlap; }
// Host wait for the handle's stream.  Behind an RCCL exchange the stream may hold a send / receive kernel whose peer never
// posts: the wait then ends with LSM_ERR_COMM on lsm_comm_abort or after LSM_COMM_TIMEOUT_MS (and aborts the communicator, which
// releases the kernel) instead of hanging.  LOCAL copies are only enqueued once every rank has posted: a plain wait ends.
int lsm_host_sync(LsmHandle* h, const char* what) {
    LsmComm* c = h->comm;
    if (c && c->transport == LSM_COMM_RCCL && c->world > 1) return stream_wait(h, c, what);
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return lsm_fail(h, LSM_ERR_HIP, std::string(what) + ": hipStreamSynchronize: " + hipGetErrorString(e));
    return LSM_OK;
}
int lsm_comm_band_overlap(const LsmHandle* h) { return h->comm ? h->comm->band_W : 0; }

extern "C" {

int lsm_comm_unique_id(void* id_out) {
    if (!id_out) return LSM_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.err.empty()) return lsm_fail(nullptr, LSM_ERR_HIP, "lsm_comm_unique_id: " + r.err);
    static_assert(LSM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "LSM_COMM_ID_BYTES must match RCCL");
    ncclUniqueId id;
    ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) return lsm_fail(nullptr, LSM_ERR_HIP, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
    memcpy(id_out, &id, sizeof id);
    return LSM_OK;
}

int lsm_comm_attach_rccl(LsmHandle* h, const void* unique_id, int rank, int world) {
    if (!h || !unique_id) return LSM_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.err.empty()) return lsm_fail(h, LSM_ERR_HIP, "lsm_comm_attach_rccl: " + r.err);
    LsmComm* c = nullptr;
    int rc = make_comm(h, LSM_COMM_RCCL, rank, world, &c);
    if (rc) return rc;
    const hipError_t e = make_stream_events(h->device, &c->stream, c->ev_ready, c->ev_done);
    if (e != hipSuccess) {
        free_comm(h, c);
        return lsm_fail(h, LSM_ERR_HIP, std::string("lsm_comm_attach_rccl: ") + hipGetErrorString(e));
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t nr = r.CommInitRank(&c->nccl, world, id, rank);   // collective: every rank of the group is in this call
    if (nr != ncclSuccess) {
        c->nccl = nullptr;
        free_comm(h, c);
        return lsm_fail(h, LSM_ERR_HIP, std::string("ncclCommInitRank: ") + r.GetErrorString(nr));
    }
    h->comm = c;
    return LSM_OK;
}

int lsm_comm_attach_local(LsmHandle* const* handles, int world) {
    if (!handles || world < 1) return LSM_ERR_INVALID;
    for (int r = 0; r < world; ++r)
        if (!handles[r]) return LSM_ERR_INVALID;
    LocalGroup* g = new LocalGroup();
    g->world = world;
    g->ranks.resize(world);
    g->red.assign(world, 0.0);
    std::vector<LsmComm*> made;
    int rc = LSM_OK;
    for (int r = 0; r < world && rc == LSM_OK; ++r) {
        LsmComm* c = nullptr;
        rc = make_comm(handles[r], LSM_COMM_LOCAL, r, world, &c);
        if (rc) break;
        made.push_back(c);
        if (handles[r]->dtype != handles[0]->dtype || handles[r]->lay.stride[handles[r]->grid.ndim - 1] != handles[0]->lay.stride[handles[0]->grid.ndim - 1])
            rc = lsm_fail(handles[r], LSM_ERR_INVALID, "lsm_comm_attach_local: the handles must share dtype and plane shape");
        LocalRank& lr = g->ranks[r];
        lr.device = handles[r]->device; lr.up = c->up; lr.dn = c->dn;
        if (rc == LSM_OK) {
            const hipError_t e = make_stream_events(lr.device, &lr.stream, lr.ev_ready, lr.ev_done);
            if (e != hipSuccess) rc = lsm_fail(handles[r], LSM_ERR_HIP, std::string("lsm_comm_attach_local: ") + hipGetErrorString(e));
        }
    }
    if (rc) {
        for (size_t q = 0; q < made.size(); ++q) { made[q]->grp = nullptr; made[q]->stream = nullptr; free_comm(handles[q], made[q]); }
        free_local_group(g);          // releases whatever streams / events were created
        return rc;
    }
    for (int r = 0; r < world; ++r) {
        LsmComm* c = made[r];
        LocalRank& lr = g->ranks[r];
        lr.present = true;
        c->grp = g;
        c->stream = lr.stream;
        for (int i = 0; i < 2; ++i) { c->ev_ready[i] = lr.ev_ready[i]; c->ev_done[i] = lr.ev_done[i]; }
        handles[r]->comm = c;
    }
    g->refs = world;
    return LSM_OK;
}

int lsm_comm_detach(LsmHandle* h) {
    if (!h) return LSM_ERR_INVALID;
    if (!h->comm) return LSM_OK;
    LsmComm* c = h->comm;
    h->comm = nullptr;
    free_comm(h, c);
    return LSM_OK;
}

// Fail the communicator of this handle — and, LOCAL, its whole group: every rank blocked in (or later entering) an exchange
// wait or the Δt all-reduce returns LSM_ERR_COMM instead of waiting for a rank that will not come.  Callable from any
// thread — but not concurrently with lsm_comm_detach / lsm_destroy of the same handle (the communicator must outlive the call).
// RCCL: ncclCommAbort releases this rank's queued kernels here and now; the other processes notice through their own timeout.
int lsm_comm_abort(LsmHandle* h) {
    if (!h) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c) return LSM_OK;
    mark_failed(c, "aborted by rank " + std::to_string(c->rank) + " (lsm_comm_abort)");
    rccl_abort(c);
    return LSM_OK;
}

int lsm_comm_set_overlap(LsmHandle* h, int enable) {
    if (!h) return LSM_ERR_INVALID;
    if (!h->comm) return lsm_fail(h, LSM_ERR_INVALID, "lsm_comm_set_overlap: no communicator attached");
    h->comm->overlap = enable != 0;
    return LSM_OK;
}

int lsm_comm_info(const LsmHandle* h, int* rank, int* world, int* transport) {
    if (!h) return LSM_ERR_INVALID;
    if (rank) *rank = h->comm ? h->comm->rank : 0;
    if (world) *world = h->comm ? h->comm->world : 1;
    if (transport) *transport = h->comm ? h->comm->transport : LSM_COMM_NONE;
    return LSM_OK;
}

// Post the exchange of `field`'s LSM_GHOST planes next to each slab interface.  Everything queued on the handle's stream
// so far (the stage that produced those planes, earlier readers of the ghost planes) is ordered before it.
int lsm_halo_start(LsmHandle* h, void* field) {
    if (!h || !field) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c) return lsm_fail(h, LSM_ERR_INVALID, "lsm_halo_start: no communicator attached (lsm_comm_attach_rccl / _local)");
    const Planes p = planes_of(h, esize(h));
    // Only the planes a stencil of the step in progress reads travel (g = 3 for WENO5, 2 for the ENO2 terms, 1 for
    // upwind / curvature — src/derivatives.jl:89-121, src/levelsetterms.jl:156-170, src/levelsetops.jl:234-244): the d planes
    // nearest the interface, into the d ghost planes nearest the receiver's interior.  d = LsmHandle::ghost_depth: set from the
    // term list by lsm_advance_* on every rank alike, LSM_GHOST outside a step (lsm_halo_exchange called by the host).
    const int d = h->ghost_depth >= 1 && h->ghost_depth <= p.G ? h->ghost_depth : p.G;
    if (d == p.G) h->slab_depth_valid = p.G;        // a full-depth exchange (any call from outside a step) makes every layer valid again
    Msg m;
    const size_t nb = p.bytes(d);
    if (c->up >= 0) { m.send_up = p.at(field, p.nloc - d - (c->wrap_up ? 1 : 0)); m.n_send_up = nb; m.recv_up = p.at(field, p.nloc); m.n_recv_up = nb; }
    if (c->dn >= 0) { m.recv_dn = p.at(field, -d); m.n_recv_dn = nb; m.send_dn = p.at(field, send_dn_from(c)); m.n_send_dn = nb; }
    return exchange_start(h, m, h->dtype == LSM_DTYPE_F32 ? ncclFloat : ncclDouble, esize(h), "lsm_halo_start");
}

int lsm_halo_wait(LsmHandle* h) {
    if (!h) return LSM_ERR_INVALID;
    if (!h->comm) return lsm_fail(h, LSM_ERR_INVALID, "lsm_halo_wait: no communicator attached");
    return exchange_wait(h, "lsm_halo_wait");
}

int lsm_halo_exchange(LsmHandle* h, void* field) {
    int rc = lsm_halo_start(h, field);
    return rc ? rc : lsm_halo_wait(h);
}

// Δt = min over ranks, NaN wins (the reference's `min` propagates NaN, src/levelsetterms.jl:22-28).  Synchronous.
// LOCAL blocks until every rank of the group has called it (one host thread per rank).
int lsm_allreduce_dt(LsmHandle* h, double* dt) {
    if (!h || !dt) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c) return lsm_fail(h, LSM_ERR_INVALID, "lsm_allreduce_dt: no communicator attached");
    if (c->world == 1) return LSM_OK;
    if (c->failed.load()) return comm_failed(h, c, "lsm_allreduce_dt");
    if (c->transport == LSM_COMM_RCCL) {
        // NaN encoded as -1 (any valid or invalid Δt is >= 0 or NaN; -Inf cannot occur): MIN then makes it win
        *c->h_dt = *dt != *dt ? -1.0 : *dt;
        COMM_HIP(h, hipSetDevice(h->device));
        // on the handle's stream, behind the stages of the previous step: one collective in flight per rank at a time
        COMM_HIP(h, hipMemcpyAsync(c->d_dt, c->h_dt, sizeof(double), hipMemcpyHostToDevice, h->stream));
        const ncclResult_t nr = rccl().AllReduce(c->d_dt, c->d_dt, 1, ncclDouble, ncclMin, c->nccl, h->stream);
        if (nr != ncclSuccess) {
            c->failed.store(true);
            return lsm_fail(h, LSM_ERR_COMM, std::string("lsm_allreduce_dt: RCCL: ") + rccl().GetErrorString(nr));
        }
        COMM_HIP(h, hipMemcpyAsync(c->h_dt, c->d_dt, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        const int rc = stream_wait(h, c, "lsm_allreduce_dt");
        if (rc) return rc;
        *dt = *c->h_dt < 0 ? __builtin_nan("") : *c->h_dt;
        return LSM_OK;
    }
    LocalGroup* g = c->grp;
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->failed) { c->failed.store(true); return lsm_fail(h, LSM_ERR_COMM, "lsm_allreduce_dt: " + g->why); }
    g->red[c->rank] = *dt;
    const unsigned long long gen = g->red_gen;
    if (++g->nred == g->world) {
        double m = g->red[0];
        for (int r = 1; r < g->world; ++r) {
            const double x = g->red[r];
            m = (m != m || x != x) ? __builtin_nan("") : (x < m ? x : m);
        }
        g->red_result = m;
        g->nred = 0;
        ++g->red_gen;
        lk.unlock();
        g->cv.notify_all();
        *dt = m;
        return LSM_OK;
    }
    const int rc = group_wait(h, c, lk, [&] { return g->red_gen != gen; }, "lsm_allreduce_dt");
    if (rc) return rc;
    *dt = g->red_result;
    return LSM_OK;
}

// ---- slab-decomposed NarrowBandMeshField --------------------------------------------------------------------------------
// A band's operations reach farther across a slab interface than a stencil (nearest band node within 6, its slope
// neighbour, the stencil's 3 — src/meshfield.jl:481-530), so the slab of each rank is created `overlap` planes larger towards
// each neighbour; those planes are ordinary planes of the slab, computed redundantly, and refreshed from their owners:
//   lsm_band_overlap_mask    whole planes of mask bytes; then both sides enumerate the band nodes of the exchanged plane
//                            ranges in index order from the (now identical) masks — no index ever travels.  Synchronous
//                            (the list lengths size the value messages).
//   lsm_band_overlap_values  the values of exactly those nodes, packed (≈0.4 MB instead of 48 MB per direction at 768³).
int lsm_band_overlap_config(LsmHandle* h, int64_t overlap) {
    if (!h) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c) return lsm_fail(h, LSM_ERR_INVALID, "lsm_band_overlap_config: no communicator attached");
    const int N = h->grid.ndim;
    const int64_t need = overlap * ((c->dn >= 0 ? 1 : 0) + (c->up >= 0 ? 1 : 0)) + overlap;
    if (overlap < 1 || h->nloc[N - 1] < need)
        return lsm_fail(h, LSM_ERR_INVALID, "lsm_band_overlap_config: the slab must hold `overlap` planes of each neighbour and at least as many of its own");
    if (c->wrap_up || c->wrap_dn) return lsm_fail(h, LSM_ERR_INVALID, "lsm_band_overlap_config: PeriodicBC is not supported on a NarrowBandMeshField");
    c->band_W = (int)overlap;
    for (int k = 0; k < 4; ++k) c->n_idx[k] = 0;
    if (!c->d_nz_total) {
        COMM_HIP(h, hipSetDevice(h->device));
        COMM_HIP(h, hipMalloc((void**)&c->d_nz_total, 4 * sizeof(unsigned)));
        COMM_HIP(h, hipHostMalloc((void**)&c->h_nz_total, 4 * sizeof(unsigned), hipHostMallocDefault));
    }
    return LSM_OK;
}

int lsm_band_overlap_mask(LsmHandle* h, void* mask) {
    if (!h || !mask) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c || c->band_W < 1) return lsm_fail(h, LSM_ERR_INVALID, "lsm_band_overlap_mask: call lsm_band_overlap_config first");
    const Planes p = planes_of(h, 1);
    const int W = c->band_W;
    int first[4];
    band_ranges(h, c, first);
    const size_t nb = p.bytes(W);
    Msg m;
    if (first[0] >= 0) { m.send_up = p.at(mask, first[0]); m.n_send_up = nb; m.recv_up = p.at(mask, first[3]); m.n_recv_up = nb; }
    if (first[1] >= 0) { m.recv_dn = p.at(mask, first[1]); m.n_recv_dn = nb; m.send_dn = p.at(mask, first[2]); m.n_send_dn = nb; }
    int rc = exchange_start(h, m, ncclChar, 1, "lsm_band_overlap_mask");
    if (rc == LSM_OK) rc = exchange_wait(h, "lsm_band_overlap_mask");
    if (rc) return rc;
    // band nodes of the four plane ranges, in index order
    const unsigned nchunks = (unsigned)((nb + NZ_CHUNK - 1) / NZ_CHUNK);
    COMM_HIP(h, hipSetDevice(h->device));
    if (c->nz_cap < (size_t)4 * nchunks) {
        if (c->d_nz_counts) (void)hipFree(c->d_nz_counts);
        c->d_nz_counts = nullptr; c->nz_cap = 0;
        COMM_HIP(h, hipMalloc((void**)&c->d_nz_counts, (size_t)4 * nchunks * sizeof(unsigned)));
        c->nz_cap = (size_t)4 * nchunks;
    }
    COMM_HIP(h, hipMemsetAsync(c->d_nz_total, 0, 4 * sizeof(unsigned), h->stream));
    for (int k = 0; k < 4; ++k) {
        if (first[k] < 0) continue;
        const unsigned char* base = (const unsigned char*)p.at(mask, first[k]);
        hipLaunchKernelGGL(nz_count_kernel, dim3(nchunks), dim3(256), 0, h->stream, base, nb, c->d_nz_counts + (size_t)k * nchunks);
        hipLaunchKernelGGL(nz_scan_kernel, dim3(1), dim3(1024), 0, h->stream, c->d_nz_counts + (size_t)k * nchunks, nchunks, c->d_nz_total + k);
    }
    COMM_HIP(h, hipMemcpyAsync(c->h_nz_total, c->d_nz_total, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    rc = stream_wait(h, c, "lsm_band_overlap_mask");      // behind the mask exchange: a peer that never posts must not hang the host
    if (rc) return rc;
    const size_t es = esize(h);
    for (int k = 0; k < 4; ++k) {
        c->n_idx[k] = first[k] >= 0 ? c->h_nz_total[k] : 0u;
        if (!c->n_idx[k]) continue;
        if (c->idx_cap[k] < c->n_idx[k]) {
            if (c->d_idx[k]) (void)hipFree(c->d_idx[k]);
            if (c->d_pack[k]) (void)hipFree(c->d_pack[k]);
            c->d_idx[k] = nullptr; c->d_pack[k] = nullptr; c->idx_cap[k] = 0;
            const size_t cap = (size_t)c->n_idx[k] + c->n_idx[k] / 2 + 1024;
            COMM_HIP(h, hipMalloc((void**)&c->d_idx[k], cap * sizeof(unsigned)));
            COMM_HIP(h, hipMalloc((void**)&c->d_pack[k], cap * sizeof(double)));
            c->idx_cap[k] = cap;
        }
        const unsigned char* base = (const unsigned char*)p.at(mask, first[k]);
        hipLaunchKernelGGL(nz_write_kernel, dim3(nchunks), dim3(256), 0, h->stream, base, nb, c->d_nz_counts + (size_t)k * nchunks, c->d_idx[k], c->idx_cap[k]);
    }
    (void)es;
    COMM_HIP(h, hipGetLastError());
    return LSM_OK;
}

int lsm_band_overlap_values(LsmHandle* h, void* field) {
    if (!h || !field) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c || c->band_W < 1) return lsm_fail(h, LSM_ERR_INVALID, "lsm_band_overlap_values: call lsm_band_overlap_config first");
    const size_t es = esize(h);
    const Planes p = planes_of(h, es);
    int first[4];
    band_ranges(h, c, first);
    COMM_HIP(h, hipSetDevice(h->device));
    for (int k : {0, 2})      // pack what this rank sends
        if (first[k] >= 0 && c->n_idx[k])
            hipLaunchKernelGGL(pack_kernel, dim3((c->n_idx[k] + 255u) / 256u), dim3(256), 0, h->stream, (const char*)p.at(field, first[k]), c->d_idx[k],
                               c->n_idx[k], (int)es, c->d_pack[k]);
    Msg m;
    m.send_up = c->d_pack[0]; m.n_send_up = es * c->n_idx[0];
    m.recv_dn = c->d_pack[1]; m.n_recv_dn = es * c->n_idx[1];
    m.send_dn = c->d_pack[2]; m.n_send_dn = es * c->n_idx[2];
    m.recv_up = c->d_pack[3]; m.n_recv_up = es * c->n_idx[3];
    int rc = exchange_start(h, m, ncclChar, 1, "lsm_band_overlap_values");
    if (rc == LSM_OK) rc = exchange_wait(h, "lsm_band_overlap_values");
    if (rc) return rc;
    for (int k : {1, 3})      // scatter what it received
        if (first[k] >= 0 && c->n_idx[k])
            hipLaunchKernelGGL(unpack_kernel, dim3((c->n_idx[k] + 255u) / 256u), dim3(256), 0, h->stream, (char*)p.at(field, first[k]), c->d_idx[k],
                               c->n_idx[k], (int)es, c->d_pack[k]);
    COMM_HIP(h, hipGetLastError());
    return LSM_OK;
}

}  // extern "C"
