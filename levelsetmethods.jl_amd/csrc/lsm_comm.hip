// lsm_comm.hip — multi-GPU side of the C ABI (include/lsm.h, "multi-GPU" section): slab decomposition of the last
// dimension, exchange of LSM_GHOST full padded planes with the neighbouring ranks after every stage, Δt all-reduce
// (the loop bodies served are src/timestepping.jl:128-137,143-164,170-202 and src/levelsetterms.jl:22-28).
//
// Two transports behind the same entry points:
//   RCCL   one process per GPU.  librccl is opened at run time (dlopen): the library has no link-time dependency on it
//          and says so loudly when it is missing.  Planes travel as grouped ncclSend/ncclRecv on a stream of the
//          communicator's own, ordered against the handle's stream by events, so that the interior update of a stage
//          overlaps the exchange of its boundary planes.
//   LOCAL  every rank is a handle of ONE process (one host thread per rank, or one thread driving all ranks stage by
//          stage).  The rank that posts last enqueues the whole group's plane copies (peer copies between the handles'
//          buffers); a rank's stream then waits for the copies into its own ghost planes AND for its neighbours' copies
//          out of its boundary planes — whatever it launches next may overwrite them.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "lsm_handle.h"

namespace {

// ---- librccl, resolved at run time -------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
    });
    return r;
}

// ---- LOCAL transport: state shared by the handles of one in-process group --------------------------------------------
struct LocalGroup {
    int world = 0;
    std::vector<LsmHandle*> handles;
    std::mutex mu;
    std::condition_variable cv;
    unsigned long long posted_seq = 0;        // exchanges whose copies have been enqueued (by the last rank to post)
    std::vector<unsigned long long> seq;      // per rank: exchanges posted
    std::vector<void*> field;                 // per rank: field of the exchange being posted
    int nposted = 0;
    // Δt all-reduce
    std::vector<double> red;
    int nred = 0;
    unsigned long long red_gen = 0;
    double red_result = 0.0;
    int refs = 0;
};

}  // namespace

struct LsmComm {
    int transport;             // LSM_COMM_RCCL / LSM_COMM_LOCAL
    int rank, world;
    int up, dn;                // neighbour ranks (-1: none — a physical boundary)
    bool wrap_up, wrap_dn;     // the neighbour lies across the periodic wrap (period n-1: the duplicate end node is skipped)
    hipStream_t stream;        // the exchange runs here
    hipEvent_t ev_ready[2];    // handle's stream: the boundary planes of the posted field are final (parity of the exchange)
    hipEvent_t ev_done[2];     // exchange stream: this rank's ghost planes are filled (LOCAL: and its copies have read the neighbours' planes)
    unsigned long long seq;    // exchanges started by this rank
    bool pending;              // lsm_halo_start without its lsm_halo_wait
    bool overlap;              // stages update the interface planes first and overlap the exchange with the interior
    ncclComm_t nccl;
    double* d_dt;              // device scalar of the Δt all-reduce
    LocalGroup* grp;
};

namespace {

#define COMM_HIP(h, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return lsm_fail(h, LSM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define COMM_NCCL(h, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return lsm_fail(h, LSM_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

inline size_t esize(const LsmHandle* h) { return h->dtype == LSM_DTYPE_F32 ? sizeof(float) : sizeof(double); }

// geometry of the exchange: elements per padded plane, local plane range -> pointer
struct Planes {
    long long sl;      // elements per padded plane of the last dimension
    int nloc, G;
    size_t es;
    char* at(void* field, int k) const { return (char*)field + es * (size_t)((long long)(k + G) * sl); }   // local plane k, -G <= k < nloc+G
    size_t bytes() const { return es * (size_t)G * (size_t)sl; }
    size_t count() const { return (size_t)G * (size_t)sl; }
};
Planes planes_of(const LsmHandle* h) {
    const int N = h->grid.ndim;
    return Planes{h->lay.stride[N - 1], h->nloc[N - 1], LSM_GHOST, esize(h)};
}
// first plane this rank sends up / down (src/boundaryconditions.jl:107-119: nodes 1 and n coincide on a periodic
// dimension, so across the wrap the sender skips its duplicate end node)
inline int send_up_from(const LsmComm* c, const Planes& p) { return p.nloc - p.G - (c->wrap_up ? 1 : 0); }
inline int send_dn_from(const LsmComm* c) { return c->wrap_dn ? 1 : 0; }

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
    c->overlap = !(getenv("LSM_SLAB_OVERLAP") && getenv("LSM_SLAB_OVERLAP")[0] == '0');
    c->seq = 0; c->pending = false; c->nccl = nullptr; c->d_dt = nullptr; c->grp = nullptr; c->stream = nullptr;
    for (auto& e : c->ev_ready) e = nullptr;
    for (auto& e : c->ev_done) e = nullptr;
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_ready[i], hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_dt, sizeof(double));
    if (e != hipSuccess) {
        delete c;
        return lsm_fail(h, LSM_ERR_HIP, std::string("lsm_comm_attach: ") + hipGetErrorString(e));
    }
    *out = c;
    return LSM_OK;
}

void free_comm(LsmHandle* h, LsmComm* c) {
    (void)hipSetDevice(h->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    for (auto e : c->ev_ready) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_done) if (e) (void)hipEventDestroy(e);
    if (c->d_dt) (void)hipFree(c->d_dt);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->grp) {
        bool last;
        { std::lock_guard<std::mutex> g(c->grp->mu); last = --c->grp->refs == 0; }
        if (last) delete c->grp;
    }
    delete c;
}

// LOCAL: enqueue the plane copies of exchange `seq` for every rank of the group (called, under the group's mutex,
// by the rank that posted last).  Rank r PULLS its ghost planes from its neighbours' posted fields on its own exchange
// stream, behind the neighbours' and its own "boundary planes final" events.
int local_enqueue(LocalGroup* g, unsigned long long seq) {
    const int par = (int)(seq & 1);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (int r = 0; r < g->world; ++r) {
        LsmHandle* h = g->handles[r];
        LsmComm* c = h->comm;
        const Planes p = planes_of(h);
        COMM_HIP(h, hipSetDevice(h->device));
        COMM_HIP(h, hipStreamWaitEvent(c->stream, c->ev_ready[par], 0));
        if (c->dn >= 0) {
            LsmHandle* n = g->handles[c->dn];
            const Planes q = planes_of(n);
            COMM_HIP(h, hipStreamWaitEvent(c->stream, n->comm->ev_ready[par], 0));
            COMM_HIP(h, hipMemcpyAsync(p.at(g->field[r], -p.G), q.at(g->field[c->dn], send_up_from(n->comm, q)), p.bytes(), hipMemcpyDefault, c->stream));
        }
        if (c->up >= 0) {
            LsmHandle* n = g->handles[c->up];
            const Planes q = planes_of(n);
            COMM_HIP(h, hipStreamWaitEvent(c->stream, n->comm->ev_ready[par], 0));
            COMM_HIP(h, hipMemcpyAsync(p.at(g->field[r], p.nloc), q.at(g->field[c->up], send_dn_from(n->comm)), p.bytes(), hipMemcpyDefault, c->stream));
        }
        COMM_HIP(h, hipEventRecord(c->ev_done[par], c->stream));
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    return LSM_OK;
}

}  // namespace

bool lsm_comm_overlap(const LsmHandle* h) { return h->comm && h->comm->overlap; }

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
    g->handles.assign(handles, handles + world);
    g->seq.assign(world, 0);
    g->field.assign(world, nullptr);
    g->red.assign(world, 0.0);
    for (int r = 0; r < world; ++r) {
        LsmComm* c = nullptr;
        int rc = make_comm(handles[r], LSM_COMM_LOCAL, r, world, &c);
        if (rc == LSM_OK && (handles[r]->dtype != handles[0]->dtype || handles[r]->lay.stride[handles[r]->grid.ndim - 1] != handles[0]->lay.stride[handles[0]->grid.ndim - 1]))
            rc = lsm_fail(handles[r], LSM_ERR_INVALID, "lsm_comm_attach_local: the handles must share dtype and plane shape");
        if (rc) {
            if (c) { c->grp = nullptr; free_comm(handles[r], c); }
            for (int q = 0; q < r; ++q) { LsmComm* d = handles[q]->comm; handles[q]->comm = nullptr; d->grp = nullptr; free_comm(handles[q], d); }
            delete g;
            return rc;
        }
        c->grp = g;
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
    if (c->pending) return lsm_fail(h, LSM_ERR_INVALID, "lsm_halo_start: the previous exchange has not been waited for (lsm_halo_wait)");
    const unsigned long long seq = ++c->seq;
    const int par = (int)(seq & 1);
    c->pending = true;
    COMM_HIP(h, hipSetDevice(h->device));
    COMM_HIP(h, hipEventRecord(c->ev_ready[par], h->stream));
    const Planes p = planes_of(h);
    if (c->transport == LSM_COMM_RCCL) {
        if (c->up < 0 && c->dn < 0) { COMM_HIP(h, hipEventRecord(c->ev_done[par], h->stream)); return LSM_OK; }
        Rccl& r = rccl();
        const ncclDataType_t ty = h->dtype == LSM_DTYPE_F32 ? ncclFloat : ncclDouble;
        COMM_HIP(h, hipStreamWaitEvent(c->stream, c->ev_ready[par], 0));
        // op order [send up, recv dn, send dn, recv up]: the messages of one pair of ranks match in posting order, which
        // matters when up == dn (two ranks on a periodic ring)
        COMM_NCCL(h, r.GroupStart());
        if (c->up >= 0) COMM_NCCL(h, r.Send(p.at(field, send_up_from(c, p)), p.count(), ty, c->up, c->nccl, c->stream));
        if (c->dn >= 0) {
            COMM_NCCL(h, r.Recv(p.at(field, -p.G), p.count(), ty, c->dn, c->nccl, c->stream));
            COMM_NCCL(h, r.Send(p.at(field, send_dn_from(c)), p.count(), ty, c->dn, c->nccl, c->stream));
        }
        if (c->up >= 0) COMM_NCCL(h, r.Recv(p.at(field, p.nloc), p.count(), ty, c->up, c->nccl, c->stream));
        COMM_NCCL(h, r.GroupEnd());
        COMM_HIP(h, hipEventRecord(c->ev_done[par], c->stream));
        return LSM_OK;
    }
    // LOCAL: post; the last rank to post enqueues the copies of the whole group
    LocalGroup* g = c->grp;
    std::unique_lock<std::mutex> lk(g->mu);
    g->seq[c->rank] = seq;
    g->field[c->rank] = field;
    if (++g->nposted == g->world) {
        g->nposted = 0;
        const int rc = local_enqueue(g, seq);
        g->posted_seq = seq;
        lk.unlock();
        g->cv.notify_all();
        return rc;
    }
    return LSM_OK;
}

// The handle's stream waits for the planes received (and, LOCAL, for the neighbours' reads of this rank's planes).
// LOCAL blocks the calling thread until every rank of the group has posted the exchange.
int lsm_halo_wait(LsmHandle* h) {
    if (!h) return LSM_ERR_INVALID;
    LsmComm* c = h->comm;
    if (!c) return lsm_fail(h, LSM_ERR_INVALID, "lsm_halo_wait: no communicator attached");
    if (!c->pending) return LSM_OK;
    c->pending = false;
    const int par = (int)(c->seq & 1);
    COMM_HIP(h, hipSetDevice(h->device));
    if (c->transport == LSM_COMM_LOCAL) {
        LocalGroup* g = c->grp;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->posted_seq >= c->seq; });
        }
        if (c->dn >= 0) COMM_HIP(h, hipStreamWaitEvent(h->stream, g->handles[c->dn]->comm->ev_done[par], 0));
        if (c->up >= 0) COMM_HIP(h, hipStreamWaitEvent(h->stream, g->handles[c->up]->comm->ev_done[par], 0));
    }
    COMM_HIP(h, hipStreamWaitEvent(h->stream, c->ev_done[par], 0));
    return LSM_OK;
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
    if (c->transport == LSM_COMM_RCCL) {
        // NaN encoded as -1 (any valid or invalid Δt is >= 0 or NaN; -Inf cannot occur): MIN then makes it win
        const double enc = *dt != *dt ? -1.0 : *dt;
        COMM_HIP(h, hipSetDevice(h->device));
        // on the handle's stream, behind the stages of the previous step: one collective in flight per rank at a time
        COMM_HIP(h, hipMemcpyAsync(c->d_dt, &enc, sizeof(double), hipMemcpyHostToDevice, h->stream));
        COMM_NCCL(h, rccl().AllReduce(c->d_dt, c->d_dt, 1, ncclDouble, ncclMin, c->nccl, h->stream));
        double out = 0.0;
        COMM_HIP(h, hipMemcpyAsync(&out, c->d_dt, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        COMM_HIP(h, hipStreamSynchronize(h->stream));
        *dt = out < 0 ? __builtin_nan("") : out;
        return LSM_OK;
    }
    LocalGroup* g = c->grp;
    std::unique_lock<std::mutex> lk(g->mu);
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
    g->cv.wait(lk, [&] { return g->red_gen != gen; });
    *dt = g->red_result;
    return LSM_OK;
}

}  // extern "C"
