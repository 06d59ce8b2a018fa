// lsm_api.hip — host side of the C ABI declared in include/lsm.h.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "lsm_internal.h"

namespace lsm {
int launch_stage_fast_1d(const Combo&, const StageArgs&, hipStream_t);
int launch_stage_fast_2d(const Combo&, const StageArgs&, hipStream_t);
int launch_stage_fast_3d(const Combo&, const StageArgs&, hipStream_t);
int launch_stage_strict_1d(const Combo&, const StageArgs&, hipStream_t);
int launch_stage_strict_2d(const Combo&, const StageArgs&, hipStream_t);
int launch_stage_strict_3d(const Combo&, const StageArgs&, hipStream_t);

// must list exactly the combinations of LSM_FOR_EACH_COMBO (stage_kernel.h)
bool combo_available(const Combo& c) {
    static const int tab[][4] = {{1, 0, 0, 0}, {2, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {0, 0, 0, 2},
                                 {2, 0, 0, 2}, {2, 0, 0, 1}, {0, 1, 1, 0}, {2, 0, 1, 0}, {2, 1, 0, 0}};
    for (auto& t : tab)
        if (t[0] == c.adv && t[1] == c.nm && t[2] == c.curv && t[3] == c.eik) return true;
    return false;
}
}  // namespace lsm

using namespace lsm;

#include "lsm_handle.h"

struct LsmHandle;
static BandArgs band_args(const LsmHandle* h, int mc, const unsigned char* work);
static bool have_lists(const LsmHandle* h, const void* tiles, int mc);
static const int MAXB = 4096;
static std::string g_create_err;

static int fail(LsmHandle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}
int lsm_fail(LsmHandle* h, int code, const std::string& msg) { return fail(h, code, msg); }
#define LSM_HIP(h, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(h, LSM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
// host wait for the handle's stream (lsm_host_sync: cannot be hung by a silent RCCL peer)
#define LSM_SYNC(h) do { int rs_ = lsm_host_sync(h, __func__); if (rs_) return rs_; } while (0)

// storage of a field value: 8 bytes, or 4 for LSM_DTYPE_F32 handles (coefficient fields, frozen signs and every
// reduction stay fp64)
static inline int is_f32(const LsmHandle* h) { return h->dtype == LSM_DTYPE_F32 ? 1 : 0; }
static inline size_t esize(const LsmHandle* h) { return h->dtype == LSM_DTYPE_F32 ? sizeof(float) : sizeof(double); }

// _lagrange_extrap_weight — src/boundaryconditions.jl:90-97
static double lagrange_w(int j, int k, int P) {
    double w = 1.0;
    for (int m = 0; m <= P; ++m) {
        if (m == j) continue;
        w *= (double)(-k - m) / (double)(j - m);
    }
    return w;
}

namespace lsm {
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
// The environment, read once per process (the only getenv calls of the library).  A switch that is merely SET counts as 1 for the
// flags that used to be tested for presence (LSM_STAGE_GENERIC, LSM_BAND_BYTES, LSM_BAND_NO_LISTS, LSM_GHOST_FULL_DEPTH).
const LsmTuning& lsm_tuning_env() {
    static const LsmTuning t = [] {
        LsmTuning u;
        auto flag = [](const char* n) { const char* v = getenv(n); return v ? (*v && atoi(v) == 0 && v[0] == '0' ? 0 : 1) : 0; };
        u.stage_tail = env_int("LSM_STAGE_TAIL", 16);
        u.stage_tail_dyn = env_int("LSM_STAGE_TAIL_DYN", 25);
        u.stage_mc = env_int("LSM_STAGE_MC", 0);
        u.stage_mc2 = env_int("LSM_STAGE_MC2", 0);
        u.pairs = env_int("LSM_PAIRS", 1);
        u.stage_generic = flag("LSM_STAGE_GENERIC");
        u.xredirect = env_int("LSM_XREDIRECT", 1);
        u.mredirect = env_int("LSM_MREDIRECT", 1);
        u.ghost_full_depth = flag("LSM_GHOST_FULL_DEPTH");
        u.band_bricks = env_int("LSM_BAND_BRICKS", 1);
        u.band_bits = env_int("LSM_BAND_BITS", 1);
        u.band_cfl_prefetch = env_int("LSM_BAND_CFL_PREFETCH", 1);
        u.band_bytes = flag("LSM_BAND_BYTES");
        u.band_no_lists = flag("LSM_BAND_NO_LISTS");
        u.status_spin = env_int("LSM_STATUS_SPIN", 1);
        u.slab_overlap = env_int("LSM_SLAB_OVERLAP", 1);
        u.comm_timeout_ms = env_int("LSM_COMM_TIMEOUT_MS", 60000);
        if (u.comm_timeout_ms <= 0) u.comm_timeout_ms = 60000;
        u.layout_align = env_int("LSM_LAYOUT_ALIGN", 1);
        return u;
    }();
    return t;
}
int* lsm_tuning_field(LsmTuning& t, const char* name) {
    if (!name) return nullptr;
    const struct { const char* n; int* p; } tab[] = {
        {"LSM_STAGE_TAIL", &t.stage_tail}, {"LSM_STAGE_TAIL_DYN", &t.stage_tail_dyn}, {"LSM_STAGE_MC", &t.stage_mc}, {"LSM_STAGE_MC2", &t.stage_mc2},
        {"LSM_PAIRS", &t.pairs}, {"LSM_STAGE_GENERIC", &t.stage_generic}, {"LSM_XREDIRECT", &t.xredirect}, {"LSM_MREDIRECT", &t.mredirect},
        {"LSM_GHOST_FULL_DEPTH", &t.ghost_full_depth}, {"LSM_BAND_BRICKS", &t.band_bricks}, {"LSM_BAND_BITS", &t.band_bits},
        {"LSM_BAND_CFL_PREFETCH", &t.band_cfl_prefetch}, {"LSM_BAND_BYTES", &t.band_bytes}, {"LSM_BAND_NO_LISTS", &t.band_no_lists},
        {"LSM_STATUS_SPIN", &t.status_spin}, {"LSM_SLAB_OVERLAP", &t.slab_overlap}, {"LSM_COMM_TIMEOUT_MS", &t.comm_timeout_ms},
        {"LSM_LAYOUT_ALIGN", &t.layout_align},
    };
    for (const auto& e : tab)
        if (strcmp(e.n, name) == 0) return e.p;
    return nullptr;
}
}  // namespace lsm

extern "C" {

int lsm_set_tuning(LsmHandle* h, const char* name, int value) {
    if (!h) return LSM_ERR_INVALID;
    int* p = lsm_tuning_field(h->tune, name);
    if (!p) return fail(h, LSM_ERR_INVALID, std::string("lsm_set_tuning: no such switch: ") + (name ? name : "(null)"));
    if (p == &h->tune.layout_align) return fail(h, LSM_ERR_INVALID, "lsm_set_tuning: LSM_LAYOUT_ALIGN is fixed when the handle is created (environment only)");
    *p = value;
    h->no_lists = h->tune.band_no_lists != 0;
    h->band_bytes = h->tune.band_bytes != 0;
    if (h->tune.band_no_lists) { h->lists_tiles = nullptr; h->lists_host_valid = false; }
    return LSM_OK;
}
int lsm_get_tuning(const LsmHandle* h, const char* name, int* value) {
    if (!h || !value) return LSM_ERR_INVALID;
    LsmTuning t = h->tune;
    const int* p = lsm_tuning_field(t, name);
    if (!p) return LSM_ERR_INVALID;
    *value = *p;
    return LSM_OK;
}

const char* lsm_version(void) { return "hiplsm 0.1 (gfx950)"; }

const char* lsm_last_error(const LsmHandle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int lsm_create(const LsmGrid* grid, const LsmBc bc[LSM_MAX_DIM][2], const LsmSlab* slab, int dtype, int mode, int device,
               LsmHandle** out) {
    if (!grid || !bc || !out) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: null argument");
    if (grid->ndim < 1 || grid->ndim > 3) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: ndim must be 1, 2 or 3");
    if (dtype != LSM_DTYPE_F64 && dtype != LSM_DTYPE_F32) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: dtype must be LSM_DTYPE_F64 or LSM_DTYPE_F32");
    if (mode != LSM_MODE_FAST && mode != LSM_MODE_STRICT) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: bad mode");
    const int N = grid->ndim;
    for (int d = 0; d < 3; ++d) {
        if (d >= N && grid->n[d] != 1) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: unused dims must have n = 1");
        if (d < N && grid->n[d] < 4) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: need at least 4 nodes per dimension");
        if (d < N && grid->n[d] > 0x7fffffff) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: dimension too large");
    }
    for (int d = 0; d < N; ++d) {
        const bool pl = bc[d][0].kind == LSM_BC_PERIODIC, pr = bc[d][1].kind == LSM_BC_PERIODIC;
        const bool slabface = bc[d][0].kind == LSM_BC_NONE || bc[d][1].kind == LSM_BC_NONE;
        if (pl != pr && !slabface)   // src/boundaryconditions.jl:184-186
            return fail(nullptr, LSM_ERR_INVALID, "periodic boundary conditions cannot be mixed with others in a dimension");
        for (int s = 0; s < 2; ++s) {
            const LsmBc& b = bc[d][s];
            if (b.kind < 0 || b.kind > LSM_BC_NONE) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: bad bc kind");
            if (b.kind == LSM_BC_EXTRAPOLATION && (b.degree < 0 || b.degree > 7))   // src/boundaryconditions.jl:42
                return fail(nullptr, LSM_ERR_INVALID, "extrapolation order P must be in 0..7");
            if (b.kind == LSM_BC_NONE && d != N - 1)
                return fail(nullptr, LSM_ERR_INVALID, "LSM_BC_NONE (slab interface) is only valid on the last dimension");
        }
    }
    int dev_count = 0;
    const hipError_t dev_err = hipGetDeviceCount(&dev_count);
    if (dev_err != hipSuccess || dev_count <= 0) {
        static char msg[256];
        snprintf(msg, sizeof msg, "lsm_create: no HIP device available (this library has no CPU path) [hipGetDeviceCount: %s, count %d]",
                 hipGetErrorString(dev_err), dev_count);
        return fail(nullptr, LSM_ERR_NO_DEVICE, msg);
    }
    if (device < 0 || device >= dev_count) return fail(nullptr, LSM_ERR_INVALID, "lsm_create: bad device index");

    LsmHandle* h = new LsmHandle();
    h->grid = *grid;
    memcpy(h->bc, bc, sizeof(h->bc));
    h->dtype = dtype; h->mode = mode; h->device = device;
    h->prof = false; h->ev_used = 0; h->prof_every = 1; h->prof_seen = 0;
    h->comm = nullptr;
    h->ghost_depth = LSM_GHOST;
    h->slab_depth_valid = LSM_GHOST;     // a slab's ghosts are the host's to make valid before its first step (include/lsm.h)
    h->status_ticket = 0;
    h->mredirect = false;
    h->reinit_ws = nullptr;
    h->d_pf_flag = nullptr;
    memset(&h->band_cfl, 0, sizeof(h->band_cfl));
    h->cfl_prefetched = false;
    h->d_stamp = nullptr;
    h->d_tail_ctr = nullptr; h->tail_ticket = 0;
    h->xredirect = false;
    h->yredirect = false;
    h->cfl_cache_on = true;
    h->d_cand_count = nullptr;
    h->d_ring = nullptr; h->nring = 0; h->d_miss = nullptr; h->d_count = nullptr; h->d_work = nullptr; h->d_tiles_old = nullptr; h->work_cap = 0; h->halo_n_key = nullptr; h->halo_n = 0;
    h->band_mask = nullptr; h->band_tiles = nullptr; h->band_mc = 0; h->band_list = nullptr; h->band_nlist = 0;
    h->d_act_list = nullptr; h->d_work_list = nullptr; h->d_lcounts = nullptr; h->lists_tiles = nullptr; h->lists_mc = 0;
    h->lists_host_valid = false; h->nact = h->nwork = h->nface = 0;
    h->tune = lsm_tuning_env();
    h->no_lists = h->tune.band_no_lists != 0;
    h->band_bytes = h->tune.band_bytes != 0;
    h->slab.lo = 0; h->slab.n = grid->n[N - 1];
    if (slab) h->slab = *slab;
    if (h->slab.lo < 0 || h->slab.n < LSM_GHOST || h->slab.lo + h->slab.n > grid->n[N - 1]) {
        delete h;
        return fail(nullptr, LSM_ERR_INVALID, "lsm_create: bad slab (needs >= 3 planes inside the grid)");
    }
    long long s = 1;
    h->lay.origin = 0;
    const bool align_rows = lsm_tuning_env().layout_align != 0;      // LSM_LAYOUT_ALIGN=0: the compact layout (A/B)
    for (int d = 0; d < 3; ++d) {
        h->nloc[d] = (int)grid->n[d];
        h->goff[d] = 0;
        h->gn[d] = (int)grid->n[d];
        if (d == N - 1) { h->nloc[d] = (int)h->slab.n; h->goff[d] = (int)h->slab.lo; }
        h->lay.n[d] = h->nloc[d];
        h->lay.g[d] = d < N ? LSM_GHOST : 0;
        h->lay.stride[d] = s;
        if (d == 0 && N >= 2 && align_rows) {
            // rows start on a 64-byte boundary: interior node 0 of every row sits ALIGN elements into its pitch (the ghosts
            // fill the end of the line before it) and the pitch is a multiple of ALIGN — a wave's 32 x 8-byte row segment is
            // then four whole lines instead of five pieces, and the ghost nodes of a row end share one line
            const long long A = dtype == LSM_DTYPE_F32 ? 16 : 8;
            const long long lead = (LSM_GHOST + A - 1) / A * A;
            h->lay.origin += lead;
            s = (lead + h->lay.n[d] + h->lay.g[d] + A - 1) / A * A;
        } else {
            h->lay.origin += h->lay.g[d] * s;
            s *= h->lay.n[d] + 2 * h->lay.g[d];
        }
    }
    h->lay.total = s;
    h->dxmin = 0;
    for (int d = 0; d < 3; ++d) {
        if (d < N) {
            h->h[d] = (grid->hc[d] - grid->lc[d]) / (double)(grid->n[d] - 1);   // src/meshes.jl:110
            h->h2[d] = h->h[d] * h->h[d];
            h->inv_h[d] = 1.0 / h->h[d];
            h->inv_h2[d] = 1.0 / h->h2[d];
            h->dxmin = d == 0 ? h->h[d] : (h->h[d] < h->dxmin ? h->h[d] : h->dxmin);
            if (!(h->h[d] > 0)) { delete h; return fail(nullptr, LSM_ERR_INVALID, "lsm_create: hc must exceed lc"); }
        } else {
            h->h[d] = h->h2[d] = h->inv_h[d] = h->inv_h2[d] = 1.0;
        }
        for (int sd = 0; sd < 2; ++sd)
            for (int k = 1; k <= LSM_GHOST; ++k)
                for (int j = 0; j < 8; ++j) {
                    const LsmBc& b = h->bc[d][sd];
                    h->w[d][sd][k - 1][j] =
                        (b.kind == LSM_BC_EXTRAPOLATION && j <= b.degree) ? lagrange_w(j, k, b.degree) : 0.0;
                }
        if (d < N)
            for (int sd = 0; sd < 2; ++sd)
                if (h->bc[d][sd].kind == LSM_BC_EXTRAPOLATION && h->bc[d][sd].degree + 1 > h->nloc[d]) {
                    delete h;
                    return fail(nullptr, LSM_ERR_INVALID, "lsm_create: extrapolation degree exceeds the node count");
                }
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    h->own_stream = true;
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_partial, sizeof(double) * 2 * MAXB);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_tail_ctr, LSM_TAIL_SLOTS * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(h->d_tail_ctr, 0, LSM_TAIL_SLOTS * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_flag, sizeof(int));
    h->cfl_stream = nullptr; h->c_partial = h->c_result = h->ch_result = nullptr; h->c_flag = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->cfl_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->c_partial, sizeof(double) * 2 * MAXB);
    if (e == hipSuccess) e = hipMalloc((void**)&h->c_flag, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&h->c_result, sizeof(double) * 2);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->ch_result, sizeof(double) * 2, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_w, sizeof(h->w));
    if (e == hipSuccess) e = hipMemcpy(h->d_w, h->w, sizeof(h->w), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_result, sizeof(double) * 16);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_result, sizeof(double) * 16, hipHostMallocMapped);
    if (e == hipSuccess) { memset(h->h_result, 0, sizeof(double) * 16); e = hipHostGetDevicePointer((void**)&h->h_result_dev, h->h_result, 0); }
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_pf_flag, sizeof(int) * 4);
    if (e != hipSuccess) {
        std::string m = std::string("lsm_create: ") + hipGetErrorString(e);
        delete h;
        return fail(nullptr, LSM_ERR_HIP, m);
    }
    *out = h;
    return LSM_OK;
}

void lsm_destroy(LsmHandle* h) {
    if (!h) return;
    (void)lsm_comm_detach(h);
    (void)hipSetDevice(h->device);
    if (h->d_stamp) (void)hipFree(h->d_stamp);
    if (h->d_tail_ctr) (void)hipFree(h->d_tail_ctr);
    (void)hipStreamSynchronize(h->stream);
    for (auto e : h->ev_start) (void)hipEventDestroy(e);
    for (auto e : h->ev_stop) (void)hipEventDestroy(e);
    for (auto& e : h->cfl_cand) (void)hipFree(e.d_cand);
    if (h->d_cand_count) (void)hipFree(h->d_cand_count);
    (void)hipFree(h->d_partial);
    (void)hipFree(h->d_flag);
    if (h->cfl_stream) (void)hipStreamDestroy(h->cfl_stream);
    (void)hipFree(h->c_partial); (void)hipFree(h->c_flag); (void)hipFree(h->c_result);
    if (h->ch_result) (void)hipHostFree(h->ch_result);
    (void)hipFree(h->d_w);
    if (h->d_ring) { (void)hipFree(h->d_ring); (void)hipFree(h->d_miss); (void)hipFree(h->d_count); }
    if (h->d_work) { (void)hipFree(h->d_work); (void)hipFree(h->d_act_list); (void)hipFree(h->d_work_list); (void)hipFree(h->d_lcounts); (void)hipFree(h->d_tiles_old); }
    (void)hipFree(h->d_result);
    if (h->d_pf_flag) (void)hipFree(h->d_pf_flag);
    reinit_workspace_free(h->reinit_ws);
    (void)hipHostFree(h->h_result);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// adopt an external stream (e.g. the one the caller's array library works on)
int lsm_set_stream(LsmHandle* h, void* stream) {
    if (!h) return LSM_ERR_INVALID;
    if (h->own_stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return LSM_OK;
}

int lsm_sync(LsmHandle* h) {
    if (!h) return LSM_ERR_INVALID;
    LSM_SYNC(h);
    return LSM_OK;
}

int lsm_layout(const LsmHandle* h, LsmLayout* out) {
    if (!h || !out) return LSM_ERR_INVALID;
    *out = h->lay;
    return LSM_OK;
}

static int copy_interior(LsmHandle* h, void* dev, void* host, bool to_dev, size_t es) {
    LSM_HIP(h, hipSetDevice(h->device));
    const size_t row = es * (size_t)h->nloc[0];
    for (int k = 0; k < h->nloc[2]; ++k) {
        char* d = (char*)dev + es * (size_t)(h->lay.origin + (long long)k * h->lay.stride[2]);
        char* s = (char*)host + es * ((size_t)k * h->nloc[0] * h->nloc[1]);
        if (to_dev)
            LSM_HIP(h, hipMemcpy2DAsync(d, es * h->lay.stride[1], s, row, row, h->nloc[1], hipMemcpyHostToDevice, h->stream));
        else
            LSM_HIP(h, hipMemcpy2DAsync(s, row, d, es * h->lay.stride[1], row, h->nloc[1], hipMemcpyDeviceToHost, h->stream));
    }
    LSM_SYNC(h);
    return LSM_OK;
}
int lsm_upload(LsmHandle* h, void* dev_padded, const void* host_dense) {
    if (!h || !dev_padded || !host_dense) return LSM_ERR_INVALID;
    return copy_interior(h, dev_padded, (void*)host_dense, true, esize(h));
}
int lsm_download(LsmHandle* h, const void* dev_padded, void* host_dense) {
    if (!h || !dev_padded || !host_dense) return LSM_ERR_INVALID;
    return copy_interior(h, (void*)dev_padded, host_dense, false, esize(h));
}
// the fp64 side arrays of a handle of either dtype: coefficient fields, frozen masks, frozen signs
int lsm_upload_f64(LsmHandle* h, void* dev_padded, const void* host_dense) {
    if (!h || !dev_padded || !host_dense) return LSM_ERR_INVALID;
    return copy_interior(h, dev_padded, (void*)host_dense, true, sizeof(double));
}
int lsm_download_f64(LsmHandle* h, const void* dev_padded, void* host_dense) {
    if (!h || !dev_padded || !host_dense) return LSM_ERR_INVALID;
    return copy_interior(h, (void*)dev_padded, host_dense, false, sizeof(double));
}

static int fill_ghosts_fused(LsmHandle* h, void* field, int mb, int me, int fill_last, hipStream_t s, bool skip_x = false, bool skip_y = false,
                             int depth = 0 /* 0: the handle's current step depth (LSM_GHOST outside a step) */) {
    const int N = h->grid.ndim;
    GhostAllArgs a;
    for (int e = 0; e < 3; ++e) a.n[e] = h->nloc[e];
    a.s1 = h->lay.stride[1]; a.s2 = h->lay.stride[2]; a.origin = h->lay.origin;
    for (int d = 0; d < 3; ++d)
        for (int sd = 0; sd < 2; ++sd) { a.kind[d][sd] = h->bc[d][sd].kind; a.degree[d][sd] = h->bc[d][sd].degree; }
    a.w = h->d_w;
    a.v = field; a.f32 = is_f32(h);
    a.mb = mb; a.me = me; a.fill_last = fill_last;
    a.skip_x = skip_x ? 1 : 0;
    a.skip_y = skip_y && N == 3 ? 1 : 0;
    const bool full_depth = h->tune.ghost_full_depth != 0;
    if (depth == 0) depth = h->ghost_depth;
    a.depth = (depth < 1 || depth > LSM_GHOST || full_depth) ? LSM_GHOST : depth;
    launch_ghost_fill_all(N, a, s);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

int lsm_fill_ghosts(LsmHandle* h, void* field, int dim_mask, void* stream) {
    if (!h || !field) return LSM_ERR_INVALID;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    const int N = h->grid.ndim;
    if ((dim_mask & ((1 << N) - 1)) == ((1 << N) - 1))
        return fill_ghosts_fused(h, field, 0, h->nloc[N - 1], 1, s);   // one launch, bit-identical to the passes below
    for (int d = 0; d < N; ++d) {
        if (!((dim_mask >> d) & 1)) continue;
        if (h->bc[d][0].kind == LSM_BC_NONE && h->bc[d][1].kind == LSM_BC_NONE) continue;
        GhostArgs a;
        for (int e = 0; e < 3; ++e) a.n[e] = h->nloc[e];
        a.s1 = h->lay.stride[1]; a.s2 = h->lay.stride[2]; a.origin = h->lay.origin;
        a.dim = d;
        for (int sd = 0; sd < 2; ++sd) { a.kind[sd] = h->bc[d][sd].kind; a.degree[sd] = h->bc[d][sd].degree; }
        memcpy(a.w, h->w[d], sizeof(a.w));
        a.v = field; a.f32 = is_f32(h);
        launch_ghost_fill(N, a, s);
    }
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

int lsm_fill_ghosts_planes(LsmHandle* h, void* field, int64_t m_begin, int64_t m_end, int fill_last, void* stream) {
    if (!h || !field) return LSM_ERR_INVALID;
    const int N = h->grid.ndim;
    if (m_begin < 0 || m_end > h->nloc[N - 1] || m_begin > m_end) return fail(h, LSM_ERR_INVALID, "lsm_fill_ghosts_planes: bad plane range");
    // inside a step whose stage kernels resolve x ghosts in their loads (h->xredirect, see XRedirect) the row ends are left alone
    return fill_ghosts_fused(h, field, (int)m_begin, (int)m_end, fill_last, stream ? (hipStream_t)stream : h->stream, h->xredirect, h->yredirect);
}

static void fill_coeff(const LsmCoeff& c, double t, CoeffArgs& o) {
    o.kind = c.kind;
    for (int k = 0; k < 4; ++k) o.v[k] = c.value[k];
    o.tfac = c.time_kind == LSM_TIME_COS ? cos(M_PI * t / c.time_param) : 1.0;
    for (int k = 0; k < 3; ++k) { o.f[k] = (const double*)c.field[k]; o.sep[k] = c.sep[k]; }
}

static void base_args(const LsmHandle* h, StageArgs& a) {
    memset(&a, 0, sizeof(a));
    for (int d = 0; d < 3; ++d) {
        a.n[d] = h->nloc[d]; a.goff[d] = h->goff[d]; a.gn[d] = h->gn[d];
        a.lc[d] = h->grid.lc[d]; a.h[d] = h->h[d]; a.h2[d] = h->h2[d]; a.inv_h[d] = h->inv_h[d]; a.inv_h2[d] = h->inv_h2[d];
    }
    a.s1 = h->lay.stride[1]; a.s2 = h->lay.stride[2]; a.origin = h->lay.origin;
    a.dxmin = h->dxmin;
    a.uniform_h = 1;                      // 1/h² equal in every used dimension: the Godunov sums are scaled once
    for (int d = 1; d < h->grid.ndim; ++d) a.uniform_h = a.uniform_h && h->inv_h2[d] == h->inv_h2[0];
    a.mask = h->band_mask; a.tile_active = h->band_tiles; a.mc = h->band_mc;
    a.tile_list = h->band_list; a.ntile_list = h->band_nlist;
    if (h->band_list) { a.brick_list = h->d_act_list; a.nbrick_list = h->nact; }
    a.f32 = is_f32(h);
    a.tail_ctr = nullptr; a.tail_wgs = 0;
    a.tail_ring = h->d_tail_ctr;          // stage_impl withdraws it from launches on a caller's stream
    a.tail_slot_host = const_cast<unsigned*>(&h->tail_ticket);
    a.xredirect = h->xredirect ? 1 : 0;
    a.xkind[0] = h->bc[0][0].kind; a.xkind[1] = h->bc[0][1].kind;
    a.yredirect = h->yredirect ? 1 : 0;
    a.ykind[0] = h->bc[1][0].kind; a.ykind[1] = h->bc[1][1].kind;
    for (int sd = 0; sd < 2; ++sd) {
        const LsmBc& mb = h->bc[h->grid.ndim - 1][sd];
        a.mredirect[sd] = h->mredirect && mb.kind == LSM_BC_EXTRAPOLATION && mb.degree == 0 ? 1 : 0;
    }
    a.stamp = h->d_stamp;
    a.tune = &h->tune;
    a.nbig = 0; a.mc_tail = 0;
}

static int check_coeff(LsmHandle* h, const LsmCoeff& c, int ncomp) {
    if (c.kind < 0 || c.kind > LSM_COEFF_FIELD) return fail(h, LSM_ERR_INVALID, "bad coefficient kind");
    if (c.kind == LSM_COEFF_FIELD)
        for (int k = 0; k < ncomp; ++k)
            if (!c.field[k]) return fail(h, LSM_ERR_INVALID, "FIELD coefficient with a null component");
    if (c.kind == LSM_COEFF_SEPARABLE)
        for (int k = 0; k < ncomp; ++k)
            if (!c.sep[k]) return fail(h, LSM_ERR_INVALID, "SEPARABLE coefficient with a null table");
    if (c.kind == LSM_COEFF_ROTATION && h->grid.ndim < 2) return fail(h, LSM_ERR_INVALID, "ROTATION needs ndim >= 2");
    return LSM_OK;
}

static int profile_pair(LsmHandle* h, hipEvent_t* a, hipEvent_t* b) {
    if (h->ev_used == h->ev_start.size()) {
        hipEvent_t x, y;
        LSM_HIP(h, hipEventCreate(&x));
        LSM_HIP(h, hipEventCreate(&y));
        h->ev_start.push_back(x);
        h->ev_stop.push_back(y);
    }
    *a = h->ev_start[h->ev_used];
    *b = h->ev_stop[h->ev_used];
    h->ev_used++;
    return LSM_OK;
}

static int stage_impl(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out, void* out2,
                      int base_mode, double cdt, double cdt2, double t_stage, int mb, int me, void* stream) {
    if (!h || !terms || !psi || !out) return h ? fail(h, LSM_ERR_INVALID, "lsm_stage: null argument") : LSM_ERR_INVALID;
    if (nterms < 1 || nterms > LSM_MAX_TERMS) return fail(h, LSM_ERR_INVALID, "lsm_stage: nterms must be in 1..8");
    if (base_mode != LSM_BASE_PSI && !phin) return fail(h, LSM_ERR_INVALID, "lsm_stage: phin required for this base mode");
    if (psi == out || psi == out2) return fail(h, LSM_ERR_INVALID, "lsm_stage: out must not alias the stencil input psi");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    const int N = h->grid.ndim;
    // the kernel addresses a plane (the whole array in 1-D, a row in 2-D) through a 2 GiB buffer descriptor
    if ((N == 1 ? h->lay.total : h->lay.stride[N - 1]) * 8ll >= (1ll << 31))
        return fail(h, LSM_ERR_INVALID, "lsm_stage: a plane of the padded array must be smaller than 2 GiB");
    int i = 0;
    bool first = true;
    while (i < nterms) {
        Combo c{0, 0, 0, 0};
        StageArgs a;
        base_args(h, a);
        // the dynamic tail's counters are ordered by the handle's own stream: a launch on any other stream (which may run
        // concurrently with one of ours) takes the static tail
        if (s != h->stream) a.tail_ring = nullptr;
        int cnt = 0;
        while (i < nterms && cnt < NSLOTS) {
            const LsmTerm& tm = terms[i];
            Combo trial = c;
            int slot;
            switch (tm.kind) {
            case LSM_TERM_ADVECTION:
                if (c.adv) goto flush;
                if (tm.scheme != LSM_SCHEME_UPWIND && tm.scheme != LSM_SCHEME_WENO5) return fail(h, LSM_ERR_INVALID, "bad scheme");
                trial.adv = tm.scheme == LSM_SCHEME_WENO5 ? 2 : 1; slot = SLOT_ADV; break;
            case LSM_TERM_NORMAL_MOTION: if (c.nm) goto flush; trial.nm = 1; slot = SLOT_NM; break;
            case LSM_TERM_CURVATURE: if (c.curv) goto flush; trial.curv = 1; slot = SLOT_CURV; break;
            case LSM_TERM_EIKONAL: if (c.eik) goto flush; trial.eik = tm.s0 ? 1 : 2; slot = SLOT_EIK; break;
            default: return fail(h, LSM_ERR_INVALID, "lsm_stage: bad term kind");
            }
            if (!combo_available(trial)) break;
            c = trial;
            a.order[cnt++] = slot;
            if (slot == SLOT_ADV) { int r = check_coeff(h, tm.coeff, N); if (r) return r; fill_coeff(tm.coeff, t_stage, a.adv); a.adv_scheme = tm.scheme; }
            if (slot == SLOT_NM) { int r = check_coeff(h, tm.coeff, 1); if (r) return r; fill_coeff(tm.coeff, t_stage, a.nm); }
            if (slot == SLOT_CURV) { int r = check_coeff(h, tm.coeff, 1); if (r) return r; fill_coeff(tm.coeff, t_stage, a.curv); }
            if (slot == SLOT_EIK) a.s0 = (const double*)tm.s0;
            ++i;
        }
    flush:
        if (cnt == 0) return fail(h, LSM_ERR_INVALID, "lsm_stage: internal planner error");
        a.nterms = cnt;
        a.natural = 1;
        for (int k = 1; k < cnt; ++k)
            if (a.order[k] <= a.order[k - 1]) a.natural = 0;
        a.psi = (const double*)psi;
        a.out = (double*)out;
        a.out2 = (double*)out2;
        a.cdt = cdt; a.cdt2 = cdt2;
        a.mb = mb; a.me = me;
        if (first) { a.base_mode = base_mode; a.phin = (const double*)(base_mode == LSM_BASE_PSI ? psi : phin); a.out2_accum = 0; }   // the kernel always loads phin
        else       { a.base_mode = LSM_BASE_OTHER; a.phin = (const double*)out; a.out2_accum = 1; }
        switch (a.base_mode) {
        case LSM_BASE_PSI: a.base_a = 0.0; a.base_b = 1.0; break;
        case LSM_BASE_RK3_S2: a.base_a = 0.75; a.base_b = 0.25; break;
        case LSM_BASE_RK3_S3: a.base_a = 1.0 / 3; a.base_b = 2.0 / 3; break;
        default: a.base_a = 1.0; a.base_b = 0.0; break;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = h->prof && (h->prof_seen++ % (unsigned long long)h->prof_every) == 0;
        if (timed) { int r = profile_pair(h, &e0, &e1); if (r) return r; LSM_HIP(h, hipEventRecord(e0, s)); }
        int r;
        if (h->mode == LSM_MODE_STRICT)
            r = N == 1 ? launch_stage_strict_1d(c, a, s) : (N == 2 ? launch_stage_strict_2d(c, a, s) : launch_stage_strict_3d(c, a, s));
        else
            r = N == 1 ? launch_stage_fast_1d(c, a, s) : (N == 2 ? launch_stage_fast_2d(c, a, s) : launch_stage_fast_3d(c, a, s));
        if (r) return fail(h, LSM_ERR_INVALID, "lsm_stage: kernel combination not instantiated");
        if (timed) LSM_HIP(h, hipEventRecord(e1, s));
        first = false;
    }
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

int lsm_stage(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out, void* out2,
              int base_mode, double cdt, double cdt2, double t_stage, void* stream) {
    if (!h) return LSM_ERR_INVALID;
    return stage_impl(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t_stage, 0, h->nloc[h->grid.ndim - 1], stream);
}

int lsm_stage_planes(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out, void* out2,
                     int base_mode, double cdt, double cdt2, double t_stage, int64_t m_begin, int64_t m_end, void* stream) {
    if (!h) return LSM_ERR_INVALID;
    const int N = h->grid.ndim;
    if (N < 2) return fail(h, LSM_ERR_INVALID, "lsm_stage_planes: needs ndim >= 2");
    if (m_begin < 0 || m_end > h->nloc[N - 1] || m_begin > m_end) return fail(h, LSM_ERR_INVALID, "lsm_stage_planes: bad plane range");
    if (m_begin == m_end) return LSM_OK;
    return stage_impl(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t_stage, (int)m_begin, (int)m_end, stream);
}

// Julia's min: NaN-propagating
static double jl_min(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? NAN : (b < a ? b : a); }

static void cfl_args(const LsmHandle* h, const LsmTerm& tm, double t, CflArgs& a) {
    memset(&a, 0, sizeof(a));
    for (int d = 0; d < 3; ++d) {
        a.n[d] = h->nloc[d]; a.goff[d] = h->goff[d]; a.gn[d] = h->gn[d];
        a.lc[d] = h->grid.lc[d]; a.h[d] = h->h[d];
    }
    a.s1 = h->lay.stride[1]; a.s2 = h->lay.stride[2]; a.origin = h->lay.origin;
    a.term_kind = tm.kind;
    fill_coeff(tm.coeff, t, a.coeff);
}

int lsm_compute_cfl(LsmHandle* h, const LsmTerm* terms, int nterms, const void* phi, double t, double* dt_out) {
    if (!h || !terms || !dt_out) return h ? fail(h, LSM_ERR_INVALID, "lsm_compute_cfl: null argument") : LSM_ERR_INVALID;
    if (nterms < 1 || nterms > LSM_MAX_TERMS) return fail(h, LSM_ERR_INVALID, "lsm_compute_cfl: nterms must be in 1..8");
    (void)phi;
    const int N = h->grid.ndim;
    double best = 0;
    for (int k = 0; k < nterms; ++k) {
        const LsmTerm& tm = terms[k];
        double dt;
        if (tm.kind == LSM_TERM_EIKONAL) {
            dt = h->dxmin;                                                 // src/levelsetterms.jl:250
        } else if (tm.coeff.kind == LSM_COEFF_CONST) {                     // node-independent: evaluate once
            if (tm.kind == LSM_TERM_CURVATURE) {
                dt = (h->dxmin * h->dxmin) / (2 * fabs(tm.coeff.value[0]));      // :123-127
            } else {
                double s = 0;
                for (int d = 0; d < N; ++d) {
                    const double u = tm.kind == LSM_TERM_ADVECTION ? tm.coeff.value[d] : tm.coeff.value[0];
                    const double q = fabs(u) / h->h[d];
                    s = d == 0 ? q : s + q;
                }
                dt = 1 / s;                                                      // :90-96, :172-178
            }
        } else {
            if (tm.kind < 0 || tm.kind > LSM_TERM_EIKONAL) return fail(h, LSM_ERR_INVALID, "bad term kind");
            int r = check_coeff(h, tm.coeff, tm.kind == LSM_TERM_ADVECTION ? N : 1);
            if (r) return r;
            // a catalogued analytic coefficient without time factor gives the same minimum every step
            const bool cacheable = h->cfl_cache_on && (tm.coeff.kind == LSM_COEFF_ROTATION ||
                                   (tm.coeff.kind == LSM_COEFF_SEPARABLE && tm.coeff.time_kind == LSM_TIME_ONE));
            bool hit = false;
            if (cacheable)
                for (auto& e : h->cfl_cache)
                    if (memcmp(&e.first, &tm, sizeof(LsmTerm)) == 0) { dt = e.second; hit = true; break; }
            if (hit) { best = k == 0 ? dt : jl_min(best, dt); continue; }
            if (h->cfl_prefetched && h->band_cfl.slot[k] >= 0) {      // lsm_compute_cfl_band: this term's Δt came home with lsm_band_status
                dt = h->band_cfl.dt[k];
                best = k == 0 ? dt : jl_min(best, dt);
                continue;
            }
            CflArgs a;
            cfl_args(h, tm, t, a);
            // ϕ-independent coefficient, dense field: the reduction runs on the handle's CFL stream with its own scratch and
            // does not queue behind the stages on the main stream.  A table is waited for once, the first time it is seen (its
            // upload was ordered on the main stream).
            // Only while the coefficient cache is on: lsm_cfl_cache(h, 0) is how a caller says "tables may be rewritten in
            // place between calls" (hooks), and such writes are ordered on the main stream only.
            const bool side = h->cfl_cache_on && !h->band_mask && tm.coeff.kind != LSM_COEFF_FIELD;
            if (side && tm.coeff.kind == LSM_COEFF_SEPARABLE)
                for (int c = 0; c < 3; ++c) {
                    const void* tp = tm.coeff.sep[c];
                    if (!tp) continue;
                    bool seen = false;
                    for (const void* q : h->cfl_seen) seen = seen || q == tp;
                    if (!seen) {
                        LSM_SYNC(h);
                        if (h->cfl_seen.size() > 64) h->cfl_seen.clear();
                        h->cfl_seen.push_back(tp);
                    }
                }
            hipStream_t cs = side ? h->cfl_stream : h->stream;
            double* const dpart = side ? h->c_partial : h->d_partial;
            int* const dflag = side ? h->c_flag : h->d_flag;
            double* const dres = side ? h->c_result : h->d_result;
            double* const hres = side ? h->ch_result : h->h_result;
            a.partial = dpart;
            a.nanflag = dflag;
            a.mask = h->band_mask;
            if (h->band_mask && h->band_tiles && N > 1) {
                const BandArgs ba = band_args(h, h->band_mc, nullptr);
                a.tile_active = h->band_tiles;
                a.tx = ba.tx; a.ty = ba.ty; a.tm = ba.tm; a.nbx = ba.nbx; a.nby = ba.nby;
            }
            int nb = cfl_blocks(N, h->nloc);
            if (nb > MAXB) nb = MAXB;
            LSM_HIP(h, hipMemsetAsync(dflag, 0, sizeof(int), cs));
            // (on a band the exact divisions run on the few band nodes only: one pass)
            const bool two_pass = !h->band_mask && tm.coeff.kind != LSM_COEFF_FIELD && tm.kind != LSM_TERM_CURVATURE;
            // u(x)·g(t): the nodes that can attain the maximum are the same for every t (recorded once, with g = 1)
            const bool sep_time = two_pass && h->cfl_cache_on && !h->band_mask && tm.coeff.kind == LSM_COEFF_SEPARABLE &&
                                  tm.coeff.time_kind != LSM_TIME_ONE && fabs(a.coeff.tfac) > 1e-200;
            const LsmHandle::CflCand* cand = nullptr;
            if (sep_time) {
                for (auto& e : h->cfl_cand)
                    if (memcmp(&e.key, &tm, sizeof(LsmTerm)) == 0) { cand = &e; break; }
                if (!cand) {
                    const unsigned cap = 8192;
                    if (!h->d_cand_count) LSM_HIP(h, hipMalloc((void**)&h->d_cand_count, sizeof(unsigned)));
                    long long* buf = nullptr;
                    LSM_HIP(h, hipMalloc((void**)&buf, cap * sizeof(long long)));
                    CflArgs b = a;
                    b.partial = h->d_partial; b.nanflag = h->d_flag;      // one-off search on the main stream, synchronised
                    b.coeff.tfac = 1.0;
                    b.cand = buf; b.cand_count = h->d_cand_count; b.cand_cap = cap;
                    LSM_HIP(h, hipMemsetAsync(h->d_cand_count, 0, sizeof(unsigned), h->stream));
                    launch_cfl(N, b, nb, 0, nullptr, h->stream);
                    launch_cfl_final(h->d_partial, nb, h->d_flag, h->d_result, tm.kind, h->dxmin, 0, h->stream);
                    launch_cfl(N, b, nb, 2, h->d_result + 1, h->stream);
                    unsigned cnt = 0;
                    int flag = 0;
                    LSM_HIP(h, hipMemcpyAsync(&cnt, h->d_cand_count, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
                    LSM_HIP(h, hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
                    LSM_SYNC(h);
                    LSM_HIP(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
                    if (!flag && cnt >= 1 && cnt <= cap) {     // NaN tables or a flat maximum: keep sweeping the grid
                        if (h->cfl_cand.size() > 16) { for (auto& e : h->cfl_cand) (void)hipFree(e.d_cand); h->cfl_cand.clear(); }
                        h->cfl_cand.push_back({tm, buf, cnt});
                        cand = &h->cfl_cand.back();
                    } else {
                        (void)hipFree(buf);
                    }
                }
            }
            int nlisted = 0;
            if (h->band_mask && N == 3 && a.tile_active && have_lists(h, h->band_tiles, h->band_mc))
                nlisted = launch_cfl_band_list(a, h->d_act_list, h->nact, nullptr, MAXB, cs);   // one workgroup per active tile
            if (nlisted > 0) {
                nb = nlisted;
            } else if (cand) {
                CflArgs b = a;
                b.cand = cand->d_cand;
                launch_cfl_candidates(N, b, cand->count, cs);
                nb = 1;
            } else if (two_pass) {
                launch_cfl(N, a, nb, 0, nullptr, cs);
                launch_cfl_final(dpart, nb, dflag, dres, tm.kind, h->dxmin, 0, cs);
                launch_cfl(N, a, nb, 1, dres + 1, cs);
            } else {
                launch_cfl(N, a, nb, 1, nullptr, cs);
            }
            launch_cfl_final(dpart, nb, dflag, dres, tm.kind, h->dxmin, 1, cs);
            LSM_HIP(h, hipMemcpyAsync(hres, dres, sizeof(double), hipMemcpyDeviceToHost, cs));
            LSM_HIP(h, hipStreamSynchronize(cs));
            dt = hres[0];
            if (cacheable) {
                if (h->cfl_cache.size() > 16) h->cfl_cache.clear();
                h->cfl_cache.emplace_back(tm, dt);
            }
        }
        best = k == 0 ? dt : jl_min(best, dt);
    }
    *dt_out = best;
    return LSM_OK;
}

static int check_single_device(LsmHandle* h) {
    const int N = h->grid.ndim;
    if (h->bc[N - 1][0].kind == LSM_BC_NONE || h->bc[N - 1][1].kind == LSM_BC_NONE)
        return fail(h, LSM_ERR_INVALID,
                    "this entry point works on a whole grid: the handle is a slab of a multi-GPU grid (attach a communicator — "
                    "lsm_comm_attach_rccl / lsm_comm_attach_local — or drive the slab with lsm_stage + lsm_fill_ghosts + lsm_halo_*)");
    return LSM_OK;
}
// a slab handle with a communicator attached: the stages exchange their ghost planes
static bool is_slab(const LsmHandle* h) {
    const int N = h->grid.ndim;
    return h->comm != nullptr && (h->bc[N - 1][0].kind == LSM_BC_NONE || h->bc[N - 1][1].kind == LSM_BC_NONE);
}
static int run_hook(LsmHandle* h, LsmStageHook hook, void* user, int stage, const void* field, double t) {
    if (!hook) return LSM_OK;
    LSM_SYNC(h);
    if (hook(user, stage, field, t)) return fail(h, LSM_ERR_INVALID, "stage hook requested abort");
    return LSM_OK;
}
#define LSM_TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)

// Steps in FAST mode (whole grid or slab): when both x faces copy ONE node (NeumannBC = degree-0 extrapolation, periodic, symmetry) the
// stage kernels resolve x ghosts in their loads (StageArgs::xredirect) and the fill before a stage leaves the ends of the
// 262 k rows of a 512³ grid alone — the scattered two thirds of its time.  The values read are the ones the fill would have
// stored, except that a copied -0.0 stays -0.0 (the reference's acc = 0 + w·value makes it +0.0): FAST's tolerance, not
// STRICT's bit pattern.  Not with a stage hook (it may read the field through its own kernels).  LSM_XREDIRECT=0: A/B switch.
struct XRedirect {
    LsmHandle* h;
    bool on;
    XRedirect(LsmHandle* h_, LsmStageHook hook, const LsmTerm* terms, int nterms) : h(h_), on(false) {
        // ghost layers the stencils of this step read (stage_kernel.h, halo_of): the fills inside the step write no more than that.
        // Not with a hook: it may read the field through its own kernels.
        int depth = 0;
        for (int k = 0; terms && k < nterms; ++k) {
            const int g = terms[k].kind == LSM_TERM_ADVECTION ? (terms[k].scheme == LSM_SCHEME_WENO5 ? 3 : 1) : (terms[k].kind == LSM_TERM_CURVATURE ? 1 : 2);
            depth = g > depth ? g : depth;
        }
        h->ghost_depth = (hook || depth < 1) ? LSM_GHOST : depth;
        const bool off = !h->tune.xredirect;                                   // LsmTuning: the fills materialise these ghosts
        auto copies = [&](int d, int sd) {
            const int k = h->bc[d][sd].kind;
            return k == LSM_BC_PERIODIC || k == LSM_BC_SYMMETRY || (k == LSM_BC_EXTRAPOLATION && h->bc[d][sd].degree == 0);
        };
        on = !off && !hook && h->mode != LSM_MODE_STRICT && h->grid.ndim >= 2 && h->nloc[0] >= 2 * LSM_GHOST + 2 && copies(0, 0) && copies(0, 1);
        h->xredirect = on;
        // dimension 2 of a 3-D grid likewise (in 2-D it is the march axis: its ghost rows stay in memory)
        h->yredirect = on && h->grid.ndim == 3 && h->nloc[1] >= 2 * LSM_GHOST + 2 && copies(1, 0) && copies(1, 1);
        // The march axis (round 4): a NeumannBC ghost plane is a copy of the boundary plane, so the march CLAMPS there (two scalars in
        // the kernel, the plane pointers still advance by one plane — the general map of round 2, with its recomputed pointers, cost
        // the kernel its scalar registers).  When every face of the march axis is NeumannBC (or a slab interface, whose planes are
        // exchanged) and the other dimensions are served by the loads, a step of the headline equation or of BASELINE config 2
        // launches no ghost fill at all.
        const int L = h->grid.ndim - 1;
        auto clamps = [&](int sd) { const LsmBc& b = h->bc[L][sd]; return b.kind == LSM_BC_NONE || (b.kind == LSM_BC_EXTRAPOLATION && b.degree == 0); };
        h->mredirect = on && h->tune.mredirect && (h->grid.ndim == 2 || h->yredirect) && h->nloc[L] >= 2 * LSM_GHOST + 2 && clamps(0) && clamps(1);
    }
    ~XRedirect() { h->xredirect = false; h->yredirect = false; h->mredirect = false; h->ghost_depth = LSM_GHOST; }
    int fill(void* field) const {
        if (!on) return lsm_fill_ghosts(h, field, 7, nullptr);
        if (h->mredirect) return LSM_OK;         // nothing of the padded array's ghost layers is read inside this step
        return fill_ghosts_fused(h, field, 0, h->nloc[h->grid.ndim - 1], 1, h->stream, true, h->yredirect);
    }
};

// One stage of a slab followed by its ghost resolution.  The LSM_GHOST+1 planes next to each slab
// interface are updated and ghost-filled first, their exchange is started, and the interior is updated while the planes
// travel; then the physical-BC ghost planes of the end ranks.  Every node is computed by the same kernel from the same
// inputs as in the plain stage -> ghost fill -> exchange order (lsm_comm_set_overlap(h, 0) selects it), so the
// results are identical.  The exchanged planes carry their own ghosts of the leading dimensions: the corner
// composition of _getindexbc (src/meshfield.jl:248-260) is preserved.
static int stage_slab(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out, void* out2,
                      int base_mode, double cdt, double cdt2, double t) {
    const int N = h->grid.ndim;
    const int nloc = h->nloc[N - 1];
    const int B = h->ghost_depth + 1;            // the planes the exchange sends (lsm_halo_start); +1: the periodic wrap sends planes shifted by one node
    if (!lsm_comm_overlap(h) || N < 2 || nloc < 2 * B + 1) {
        LSM_TRY(lsm_stage(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, nullptr));
        if (h->mredirect) {}                                  // every ghost the step reads is served by the loads or exchanged
        else if (h->xredirect) LSM_TRY(fill_ghosts_fused(h, out, 0, nloc, 1, h->stream, true, h->yredirect));
        else LSM_TRY(lsm_fill_ghosts(h, out, 7, nullptr));
        return lsm_halo_exchange(h, out);
    }
    const int64_t edge[2][2] = {{0, B}, {nloc - B, nloc}};
    for (auto& e : edge) {
        LSM_TRY(lsm_stage_planes(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, e[0], e[1], nullptr));
        LSM_TRY(lsm_fill_ghosts_planes(h, out, e[0], e[1], 0, nullptr));
    }
    LSM_TRY(lsm_halo_start(h, out));
    LSM_TRY(lsm_stage_planes(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, B, nloc - B, nullptr));
    LSM_TRY(lsm_fill_ghosts_planes(h, out, B, nloc - B, 0, nullptr));
    LSM_TRY(lsm_halo_wait(h));
    if (h->mredirect) return LSM_OK;                         // NeumannBC end faces: the march clamps at the boundary plane
    return lsm_fill_ghosts(h, out, 1 << (N - 1), nullptr);   // physical ghost planes of the last dimension (interfaces are skipped)
}

// _advance! of a slab (communicator attached).  phi's ghosts — boundary conditions AND the neighbours' planes — must be
// valid on entry: they are on return of the previous step; after the field was written from outside call
// lsm_fill_ghosts(7) + lsm_halo_exchange first.
static int advance_slab_stages(LsmHandle* h, int integ, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2, double tc,
                               double dt, LsmStageHook hook, void* user);
static int advance_slab(LsmHandle* h, int integ, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2, double tc,
                        double dt, LsmStageHook hook, void* user) {
    const XRedirect xr(h, hook, terms, nterms);   // the x ghosts of a slab's planes (its own and the received ones) are resolved by the loads too
    // The previous step left ϕ with as many valid ghost layers (boundary conditions and neighbours' planes) as ITS stencils read.
    // A step whose stencils reach farther — the term list changed in between; the same on every rank — refreshes all of them first.
    if (h->ghost_depth > h->slab_depth_valid) {
        const int d = h->ghost_depth;
        h->ghost_depth = LSM_GHOST;
        int r = lsm_fill_ghosts(h, phi, 7, nullptr);
        if (r == LSM_OK) r = lsm_halo_exchange(h, phi);
        h->ghost_depth = d;
        LSM_TRY(r);
    }
    struct DepthNote {          // what ϕ's ghosts hold when the step is over (also after a failure half-way: then nothing is promised)
        LsmHandle* h; int d; bool ok;
        ~DepthNote() { h->slab_depth_valid = ok ? d : 0; }
    } note{h, h->ghost_depth, false};
    const int rc = advance_slab_stages(h, integ, terms, nterms, phi, buf1, buf2, tc, dt, hook, user);
    note.ok = rc == LSM_OK;
    return rc;
}
static int advance_slab_stages(LsmHandle* h, int integ, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2, double tc,
                               double dt, LsmStageHook hook, void* user) {
    LSM_TRY(run_hook(h, hook, user, 0, phi, tc));
    if (integ == 0) {           // ForwardEuler — src/timestepping.jl:128-137
        LSM_TRY(stage_slab(h, terms, nterms, phi, nullptr, buf1, nullptr, LSM_BASE_PSI, dt, 0.0, tc));
        LSM_HIP(h, hipMemcpyAsync(phi, buf1, esize(h) * (size_t)h->lay.total, hipMemcpyDeviceToDevice, h->stream));   // copy!(ϕ, dst): ghosts travel with the padded buffer
        return LSM_OK;
    }
    if (integ == 1) {           // RK2 — src/timestepping.jl:143-164
        LSM_TRY(stage_slab(h, terms, nterms, phi, nullptr, buf1, buf2, LSM_BASE_PSI, dt, 0.5 * dt, tc));
        LSM_TRY(run_hook(h, hook, user, 1, buf1, tc + dt));
        return stage_slab(h, terms, nterms, buf1, buf2, phi, nullptr, LSM_BASE_OTHER, 0.5 * dt, 0.0, tc + dt);
    }
    // RK3 — src/timestepping.jl:170-202
    LSM_TRY(stage_slab(h, terms, nterms, phi, nullptr, buf1, nullptr, LSM_BASE_PSI, dt, 0.0, tc));
    LSM_TRY(run_hook(h, hook, user, 1, buf1, tc + dt));
    LSM_TRY(stage_slab(h, terms, nterms, buf1, phi, buf2, nullptr, LSM_BASE_RK3_S2, 0.25 * dt, 0.0, tc + dt));
    LSM_TRY(run_hook(h, hook, user, 2, buf2, tc + 0.5 * dt));
    return stage_slab(h, terms, nterms, buf2, phi, phi, nullptr, LSM_BASE_RK3_S3, (2.0 / 3) * dt, 0.0, tc + 0.5 * dt);
}

// _advance!(::ForwardEuler) — src/timestepping.jl:128-137
int lsm_advance_fe(LsmHandle* h, const LsmTerm* terms, int nterms, void* phi, void* buf1, double tc, double dt,
                   LsmStageHook hook, void* user) {
    if (!h || !phi || !buf1) return LSM_ERR_INVALID;
    if (is_slab(h)) return advance_slab(h, 0, terms, nterms, phi, buf1, nullptr, tc, dt, hook, user);
    LSM_TRY(check_single_device(h));
    const XRedirect xr(h, hook, terms, nterms);
    LSM_TRY(xr.fill(phi));
    LSM_TRY(run_hook(h, hook, user, 0, phi, tc));
    LSM_TRY(lsm_stage(h, terms, nterms, phi, nullptr, buf1, nullptr, LSM_BASE_PSI, dt, 0.0, tc, nullptr));
    LSM_HIP(h, hipMemcpyAsync(phi, buf1, esize(h) * (size_t)h->lay.total, hipMemcpyDeviceToDevice, h->stream));   // copy!(ϕ, dst)
    return LSM_OK;     // ϕ's ghost layers are stale now: whoever reads them next fills them (as this function did on entry)
}

// _advance!(::RK2) — src/timestepping.jl:143-164 (pred = buf1, corr = buf2; the final copy!(ϕ, corr)
// is folded into the corrector, which writes ϕ directly)
int lsm_advance_rk2(LsmHandle* h, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2, double tc, double dt,
                    LsmStageHook hook, void* user) {
    if (!h || !phi || !buf1 || !buf2) return LSM_ERR_INVALID;
    if (is_slab(h)) return advance_slab(h, 1, terms, nterms, phi, buf1, buf2, tc, dt, hook, user);
    LSM_TRY(check_single_device(h));
    const XRedirect xr(h, hook, terms, nterms);
    LSM_TRY(xr.fill(phi));
    LSM_TRY(run_hook(h, hook, user, 0, phi, tc));
    LSM_TRY(lsm_stage(h, terms, nterms, phi, nullptr, buf1, buf2, LSM_BASE_PSI, dt, 0.5 * dt, tc, nullptr));
    LSM_TRY(xr.fill(buf1));
    LSM_TRY(run_hook(h, hook, user, 1, buf1, tc + dt));
    return lsm_stage(h, terms, nterms, buf1, buf2, phi, nullptr, LSM_BASE_OTHER, 0.5 * dt, 0.0, tc + dt, nullptr);
}

// _advance!(::RK3) — src/timestepping.jl:170-202 (stage 3 writes ϕ in place: it reads ϕ only pointwise)
int lsm_advance_rk3(LsmHandle* h, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2, double tc, double dt,
                    LsmStageHook hook, void* user) {
    if (!h || !phi || !buf1 || !buf2) return LSM_ERR_INVALID;
    if (is_slab(h)) return advance_slab(h, 2, terms, nterms, phi, buf1, buf2, tc, dt, hook, user);
    LSM_TRY(check_single_device(h));
    const XRedirect xr(h, hook, terms, nterms);
    LSM_TRY(xr.fill(phi));
    LSM_TRY(run_hook(h, hook, user, 0, phi, tc));
    LSM_TRY(lsm_stage(h, terms, nterms, phi, nullptr, buf1, nullptr, LSM_BASE_PSI, dt, 0.0, tc, nullptr));
    LSM_TRY(xr.fill(buf1));
    LSM_TRY(run_hook(h, hook, user, 1, buf1, tc + dt));
    LSM_TRY(lsm_stage(h, terms, nterms, buf1, phi, buf2, nullptr, LSM_BASE_RK3_S2, 0.25 * dt, 0.0, tc + dt, nullptr));
    LSM_TRY(xr.fill(buf2));
    LSM_TRY(run_hook(h, hook, user, 2, buf2, tc + 0.5 * dt));
    return lsm_stage(h, terms, nterms, buf2, phi, phi, nullptr, LSM_BASE_RK3_S3, (2.0 / 3) * dt, 0.0, tc + 0.5 * dt, nullptr);
}

int lsm_eikonal_sign(LsmHandle* h, const void* phi0, void* s0_out, void* stream) {
    if (!h || !phi0 || !s0_out) return LSM_ERR_INVALID;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    launch_eikonal_sign(h->grid.ndim, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->dxmin,
                        phi0, is_f32(h), (double*)s0_out, s);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

int lsm_extrema(LsmHandle* h, const void* phi, double* vmin, double* vmax) {
    if (!h || !phi || !vmin || !vmax) return LSM_ERR_INVALID;
    int nb = cfl_blocks(h->grid.ndim, h->nloc);
    if (nb > MAXB) nb = MAXB;
    launch_extrema(h->grid.ndim, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, phi, is_f32(h), h->d_partial,
                   h->d_partial + MAXB, nb, h->d_result, h->stream);
    LSM_HIP(h, hipMemcpyAsync(h->h_result, h->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LSM_SYNC(h);
    *vmin = h->h_result[0];
    *vmax = h->h_result[1];
    return LSM_OK;
}

#ifdef LSM_STAMP
// Diagnostic build only (`make stamp` -> variants/libhiplsm_stamp.so; not declared in include/lsm.h): the stage kernels
// stamp s_memtime / s_memrealtime around their plane loop.  enable = 1 arms the stamps, enable = 0 reads them back:
// *clock_ghz = median over the workgroups of Δs_memtime ÷ Δs_memrealtime × 0.1 (MI355X_MICROARCH.md, DVFS item 6).
int lsm_debug_stamp(LsmHandle* h, int enable, double* clock_ghz, double* loop_us) {
    if (!h) return LSM_ERR_INVALID;
    LSM_HIP(h, hipSetDevice(h->device));
    if (enable) {
        if (!h->d_stamp) LSM_HIP(h, hipMalloc((void**)&h->d_stamp, sizeof(unsigned long long) * 4 * 16384));
        LSM_HIP(h, hipMemsetAsync(h->d_stamp, 0, sizeof(unsigned long long) * 4 * 16384, h->stream));
        return LSM_OK;
    }
    if (!h->d_stamp || !clock_ghz) return LSM_ERR_INVALID;
    std::vector<unsigned long long> v(4 * 16384);
    LSM_SYNC(h);
    LSM_HIP(h, hipMemcpy(v.data(), h->d_stamp, sizeof(unsigned long long) * v.size(), hipMemcpyDeviceToHost));
    std::vector<double> clk, us;
    for (int i = 0; i < 16384; ++i)
        if (v[4 * i + 3]) { const double rt = (double)(v[4 * i + 3] - v[4 * i + 2]); clk.push_back((double)v[4 * i] / rt * 0.1); us.push_back(rt * 0.01); }
    if (clk.empty()) return fail(h, LSM_ERR_INVALID, "lsm_debug_stamp: no stamps (no stage kernel ran since the stamps were armed)");
    std::sort(clk.begin(), clk.end());
    std::sort(us.begin(), us.end());
    *clock_ghz = clk[clk.size() / 2];
    if (loop_us) *loop_us = us[us.size() / 2];
    return LSM_OK;
}
// the raw stamps of the last launch: 16384 x {Δs_memtime over the plane loop, s_memrealtime at kernel entry, at loop start, at loop end}
int lsm_debug_stamp_raw(LsmHandle* h, unsigned long long* out) {
    if (!h || !h->d_stamp || !out) return LSM_ERR_INVALID;
    LSM_HIP(h, hipSetDevice(h->device));
    LSM_SYNC(h);
    LSM_HIP(h, hipMemcpy(out, h->d_stamp, sizeof(unsigned long long) * 4 * 16384, hipMemcpyDeviceToHost));
    return LSM_OK;
}
#endif

int lsm_check_range(LsmHandle* h, const void* phi, int* ok, double* max_abs) {
    if (!h || !phi || !ok) return LSM_ERR_INVALID;
    double lo = 0.0, hi = 0.0;
    LSM_TRY(lsm_extrema(h, phi, &lo, &hi));     // the reduction skips NaN entries unless every entry is NaN
    double m = fabs(lo) > fabs(hi) ? fabs(lo) : fabs(hi);
    if (lo > hi) m = 0.0;                       // every entry NaN: nothing to bound
    if (max_abs) *max_abs = m;
    *ok = h->mode == LSM_MODE_STRICT || m <= LSM_FAST_MAX_ABS;
    return LSM_OK;
}

// volume / perimeter of the LOCAL slab (src/levelsetops.jl:27-33,139-149); mode 0 / 1
static int measure(LsmHandle* h, int mode, void* phi, double* out) {
    if (!h || !phi || !out) return LSM_ERR_INVALID;
    const int N = h->grid.ndim;
    if (mode == 1) { int r = lsm_fill_ghosts(h, phi, 7, nullptr); if (r) return r; }   // D⁰ reaches one ghost layer
    int nb = cfl_blocks(N, h->nloc);
    if (nb > MAXB) nb = MAXB;
    double scale = 1.0;
    for (int d = 0; d < N; ++d) scale = d == 0 ? h->h[0] : scale * h->h[d];            // prod(δ)
    launch_measure(mode, N, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->h, h->dxmin, scale,
                   phi, is_f32(h), h->d_partial, nb, h->d_result, h->stream);
    LSM_HIP(h, hipMemcpyAsync(h->h_result, h->d_result, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LSM_SYNC(h);
    *out = h->h_result[0];
    return LSM_OK;
}
int lsm_volume(LsmHandle* h, const void* phi, double* out) { return measure(h, 0, (void*)phi, out); }
int lsm_perimeter(LsmHandle* h, void* phi, double* out) { return measure(h, 1, phi, out); }

// volume / perimeter of a NarrowBandMeshField (src/levelsetops.jl:34-116,150-166)
int lsm_band_volume(LsmHandle* h, const void* phi, const void* mask, double* out) {
    if (!h || !phi || !mask || !out) return h ? fail(h, LSM_ERR_INVALID, "lsm_band_volume: null argument") : LSM_ERR_INVALID;
    LSM_TRY(check_single_device(h));
    const int N = h->grid.ndim;
    double scale = 1.0;
    for (int d = 0; d < N; ++d) scale = d == 0 ? h->h[0] : scale * h->h[d];
    if (launch_band_volume(N, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->dxmin, scale, phi, is_f32(h),
                           (const unsigned char*)mask, h->d_result, h->stream))
        return fail(h, LSM_ERR_HIP, "lsm_band_volume: device error");
    LSM_HIP(h, hipMemcpyAsync(h->h_result, h->d_result, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LSM_SYNC(h);
    *out = h->h_result[0];
    return LSM_OK;
}
int lsm_band_perimeter(LsmHandle* h, const void* phi, const void* mask, double* out) {
    if (!h || !phi || !mask || !out) return h ? fail(h, LSM_ERR_INVALID, "lsm_band_perimeter: null argument") : LSM_ERR_INVALID;
    LSM_TRY(check_single_device(h));
    const int N = h->grid.ndim;
    int nb = cfl_blocks(N, h->nloc);
    if (nb > MAXB) nb = MAXB;
    double scale = 1.0;
    for (int d = 0; d < N; ++d) scale = d == 0 ? h->h[0] : scale * h->h[d];
    launch_measure(1, N, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->h, h->dxmin, scale, phi, is_f32(h), h->d_partial, nb,
                   h->d_result, h->stream, (const unsigned char*)mask);
    LSM_HIP(h, hipMemcpyAsync(h->h_result, h->d_result, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LSM_SYNC(h);
    *out = h->h_result[0];
    return LSM_OK;
}

// curvature / gradient / normal fields (src/levelsetops.jl:197-226)
int lsm_geometry(LsmHandle* h, int what, void* phi, double scale, double band_width, double fill, void* out0, void* out1, void* out2,
                 void* frozen_out, void* stream) {
    if (!h || !phi || !out0) return h ? fail(h, LSM_ERR_INVALID, "lsm_geometry: null argument") : LSM_ERR_INVALID;
    if (what < LSM_GEOM_CURVATURE || what > LSM_GEOM_NORMAL) return fail(h, LSM_ERR_INVALID, "lsm_geometry: bad selector");
    const int N = h->grid.ndim;
    if (what != LSM_GEOM_CURVATURE && ((N > 1 && !out1) || (N > 2 && !out2)))
        return fail(h, LSM_ERR_INVALID, "lsm_geometry: one output array per dimension is required");
    if (phi == out0 || phi == out1 || phi == out2 || phi == frozen_out) return fail(h, LSM_ERR_INVALID, "lsm_geometry: outputs must not alias phi");
    LSM_TRY(check_single_device(h));     // the planes of a slab interface would have to be exchanged first
    LSM_TRY(lsm_fill_ghosts(h, phi, 7, stream));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    launch_geometry(what, N, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->h, scale, band_width, fill, phi, is_f32(h),
                    (double*)out0, (double*)out1, (double*)out2, (double*)frozen_out, s);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

// the same at the active nodes of a NarrowBandMeshField; phi must be a prepared stage input (lsm_band_prepare)
int lsm_band_geometry(LsmHandle* h, int what, const void* phi, const void* mask, double scale, double band_width, double fill, void* out0,
                      void* out1, void* out2, void* frozen_out, void* stream) {
    if (!h || !phi || !mask || !out0) return h ? fail(h, LSM_ERR_INVALID, "lsm_band_geometry: null argument") : LSM_ERR_INVALID;
    if (what < LSM_GEOM_CURVATURE || what > LSM_GEOM_NORMAL) return fail(h, LSM_ERR_INVALID, "lsm_band_geometry: bad selector");
    const int N = h->grid.ndim;
    if (what != LSM_GEOM_CURVATURE && ((N > 1 && !out1) || (N > 2 && !out2)))
        return fail(h, LSM_ERR_INVALID, "lsm_band_geometry: one output array per dimension is required");
    if (phi == out0 || phi == out1 || phi == out2 || phi == frozen_out) return fail(h, LSM_ERR_INVALID, "lsm_band_geometry: outputs must not alias phi");
    LSM_TRY(check_single_device(h));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    launch_geometry(what, N, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->h, scale, band_width, fill, phi, is_f32(h),
                    (double*)out0, (double*)out1, (double*)out2, (double*)frozen_out, s, (const unsigned char*)mask);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

// extend_along_normals! (src/velocityextension.jl:20-67): nb_iters first-order upwind pseudo-time
// sweeps F <- F - τ Σ_d a_d (a_d>0 ? D⁻F : D⁺F) with a = sign-weighted unit normal of ϕ, frozen nodes
// held fixed.  Each sweep is the fused stage kernel with an Upwind advection term whose velocity is
// the (frozen-masked) normal field.
int lsm_extend_along_normals(LsmHandle* h, void* F, void* phi, const void* frozen, void* work0, void* work1, void* work2,
                             void* work3, int nb_iters, double cfl, double interface_band, double min_norm) {
    if (!h || !F || !phi || !work0 || !work1) return h ? fail(h, LSM_ERR_INVALID, "lsm_extend_along_normals: null argument") : LSM_ERR_INVALID;
    const int N = h->grid.ndim;
    if ((N > 1 && !work2) || (N > 2 && !work3)) return fail(h, LSM_ERR_INVALID, "lsm_extend_along_normals: need ndim+1 work buffers");
    if (nb_iters < 0) return fail(h, LSM_ERR_INVALID, "nb_iters must be non-negative");
    if (!(cfl > 0)) return fail(h, LSM_ERR_INVALID, "cfl must be strictly positive");
    if (!(interface_band >= 0)) return fail(h, LSM_ERR_INVALID, "interface_band must be non-negative");
    if (!(min_norm >= 0)) return fail(h, LSM_ERR_INVALID, "min_norm must be non-negative");
    LSM_TRY(check_single_device(h));
    const double delta = h->dxmin;
    const double tau = cfl * delta;
    LSM_TRY(lsm_fill_ghosts(h, phi, 7, nullptr));
    void* comp[3] = {work1, work2, work3};
    launch_signed_normals(N, h->nloc, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->h, delta, interface_band * delta,
                          min_norm * min_norm, phi, is_f32(h), (const double*)frozen, (double*)comp[0], (double*)comp[1],
                          (double*)comp[2], h->stream);
    LSM_HIP(h, hipGetLastError());
    LsmTerm term;
    memset(&term, 0, sizeof(term));
    term.kind = LSM_TERM_ADVECTION;
    term.scheme = LSM_SCHEME_UPWIND;
    term.coeff.kind = LSM_COEFF_FIELD;
    for (int d = 0; d < N; ++d) term.coeff.field[d] = comp[d];
    void* cur = F;
    void* nxt = work0;
    for (int it = 0; it < nb_iters; ++it) {
        LSM_TRY(lsm_fill_ghosts(h, cur, 7, nullptr));
        LSM_TRY(lsm_stage(h, &term, 1, cur, nullptr, nxt, nullptr, LSM_BASE_PSI, tau, 0.0, 0.0, nullptr));
        void* t = cur; cur = nxt; nxt = t;
    }
    if (cur != F) LSM_HIP(h, hipMemcpyAsync(F, cur, esize(h) * (size_t)h->lay.total, hipMemcpyDeviceToDevice, h->stream));
    return LSM_OK;
}

// ------------------------------------------------------------------------------------------------
// narrow band (NarrowBandMeshField, src/meshfield.jl:314-588) — see lsm_band.hip
// ------------------------------------------------------------------------------------------------
static const int BAND_SEARCH_RADIUS = 6;   // src/meshfield.jl:513

static BandArgs band_args(const LsmHandle* h, int mc, const unsigned char* work) {
    BandArgs a;
    a.ndim = h->grid.ndim;
    for (int e = 0; e < 3; ++e) a.n[e] = h->nloc[e];
    a.s1 = h->lay.stride[1]; a.s2 = h->lay.stride[2]; a.origin = h->lay.origin;
    stage_tile_shape(a.ndim, &a.tx, &a.ty);
    a.tm = mc;
    a.nbx = (h->nloc[0] + a.tx - 1) / a.tx;
    a.nby = a.ndim == 3 ? (h->nloc[1] + a.ty - 1) / a.ty : 1;
    a.nbm = a.ndim >= 2 ? (h->nloc[a.ndim - 1] + mc - 1) / mc : 1;
    a.work = work;
    a.list = nullptr; a.nlist = 0;
    a.f32 = is_f32(h);
    a.force_bytes = h->band_bytes ? 1 : 0;
    return a;
}

// _ring_offsets (src/meshfield.jl:515-520): all offsets of the (2R+1)^N box in column-major order,
// stably sorted by squared length
static int ensure_ring(LsmHandle* h) {
    if (h->d_ring) return LSM_OK;
    const int N = h->grid.ndim, R = BAND_SEARCH_RADIUS;
    std::vector<std::array<signed char, 3>> off;
    for (int z = (N > 2 ? -R : 0); z <= (N > 2 ? R : 0); ++z)
        for (int y = (N > 1 ? -R : 0); y <= (N > 1 ? R : 0); ++y)
            for (int x = -R; x <= R; ++x) off.push_back({(signed char)x, (signed char)y, (signed char)z});
    std::stable_sort(off.begin(), off.end(), [](const std::array<signed char, 3>& p, const std::array<signed char, 3>& q) {
        return p[0] * p[0] + p[1] * p[1] + p[2] * p[2] < q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    });
    h->nring = (int)off.size();
    h->nring_lds = h->nring;
    for (int r = h->nring - 1; r >= 0; --r)
        if (std::abs((int)off[r][0]) > 3 || std::abs((int)off[r][1]) > 3 || std::abs((int)off[r][2]) > 3) h->nring_lds = r;
    LSM_HIP(h, hipMalloc((void**)&h->d_ring, off.size() * 3));
    LSM_HIP(h, hipMemcpy(h->d_ring, off.data(), off.size() * 3, hipMemcpyHostToDevice));
    LSM_HIP(h, hipMalloc((void**)&h->d_miss, sizeof(int)));
    LSM_HIP(h, hipMalloc((void**)&h->d_count, sizeof(unsigned long long)));
    LSM_HIP(h, hipMemset(h->d_miss, 0, sizeof(int)));
    return LSM_OK;
}

// handle-owned scratch for the per-tile work flags
static int ensure_work(LsmHandle* h, int64_t ntiles) {
    if (h->work_cap >= ntiles) return LSM_OK;
    if (h->d_work) { (void)hipFree(h->d_work); (void)hipFree(h->d_act_list); (void)hipFree(h->d_work_list); (void)hipFree(h->d_lcounts); (void)hipFree(h->d_tiles_old); }
    LSM_HIP(h, hipMalloc((void**)&h->d_work, (size_t)ntiles));
    LSM_HIP(h, hipMalloc((void**)&h->d_tiles_old, (size_t)ntiles));
    LSM_HIP(h, hipMalloc((void**)&h->d_act_list, (size_t)ntiles * sizeof(int)));
    LSM_HIP(h, hipMalloc((void**)&h->d_work_list, (size_t)ntiles * sizeof(int)));
    LSM_HIP(h, hipMalloc((void**)&h->d_lcounts, 4 * sizeof(unsigned)));
    h->work_cap = ntiles;
    h->lists_tiles = nullptr; h->lists_host_valid = false;
    return LSM_OK;
}

// are the compact tile lists (and their lengths on the host) those of this tile-flag buffer?
static bool have_lists(const LsmHandle* h, const void* tiles, int mc) {
    return !h->no_lists && h->lists_host_valid && h->lists_tiles == tiles && h->lists_mc == mc;
}

int lsm_band_tile_count(LsmHandle* h, int mc, int64_t* ntiles) {
    if (!h || !ntiles || mc < 1) return LSM_ERR_INVALID;
    const BandArgs a = band_args(h, mc, nullptr);
    *ntiles = (int64_t)a.nbx * a.nby * a.nbm;
    return LSM_OK;
}

// update_band! (src/meshfield.jl:555-588) and everything derived from the new band, in one call:
//   mask      in: old band (ignored when from_dense), out: new band
//   tiles     in: active tiles of the old band (ignored when from_dense), out: active tiles of the new band
//   halo_mask out: nodes within Chebyshev distance 3 of the new band
// Newly active nodes receive the affine extrapolant from the OLD band.  scratch_a/b: mask-sized buffers.
// `visit`: the tiles to cover (from lsm_band_update); NULL = the band's own work tiles
// `interior`: 1 = no stencil of the band reaches outside the grid (known by the caller), 0 = unknown / it may, -1 = ask the lists
static int band_halo_impl(LsmHandle* h, const void* vals, const void* mask, void* halo_mask, const void* tiles, int mc, void* halo_list,
                          int64_t halo_cap, void* halo_count, const BandArgs* visit, int interior_hint, bool halo_cleared = false) {
    if (!h || !vals || !mask || !halo_mask || !tiles || mc < 1) return LSM_ERR_INVALID;
    LSM_TRY(ensure_ring(h));
    BandArgs a = band_args(h, mc, nullptr);
    LSM_TRY(ensure_work(h, (int64_t)a.nbx * a.nby * a.nbm));
    if (visit) {
        a = *visit;
    } else if (have_lists(h, tiles, mc)) {
        a.list = h->d_work_list; a.nlist = h->nwork;
    } else {
        launch_band_work(a, (const unsigned char*)tiles, h->d_work, nullptr, nullptr, h->stream);
        a.work = h->d_work;
    }
    const int N = h->grid.ndim;
    // nothing reads the halo mask outside the tiles visited here: clear those (0.5 % of a 768³ grid) instead of every byte
    if (halo_cleared && visit) {}                                   // lsm_band_update's copy pass cleared the visited tiles
    else if (a.list || a.work) launch_band_zero(a, (unsigned char*)halo_mask, h->stream);
    else LSM_HIP(h, hipMemsetAsync(halo_mask, 0, (size_t)h->lay.total, h->stream));
    BandBcArgs bc;
    for (int d = 0; d < 3; ++d)
        for (int sd = 0; sd < 2; ++sd) { bc.kind[d][sd] = d < N ? h->bc[d][sd].kind : LSM_BC_NONE; bc.degree[d][sd] = d < N ? h->bc[d][sd].degree : 0; }
    // no work tile of the band these lists describe touches a face of the grid: no stencil reaches outside it
    const bool interior = interior_hint < 0 ? (have_lists(h, tiles, mc) && h->nface == 0) : interior_hint != 0;
    if (!interior)
        for (int d = N - 1; d >= 0; --d)
            launch_band_halo_bc(a, bc, d, LSM_GHOST, (const unsigned char*)mask, (unsigned char*)halo_mask, h->stream);
    if (halo_count) LSM_HIP(h, hipMemsetAsync(halo_count, 0, sizeof(unsigned), h->stream));
    h->halo_n_key = nullptr;
    launch_band_extrapolate(a, nullptr, (unsigned char*)halo_mask, (const unsigned char*)mask, h->d_ring, h->nring, h->nring_lds,
                            vals, nullptr, h->d_miss, halo_count ? (BandEntry*)halo_list : nullptr,
                            (unsigned*)halo_count, halo_list && halo_count ? (unsigned)halo_cap : 0u, h->stream);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

// Δt of the next step over the band just built (see LsmHandle::BandCfl): the reductions of the armed term list run over
// the new compact tile list, whose length is still on the device; lsm_band_status reads the results back with its own numbers.
static bool band_cfl_node_dependent(const LsmTerm& tm) {
    return tm.kind != LSM_TERM_EIKONAL && tm.coeff.kind != LSM_COEFF_CONST;
}
static const int PF_PARTIALS = 2048;       // workgroups (partials) per prefetched reduction: 4 slots in the 2·MAXB doubles of d_partial
static int band_cfl_prefetch(LsmHandle* h, const void* mask, const void* tiles, int mc, const unsigned* rowbits = nullptr) {
    hipStream_t s = h->stream;
    LsmHandle::BandCfl& pf = h->band_cfl;
    pf.pending = false; pf.valid = false;
    const bool off = !h->tune.band_cfl_prefetch;
    if (off || !pf.armed || pf.mask != mask || pf.tiles != tiles || pf.mc != mc || h->grid.ndim != 3 || h->no_lists) return LSM_OK;
    const BandArgs ba = band_args(h, mc, nullptr);
    for (int k = 0; k < pf.nterms; ++k) {
        const int sl = pf.slot[k];
        if (sl < 0) continue;
        CflArgs a;
        cfl_args(h, pf.terms[k], 0.0, a);              // armed terms carry no time factor
        a.partial = h->d_partial + PF_PARTIALS * sl;
        a.nanflag = h->d_pf_flag + sl;
        a.mask = (const unsigned char*)mask;
        a.tile_active = (const unsigned char*)tiles;
        a.rowbits = rowbits;                            // the new band's row words, still in the update's scratch (NULL: the byte mask)
        a.tx = ba.tx; a.ty = ba.ty; a.tm = ba.tm; a.nbx = ba.nbx; a.nby = ba.nby;
        if (launch_cfl_band_list(a, h->d_act_list, 0, h->d_lcounts, PF_PARTIALS, s) != PF_PARTIALS) return LSM_OK;   // tile shape not served: no prefetch
        // (the second stage of the reduction is lsm_band_status's kernel)
    }
    LSM_HIP(h, hipGetLastError());
    pf.pending = true;
    return LSM_OK;
}

int lsm_band_update(LsmHandle* h, void* vals, void* mask, int from_dense, int nlayers, void* scratch_a, void* scratch_b,
                    void* halo_mask, void* tiles, int mc, void* halo_list, int64_t halo_cap, void* halo_count) {
    if (!h || !vals || !mask || !scratch_a || !scratch_b || !halo_mask || !tiles)
        return h ? fail(h, LSM_ERR_INVALID, "lsm_band_update: null argument") : LSM_ERR_INVALID;
    if (nlayers < 0 || mc < 1) return fail(h, LSM_ERR_INVALID, "lsm_band_update: nlayers must be >= 0 and mc >= 1");
    h->band_cfl.valid = false; h->band_cfl.pending = false;      // the band changes: a prefetched Δt is void
    const int N = h->grid.ndim;
    for (int d = 0; d < N; ++d)
        for (int sd = 0; sd < 2; ++sd)
            if (h->bc[d][sd].kind == LSM_BC_PERIODIC)   // src/meshfield.jl:339-340
                return fail(h, LSM_ERR_INVALID, "PeriodicBC is not supported on a NarrowBandMeshField");
    LSM_TRY(ensure_ring(h));
    BandArgs a = band_args(h, mc, nullptr);
    const int64_t ntiles = (int64_t)a.nbx * a.nby * a.nbm;
    LSM_TRY(ensure_work(h, ntiles));
    // The new band grows out of cut cells of the OLD band, so it lies within nlayers nodes of it, its halo within
    // nlayers + 3 and the boundary-condition sources of its stencils within nlayers + P: all inside the old band's
    // tiles and their neighbours ("work tiles") as long as a tile is at least that thick (8 nodes in its thinnest
    // direction); wider bands fall back to visiting every tile.
    int tmin = a.tx, reach = LSM_GHOST;
    if (N == 3) tmin = a.ty < tmin ? a.ty : tmin;
    if (N >= 2) tmin = a.tm < tmin ? a.tm : tmin;
    for (int d = 0; d < N; ++d)
        for (int sd = 0; sd < 2; ++sd)
            if (h->bc[d][sd].kind == LSM_BC_EXTRAPOLATION && h->bc[d][sd].degree > reach) reach = h->bc[d][sd].degree;
    const bool local = !from_dense && nlayers + reach <= tmin;
    const bool listed = local && have_lists(h, tiles, mc);   // the work tiles as a compact list: one block per work tile
    if (listed) {
        a.list = h->d_work_list; a.nlist = h->nwork;
    } else if (local) {
        launch_band_work(a, (const unsigned char*)tiles, h->d_work, nullptr, nullptr, h->stream);
        a.work = h->d_work;
    }
    const size_t bytes = (size_t)h->lay.total;
    // Bit-row path (lsm_band.hip): every node read once, the mask updated in place, new nodes extrapolated by the grow kernel itself
    const bool bits_env = h->tune.band_bits != 0;
    const size_t words = (size_t)ntiles * (size_t)(a.ty * a.tm);                 // u32 words per bit array
    if (bits_env && local && nlayers <= 3 && band_bits_fit(a, nlayers) && 2 * words * sizeof(unsigned) <= bytes &&
        ((uintptr_t)scratch_a & 7) == 0 && ((uintptr_t)scratch_b & 7) == 0) {
        unsigned *OB = (unsigned*)scratch_a, *LE = OB + words, *GE = (unsigned*)scratch_b, *NB = GE + words;
        BandArgs act = band_args(h, mc, nullptr);
        if (listed) { act.list = h->d_act_list; act.nlist = h->nact; }
        else act.work = (const unsigned char*)tiles;           // unchanged until the grow kernel
        const bool interior_b = listed && h->nface == 0;     // the new band lies in the old work tiles: none on a face
        const bool halo_bits = interior_b && halo_list && halo_count;
        launch_band_bits(act, vals, (const unsigned char*)mask, OB, LE, GE, (const unsigned char*)tiles, h->d_tiles_old,
                         halo_bits ? (unsigned*)halo_count : nullptr, h->stream);
        launch_band_grow_bits(a, vals, (unsigned char*)mask, nlayers, h->d_tiles_old, (unsigned char*)tiles, OB, LE, GE, NB, h->d_miss, h->stream);
        LSM_HIP(h, hipGetLastError());
        h->lists_host_valid = false;                          // from here on the lists describe the previous band
        // (Measured in round 4 and not kept: the chain below — four small latency-bound launches — on a stream of its own beside the
        // halo search, with the old work list kept aside: 0.533 against 0.524 ms per 768³ step.  The 1024-thread workgroups of the
        // lists kernel find no room while the search fills every CU, and the fork / join events cost what the overlap gives.)
        if (halo_bits) {
            h->halo_n_key = nullptr;                          // (the counter was cleared by band_bits_kernel)
            launch_band_halo_bits(a, (const unsigned char*)tiles, NB, (unsigned char*)halo_mask, h->d_miss, (BandEntry*)halo_list,
                                  (unsigned*)halo_count, (unsigned)halo_cap, h->stream);
            LSM_HIP(h, hipGetLastError());
        } else {
            LSM_TRY(band_halo_impl(h, vals, mask, halo_mask, tiles, mc, halo_list, halo_cap, halo_count, listed ? &a : nullptr, interior_b ? 1 : 0));
        }
        BandArgs full = band_args(h, mc, nullptr);
        launch_band_work(full, (const unsigned char*)tiles, h->d_work, h->d_lcounts + 2, h->d_pf_flag, h->stream);
        launch_band_lists(full, (const unsigned char*)tiles, h->d_work, h->d_act_list, h->d_work_list, h->d_lcounts, h->stream);
        h->lists_tiles = tiles; h->lists_mc = mc; h->lists_host_valid = false;
        LSM_HIP(h, hipGetLastError());
        return band_cfl_prefetch(h, mask, tiles, mc, NB);
    }
    unsigned char *A = (unsigned char*)scratch_a, *B = (unsigned char*)scratch_b;
    const unsigned char* old_mask = from_dense ? nullptr : (const unsigned char*)mask;
    const bool fused = band_grow_fits(a, nlayers);
    if (fused) {
        // cut cells, seeds, dilations and the tile flags in one LDS-resident kernel; A is written on every visited tile
        launch_band_grow(a, vals, old_mask, nlayers, A, (unsigned char*)tiles, h->stream);
    } else {
        LSM_HIP(h, hipMemsetAsync(A, 0, bytes, h->stream));
        LSM_HIP(h, hipMemsetAsync(B, 0, bytes, h->stream));
        launch_band_cut(a, vals, old_mask, A, h->stream);
        for (int l = 0; l < nlayers; ++l) {
            launch_band_dilate(a, A, B, h->stream);
            unsigned char* t = A; A = B; B = t;
        }
    }
    if (!from_dense)
        launch_band_extrapolate(a, A, nullptr, (const unsigned char*)mask, h->d_ring, h->nring, h->nring_lds, vals,
                                vals, h->d_miss, nullptr, nullptr, 0, h->stream);
    const bool clear_with_copy = listed;                       // the halo pass below visits the same listed tiles
    launch_band_copy(a, A, (unsigned char*)mask, clear_with_copy ? (unsigned char*)halo_mask : nullptr, h->stream);   // the old band lies inside the visited tiles
    if (!fused) launch_band_tiles(a, (const unsigned char*)mask, (unsigned char*)tiles, h->stream);
    LSM_HIP(h, hipGetLastError());
    // halo of the new band: over the same (old) work tiles when they are listed, else over the new band's
    const int interior = listed && h->nface == 0 ? 1 : 0;   // the new band lies in the old work tiles: none on a face
    h->lists_host_valid = false;                             // from here on the lists describe the previous band
    LSM_TRY(band_halo_impl(h, vals, mask, halo_mask, tiles, mc, halo_list, halo_cap, halo_count, listed ? &a : nullptr, interior, clear_with_copy));
    // compact lists of the new band's tiles for the launches that follow lsm_band_status
    BandArgs full = band_args(h, mc, nullptr);
    launch_band_work(full, (const unsigned char*)tiles, h->d_work, h->d_lcounts + 2, h->d_pf_flag, h->stream);
    launch_band_lists(full, (const unsigned char*)tiles, h->d_work, h->d_act_list, h->d_work_list, h->d_lcounts, h->stream);
    h->lists_tiles = tiles; h->lists_mc = mc; h->lists_host_valid = false;
    LSM_HIP(h, hipGetLastError());
    return band_cfl_prefetch(h, mask, tiles, mc);
}

// halo_mask := the in-grid nodes stencils centred on band nodes read, directly (axis lines of length LSM_GHOST and
// the 3^N box) or through the boundary conditions; halo_list := for each of them outside the band, its nearest band
// node (a function of the mask alone: found once per band, applied to every stage input by lsm_band_fill_list).
// *halo_count may exceed halo_cap: the list is then truncated and the call must be repeated with a larger one.
int lsm_band_halo(LsmHandle* h, const void* vals, const void* mask, void* halo_mask, const void* tiles, int mc, void* halo_list,
                  int64_t halo_cap, void* halo_count) {
    return band_halo_impl(h, vals, mask, halo_mask, tiles, mc, halo_list, halo_cap, halo_count, nullptr, -1);
}

// tiles := per-tile activity flags of `mask`, and the compact tile lists of it — after the mask was changed from
// outside (a slab that received its overlap planes from the neighbouring ranks); follow with lsm_band_status and
// lsm_band_halo
int lsm_band_retile(LsmHandle* h, const void* mask, void* tiles, int mc) {
    if (!h || !mask || !tiles || mc < 1) return LSM_ERR_INVALID;
    h->band_cfl.valid = false; h->band_cfl.pending = false;      // the mask was changed from outside
    LSM_TRY(ensure_ring(h));
    BandArgs a = band_args(h, mc, nullptr);
    LSM_TRY(ensure_work(h, (int64_t)a.nbx * a.nby * a.nbm));
    launch_band_tiles(a, (const unsigned char*)mask, (unsigned char*)tiles, h->stream);
    launch_band_work(a, (const unsigned char*)tiles, h->d_work, h->d_lcounts + 2, h->d_pf_flag, h->stream);
    launch_band_lists(a, (const unsigned char*)tiles, h->d_work, h->d_act_list, h->d_work_list, h->d_lcounts, h->stream);
    h->lists_tiles = tiles; h->lists_mc = mc; h->lists_host_valid = false;
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

// What the handle remembers about caller-owned band buffers — the compact tile lists of `tiles`, the halo list's length as the
// host last read it, a prefetched Δt — is keyed by their ADDRESSES.  A caller that rewrites mask / tiles / halo_list /
// halo_count in place (copy! of another band into these buffers) calls this, or lsm_band_retile + lsm_band_status, which rebuild
// that state; until then the band kernels run over all tiles and read the list's length on the device.
int lsm_band_invalidate(LsmHandle* h) {
    if (!h) return LSM_ERR_INVALID;
    h->lists_tiles = nullptr; h->lists_host_valid = false;
    h->halo_n_key = nullptr; h->halo_n = 0;
    h->band_cfl.valid = false; h->band_cfl.pending = false;
    return LSM_OK;
}

// ϕ[I] for the non-band nodes flagged in `targets`: _extrapolate_to_ghost materialised
// (src/meshfield.jl:481-511) by a fresh nearest-node search over the whole grid (scalar getindex path).
int lsm_band_fill(LsmHandle* h, void* vals, const void* mask, const void* targets, const void* tiles, int mc) {
    if (!h || !vals || !mask || !targets || mc < 1) return LSM_ERR_INVALID;
    LSM_TRY(ensure_ring(h));
    BandArgs a = band_args(h, mc, nullptr);
    if (tiles && have_lists(h, tiles, mc)) {
        a.list = h->d_work_list; a.nlist = h->nwork;
    } else if (tiles) {
        LSM_TRY(ensure_work(h, (int64_t)a.nbx * a.nby * a.nbm));
        launch_band_work(a, (const unsigned char*)tiles, h->d_work, nullptr, nullptr, h->stream);
        a.work = h->d_work;
    }
    launch_band_extrapolate(a, (const unsigned char*)targets, nullptr, (const unsigned char*)mask, h->d_ring, h->nring, h->nring_lds,
                            vals, vals, h->d_miss, nullptr, nullptr, 0, h->stream);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

// The same for the halo of the band, from the (node, nearest band node) list lsm_band_update prepared:
// a gather per stage input.  Call lsm_fill_ghosts afterwards for the out-of-grid layers.
int lsm_band_fill_list(LsmHandle* h, void* vals, const void* mask, const void* halo_list, int64_t halo_cap, const void* halo_count) {
    if (!h || !vals || !mask || !halo_list || !halo_count) return LSM_ERR_INVALID;
    const long long n_host = (h->halo_n_key == halo_count && h->halo_n_key) ? h->halo_n : -1;
    launch_band_apply(band_args(h, 8, nullptr), (const BandEntry*)halo_list, (const unsigned*)halo_count, n_host, (unsigned)halo_cap,
                      (const unsigned char*)mask, vals, vals, h->stream);
    LSM_HIP(h, hipGetLastError());
    return LSM_OK;
}

// lsm_band_fill_list followed by lsm_fill_ghosts — what a stage input needs; the ghost fill is skipped while no
// work tile of the band touches a face of the grid (then no band stencil reads a ghost)
int lsm_band_prepare(LsmHandle* h, void* vals, const void* mask, const void* halo_list, int64_t halo_cap, const void* halo_count,
                     const void* tiles, int mc) {
    const int r = lsm_band_fill_list(h, vals, mask, halo_list, halo_cap, halo_count);
    if (r != LSM_OK) return r;
    if (tiles && have_lists(h, tiles, mc) && h->nface == 0) return LSM_OK;
    return lsm_fill_ghosts(h, vals, 7, nullptr);
}

int lsm_band_count(LsmHandle* h, const void* mask, int64_t* count) {
    if (!h || !mask || !count) return LSM_ERR_INVALID;
    LSM_TRY(ensure_ring(h));
    LSM_HIP(h, hipMemsetAsync(h->d_count, 0, sizeof(unsigned long long), h->stream));
    launch_band_count(band_args(h, 8, nullptr), (const unsigned char*)mask, h->d_count, h->stream);
    unsigned long long c = 0;
    LSM_HIP(h, hipMemcpyAsync(&c, h->d_count, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    LSM_SYNC(h);
    *count = (int64_t)c;
    return LSM_OK;
}

// a node asked for a value farther than the search radius from the band since the last call
// (the reference throws ArgumentError, src/meshfield.jl:499-500)
int lsm_band_missed(LsmHandle* h, int* missed) {
    if (!h || !missed) return LSM_ERR_INVALID;
    LSM_TRY(ensure_ring(h));
    LSM_HIP(h, hipMemcpyAsync(missed, h->d_miss, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LSM_SYNC(h);
    LSM_HIP(h, hipMemsetAsync(h->d_miss, 0, sizeof(int), h->stream));
    return LSM_OK;
}

// lsm_band_missed + the number of entries the last lsm_band_halo wanted to write, in one synchronisation
int lsm_band_status(LsmHandle* h, const void* halo_count, int64_t* count, int* missed) {
    if (!h || !halo_count || !count || !missed) return LSM_ERR_INVALID;
    LSM_TRY(ensure_ring(h));
    // one gather kernel and ONE copy into pinned memory instead of three small copies into pageable memory (a host round trip each)
    // the kernel writes into the host's pinned page itself ([2..7] status, [8..11] prefetched Δt): no copy, one synchronisation
    BandStatusCfl sc;
    memset(&sc, 0, sizeof(sc));
    if (h->band_cfl.pending) {
        sc.npartials = PF_PARTIALS; sc.partial = h->d_partial; sc.nanflag = h->d_pf_flag; sc.dxmin = h->dxmin;
        for (int k = 0; k < h->band_cfl.nterms; ++k)
            if (h->band_cfl.slot[k] >= 0) { sc.kind[h->band_cfl.slot[k]] = h->band_cfl.terms[k].kind; sc.n = h->band_cfl.slot[k] + 1 > sc.n ? h->band_cfl.slot[k] + 1 : sc.n; }
    }
    launch_band_status((const unsigned*)halo_count, h->d_miss, h->lists_tiles ? h->d_lcounts : nullptr, sc, h->h_result_dev + 2, (double)++h->status_ticket,
                       h->stream);
    LSM_HIP(h, hipGetLastError());
    // The kernel writes its numbers into the host's pinned page and its ticket last: spin on the ticket (the stream is in order: all
    // that was queued before has finished when it appears) — a stream synchronisation costs the step tens of microseconds of
    // wake-up during which the device idles.  Not seen within 20 ms (a long queue, a fault, a silent RCCL peer): the timed wait.
    {
        const volatile double* tk = h->h_result + 13;
        const double want = (double)h->status_ticket;
        const auto t0 = std::chrono::steady_clock::now();
        const bool no_spin = !h->tune.status_spin;
        bool seen = false;
        while (!no_spin && !(seen = (*tk == want))) {
            __builtin_ia32_pause();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (!seen) LSM_SYNC(h);
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (h->band_cfl.pending) {
        for (int k = 0; k < h->band_cfl.nterms; ++k)
            if (h->band_cfl.slot[k] >= 0) h->band_cfl.dt[k] = h->h_result[8 + h->band_cfl.slot[k]];
        h->band_cfl.pending = false;
        h->band_cfl.valid = true;
    }
    *missed = (int)h->h_result[3];
    if (h->lists_tiles) { h->nact = (unsigned)h->h_result[4]; h->nwork = (unsigned)h->h_result[5]; h->nface = (unsigned)h->h_result[6]; h->lists_host_valid = true; }
    if (*missed) LSM_HIP(h, hipMemsetAsync(h->d_miss, 0, sizeof(int), h->stream));
    *count = (int64_t)h->h_result[2];
    h->halo_n_key = halo_count; h->halo_n = *count;
    return LSM_OK;
}

// one stage on the band: tiles without band nodes are skipped, only band nodes are stored
int lsm_stage_band(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out, void* out2,
                   int base_mode, double cdt, double cdt2, double t_stage, const void* mask, const void* tiles, int mc,
                   void* stream) {
    if (!h || !mask) return LSM_ERR_INVALID;
    h->band_mask = (const unsigned char*)mask;
    h->band_tiles = (const unsigned char*)tiles;
    h->band_mc = tiles ? mc : 0;
    if (tiles && have_lists(h, tiles, mc)) { h->band_list = h->d_act_list; h->band_nlist = h->nact; }
    const int r = stage_impl(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t_stage, 0, h->nloc[h->grid.ndim - 1], stream);
    h->band_mask = nullptr; h->band_tiles = nullptr; h->band_mc = 0; h->band_list = nullptr; h->band_nlist = 0;
    return r;
}

int lsm_compute_cfl_band(LsmHandle* h, const LsmTerm* terms, int nterms, const void* phi, const void* mask, const void* tiles, int mc,
                         double t, double* dt_out) {
    if (!h || !mask || (tiles && mc < 1)) return LSM_ERR_INVALID;
    LsmHandle::BandCfl& pf = h->band_cfl;
    const bool same = terms && nterms >= 1 && nterms <= LSM_MAX_TERMS && pf.nterms == nterms && pf.mask == mask && pf.tiles == tiles && pf.mc == mc &&
                      memcmp(pf.terms, terms, sizeof(LsmTerm) * (size_t)nterms) == 0;
    h->band_mask = (const unsigned char*)mask;
    h->band_tiles = (const unsigned char*)tiles;
    h->band_mc = tiles ? mc : 0;
    const bool keep = h->cfl_cache_on;
    h->cfl_cache_on = false;            // the minimum runs over the current band only
    h->cfl_prefetched = same && pf.valid;      // Δt of the node-dependent terms came home with the last lsm_band_status
    const int r = lsm_compute_cfl(h, terms, nterms, phi, t, dt_out);
    h->cfl_prefetched = false;
    h->cfl_cache_on = keep;
    h->band_mask = nullptr; h->band_tiles = nullptr; h->band_mc = 0;
    if (r == LSM_OK && !same && terms && nterms >= 1 && nterms <= LSM_MAX_TERMS) {
        // arm the prefetch for the bands to come: every term independent of t and of device fields, at most 4 node-dependent ones
        pf.valid = false; pf.pending = false; pf.armed = tiles != nullptr && h->cfl_cache_on;
        memcpy(pf.terms, terms, sizeof(LsmTerm) * (size_t)nterms);
        pf.nterms = nterms; pf.mask = mask; pf.tiles = tiles; pf.mc = mc;
        int nslots = 0;
        for (int k = 0; k < nterms; ++k) {
            const LsmTerm& tm = terms[k];
            pf.slot[k] = -1;
            if (!band_cfl_node_dependent(tm)) continue;
            const bool timeless = tm.coeff.kind == LSM_COEFF_ROTATION || (tm.coeff.kind == LSM_COEFF_SEPARABLE && tm.coeff.time_kind == LSM_TIME_ONE);
            if (!timeless || nslots == 4) { pf.armed = false; break; }
            pf.slot[k] = nslots++;
        }
        if (nslots == 0) pf.armed = false;     // nothing to launch: the host evaluates every term
    }
    return r;
}

// _advance!(integrator, ϕ::NarrowBandMeshField, buffers, terms, tc, Δt) — src/timestepping.jl:128-137,143-164,170-202 with
// active_nodeindices = the band (src/meshfield.jl:330-333): the stages of the dense step restricted to the band nodes.  Every
// stage input is made readable first (lsm_band_prepare: off-band stencil nodes by affine extrapolation from the band,
// src/meshfield.jl:481-511, then the boundary-condition ghosts).  On a slab with lsm_band_overlap_config the overlap planes
// of every stage result are refreshed from their owners.  Off-band entries of phi / buf1 / buf2 are scratch.
static int band_check(LsmHandle* h, const LsmBand* b, const char* what) {
    if (!b || !b->mask || !b->tiles || !b->halo_list || !b->halo_count || b->mc < 1)
        return fail(h, LSM_ERR_INVALID, std::string(what) + ": incomplete LsmBand (mask, tiles, mc, halo_list, halo_count)");
    return LSM_OK;
}
static int band_stage(LsmHandle* h, const LsmBand* b, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out,
                      void* out2, int base_mode, double cdt, double cdt2, double t) {
    return lsm_stage_band(h, terms, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, b->mask, b->tiles, b->mc, nullptr);
}
static int band_ready(LsmHandle* h, const LsmBand* b, void* field, bool refresh) {
    if (refresh && lsm_comm_band_overlap(h) > 0) LSM_TRY(lsm_band_overlap_values(h, field));
    return lsm_band_prepare(h, field, b->mask, b->halo_list, b->halo_cap, b->halo_count, b->tiles, b->mc);
}
static int advance_band(LsmHandle* h, int integ, const LsmTerm* terms, int nterms, const LsmBand* b, void* phi, void* buf1, void* buf2,
                        double tc, double dt, LsmStageHook hook, void* user) {
    if (!h || !phi || !buf1 || (integ > 0 && !buf2)) return h ? fail(h, LSM_ERR_INVALID, "lsm_advance_band: null argument") : LSM_ERR_INVALID;
    LSM_TRY(band_check(h, b, "lsm_advance_band"));
    const bool slabbed = lsm_comm_band_overlap(h) > 0;
    if (!slabbed) LSM_TRY(check_single_device(h));
    LSM_TRY(band_ready(h, b, phi, false));
    LSM_TRY(run_hook(h, hook, user, 0, phi, tc));
    if (integ == 0) {           // ForwardEuler: dst = copy of ϕ, updated on the band, copied back (:128-137) — only band entries matter
        LSM_TRY(band_stage(h, b, terms, nterms, phi, nullptr, buf1, nullptr, LSM_BASE_PSI, dt, 0.0, tc));
        BandArgs a = band_args(h, b->mc, nullptr);
        if (have_lists(h, b->tiles, b->mc)) { a.list = h->d_act_list; a.nlist = h->nact; }
        else a.work = (const unsigned char*)b->tiles;
        launch_band_copy_values(a, (const unsigned char*)b->mask, buf1, phi, h->stream);
        LSM_HIP(h, hipGetLastError());
    } else if (integ == 1) {    // RK2 (:143-164)
        LSM_TRY(band_stage(h, b, terms, nterms, phi, nullptr, buf1, buf2, LSM_BASE_PSI, dt, 0.5 * dt, tc));
        LSM_TRY(band_ready(h, b, buf1, true));
        LSM_TRY(run_hook(h, hook, user, 1, buf1, tc + dt));
        LSM_TRY(band_stage(h, b, terms, nterms, buf1, buf2, phi, nullptr, LSM_BASE_OTHER, 0.5 * dt, 0.0, tc + dt));
    } else {                    // RK3 (:170-202)
        LSM_TRY(band_stage(h, b, terms, nterms, phi, nullptr, buf1, nullptr, LSM_BASE_PSI, dt, 0.0, tc));
        LSM_TRY(band_ready(h, b, buf1, true));
        LSM_TRY(run_hook(h, hook, user, 1, buf1, tc + dt));
        LSM_TRY(band_stage(h, b, terms, nterms, buf1, phi, buf2, nullptr, LSM_BASE_RK3_S2, 0.25 * dt, 0.0, tc + dt));
        LSM_TRY(band_ready(h, b, buf2, true));
        LSM_TRY(run_hook(h, hook, user, 2, buf2, tc + 0.5 * dt));
        LSM_TRY(band_stage(h, b, terms, nterms, buf2, phi, phi, nullptr, LSM_BASE_RK3_S3, (2.0 / 3) * dt, 0.0, tc + 0.5 * dt));
    }
    if (slabbed) LSM_TRY(lsm_band_overlap_values(h, phi));
    return LSM_OK;
}
int lsm_advance_band_fe(LsmHandle* h, const LsmTerm* terms, int nterms, const LsmBand* band, void* phi, void* buf1, double tc, double dt,
                        LsmStageHook hook, void* user) {
    return advance_band(h, 0, terms, nterms, band, phi, buf1, nullptr, tc, dt, hook, user);
}
int lsm_advance_band_rk2(LsmHandle* h, const LsmTerm* terms, int nterms, const LsmBand* band, void* phi, void* buf1, void* buf2, double tc,
                         double dt, LsmStageHook hook, void* user) {
    return advance_band(h, 1, terms, nterms, band, phi, buf1, buf2, tc, dt, hook, user);
}
int lsm_advance_band_rk3(LsmHandle* h, const LsmTerm* terms, int nterms, const LsmBand* band, void* phi, void* buf1, void* buf2, double tc,
                         double dt, LsmStageHook hook, void* user) {
    return advance_band(h, 2, terms, nterms, band, phi, buf1, buf2, tc, dt, hook, user);
}

// reinitialize!(ϕ; order, upsample, maxiters, xtol, ftol) — src/reinitializer.jl:12-42.  `phi` must be readable by
// stencils (ghosts filled; band halo filled when mask != NULL); `work` is a field-sized scratch array.
int lsm_reinitialize(LsmHandle* h, void* phi, const void* mask, void* work, int order, int upsample, int maxiters, double xtol, double ftol,
                     int64_t* ncandidate_cells, int64_t* nfail, int64_t* nfar) {
    if (!h || !phi || !work) return h ? fail(h, LSM_ERR_INVALID, "lsm_reinitialize: null argument") : LSM_ERR_INVALID;
    if (upsample < 1 || upsample > 16) return fail(h, LSM_ERR_INVALID, "lsm_reinitialize: upsample must be in 1..16");
    if (maxiters < 1) return fail(h, LSM_ERR_INVALID, "lsm_reinitialize: maxiters must be positive");
    if (!(xtol > 0) || !(ftol > 0)) return fail(h, LSM_ERR_INVALID, "lsm_reinitialize: tolerances must be positive");
    // a slab is fine for a band whose slab carries its neighbours' planes (every node's samples and patches lie
    // within the band's reach; the caller refreshes the overlap planes afterwards); a dense slab is not
    if (!mask) LSM_TRY(check_single_device(h));
    if (h->comm) LSM_SYNC(h);           // the pipeline's own host waits follow a stream an exchange can no longer hold up
    const int N = h->grid.ndim;
    double lc[3] = {0, 0, 0};
    for (int d = 0; d < N; ++d) lc[d] = h->grid.lc[d];
    long long counts[3] = {0, 0, 0};
    const char* err = nullptr;
    const int r = reinit_run(N, h->nloc, h->goff, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->lay.total, lc, h->h, order, upsample, maxiters, xtol, ftol,
                             phi, is_f32(h), (const unsigned char*)mask, work, h->stream, counts, &err, &h->reinit_ws);
    if (r == 1) return fail(h, LSM_ERR_INVALID, err ? err : "lsm_reinitialize");
    if (r) return fail(h, LSM_ERR_HIP, err ? err : "lsm_reinitialize");
    if (ncandidate_cells) *ncandidate_cells = counts[0];
    if (nfail) *nfail = counts[1];
    if (nfar) *nfar = counts[2];
    return LSM_OK;
}

// InterpolatedField(ϕ, order) evaluated at points — src/interpolation.jl:117-151,228-260
int lsm_interpolate(LsmHandle* h, void* phi, int order, int64_t npoints, const void* points, void* values, void* gradients, void* hessians,
                    void* stream) {
    if (!h || !phi || (npoints > 0 && (!points || !values))) return h ? fail(h, LSM_ERR_INVALID, "lsm_interpolate: null argument") : LSM_ERR_INVALID;
    if (npoints < 0) return fail(h, LSM_ERR_INVALID, "lsm_interpolate: negative point count");
    LSM_TRY(check_single_device(h));
    LSM_TRY(lsm_fill_ghosts(h, phi, 7, stream));
    const int N = h->grid.ndim;
    double lc[3] = {0, 0, 0};
    for (int d = 0; d < N; ++d) lc[d] = h->grid.lc[d];
    const char* err = nullptr;
    const int r = interp_run(N, h->nloc, h->goff, h->lay.stride[1], h->lay.stride[2], h->lay.origin, lc, h->h, order, phi, is_f32(h), npoints,
                             (const double*)points, (double*)values, (double*)gradients, (double*)hessians, stream ? (hipStream_t)stream : h->stream, &err);
    if (r == 1) return fail(h, LSM_ERR_INVALID, err ? err : "lsm_interpolate");
    if (r) return fail(h, LSM_ERR_HIP, "lsm_interpolate: launch failed");
    return LSM_OK;
}

// NewtonSDF (src/sdf.jl:57-127): build once, query signed distances at points, read the samples back
struct LsmSdf { LsmHandle* h; SdfObject* o; int ndim; };
int lsm_sdf_create(LsmHandle* h, void* phi, const void* mask, int order, int upsample, int maxiters, double xtol, double ftol, LsmSdf** out,
                   int64_t* nsamples) {
    if (!h || !phi || !out) return h ? fail(h, LSM_ERR_INVALID, "lsm_sdf_create: null argument") : LSM_ERR_INVALID;
    if (upsample < 1 || upsample > 16) return fail(h, LSM_ERR_INVALID, "lsm_sdf_create: upsample must be in 1..16");
    if (maxiters < 1) return fail(h, LSM_ERR_INVALID, "lsm_sdf_create: maxiters must be positive");
    if (!(xtol > 0) || !(ftol > 0)) return fail(h, LSM_ERR_INVALID, "lsm_sdf_create: tolerances must be positive");
    LSM_TRY(check_single_device(h));
    const int N = h->grid.ndim;
    double lc[3] = {0, 0, 0};
    for (int d = 0; d < N; ++d) lc[d] = h->grid.lc[d];
    const char* err = nullptr;
    SdfObject* o = nullptr;
    long long ns = 0;
    const int r = sdf_build(N, h->nloc, h->goff, h->lay.stride[1], h->lay.stride[2], h->lay.origin, h->lay.total, lc, h->h, order, upsample, maxiters,
                            xtol, ftol, phi, is_f32(h), (const unsigned char*)mask, h->stream, &o, &ns, &err);
    if (r == 1) return fail(h, LSM_ERR_INVALID, err ? err : "lsm_sdf_create");
    if (r) return fail(h, LSM_ERR_HIP, err ? err : "lsm_sdf_create");
    *out = new LsmSdf{h, o, N};
    if (nsamples) *nsamples = ns;
    return LSM_OK;
}
int lsm_sdf_eval(LsmSdf* s, int64_t npoints, const void* points, void* distances, void* closest_points, int64_t* nfail) {
    if (!s || npoints < 0 || (npoints > 0 && (!points || !distances))) return LSM_ERR_INVALID;
    const char* err = nullptr;
    long long nf = 0;
    if (sdf_eval(s->o, npoints, (const double*)points, (double*)distances, (double*)closest_points, &nf, &err))
        return fail(s->h, LSM_ERR_HIP, err ? err : "lsm_sdf_eval");
    if (nfail) *nfail = nf;
    return LSM_OK;
}
int lsm_sdf_samples(LsmSdf* s, void* points_out) {
    if (!s || !points_out) return LSM_ERR_INVALID;
    const char* err = nullptr;
    if (sdf_samples(s->o, (double*)points_out, &err)) return fail(s->h, LSM_ERR_HIP, err ? err : "lsm_sdf_samples");
    return LSM_OK;
}
void lsm_sdf_destroy(LsmSdf* s) {
    if (!s) return;
    sdf_free(s->o);
    delete s;
}

int lsm_cfl_cache(LsmHandle* h, int enable) {
    if (!h) return LSM_ERR_INVALID;
    h->cfl_cache_on = enable != 0;
    h->cfl_cache.clear();
    h->band_cfl.armed = h->band_cfl.valid = h->band_cfl.pending = false;   // a prefetched band Δt goes with the cache
    h->band_cfl.nterms = 0;
    for (auto& e : h->cfl_cand) (void)hipFree(e.d_cand);
    h->cfl_cand.clear();
    return LSM_OK;
}

int lsm_profile_enable(LsmHandle* h, int on) {
    if (!h) return LSM_ERR_INVALID;
    h->prof = on != 0;
    h->prof_every = on > 1 ? on : 1;
    h->prof_seen = 0;
    h->ev_used = 0;
    return LSM_OK;
}

int lsm_profile_read(LsmHandle* h, int64_t* n_stage_launches, double* stage_ms_total) {
    if (!h || !n_stage_launches || !stage_ms_total) return LSM_ERR_INVALID;
    LSM_SYNC(h);
    double tot = 0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0;
        LSM_HIP(h, hipEventElapsedTime(&ms, h->ev_start[i], h->ev_stop[i]));
        tot += ms;
    }
    // every launch counts; with a sampling period the time of the sampled launches stands for all of them
    *n_stage_launches = (int64_t)h->prof_seen;
    *stage_ms_total = h->ev_used ? tot * ((double)h->prof_seen / (double)h->ev_used) : 0.0;
    h->ev_used = 0;
    h->prof_seen = 0;
    return LSM_OK;
}

}  // extern "C"
