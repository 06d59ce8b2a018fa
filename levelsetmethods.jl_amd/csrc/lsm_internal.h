// lsm_internal.h — structs shared by the host API (lsm_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lsm.h"

namespace lsm {

// Field values are fp64, or fp32 for LSM_DTYPE_F32 handles (widened exactly on load, rounded to nearest on
// store; every computation is fp64).  The small kernels take the storage type as a run-time flag.
__device__ __forceinline__ double ld_val(const void* p, long long i, int f32) {
    return f32 ? (double)static_cast<const float*>(p)[i] : static_cast<const double*>(p)[i];
}
__device__ __forceinline__ void st_val(void* p, long long i, int f32, double v) {
    if (f32) static_cast<float*>(p)[i] = (float)v;
    else static_cast<double*>(p)[i] = v;
}

// Tuning switches (include/lsm.h, lsm_set_tuning).  The environment is read ONCE per process (lsm_tuning_env), every handle starts
// from that copy, lsm_set_tuning changes one handle's; the launchers see the handle's through StageArgs::tune.
struct LsmTuning {
    int stage_tail;         // LSM_STAGE_TAIL        planes per chunk of a dense 3-D launch's graded tail (0 = no tail)          16
    int stage_tail_dyn;     // LSM_STAGE_TAIL_DYN    % spare workgroups of the dynamic tail (0 = static tail)                    25
    int stage_mc;           // LSM_STAGE_MC          planes per march chunk in 3-D (0 = 64, shorter on small grids)               0
    int stage_mc2;          // LSM_STAGE_MC2         rows per march chunk in 2-D (0 = 8)                                         0
    int pairs;              // LSM_PAIRS             two nodes per thread for the dense single-term kernels                      1
    int stage_generic;      // LSM_STAGE_GENERIC     general stage variants instead of the plain ones (diagnostic)               0
    int xredirect;          // LSM_XREDIRECT         x / y ghosts of copy-type faces served by the stage kernel's loads          1
    int mredirect;          // LSM_MREDIRECT         ... and NeumannBC faces of the march axis by clamping the march: no fill left  1
    int ghost_full_depth;   // LSM_GHOST_FULL_DEPTH  fills write all three ghost layers whatever the step reads                  0
    int band_bricks;        // LSM_BAND_BRICKS       band stage with one lane per band node (stage_brick.h)                      1
    int band_bits;          // LSM_BAND_BITS         update_band! on bit rows                                                     1
    int band_cfl_prefetch;  // LSM_BAND_CFL_PREFETCH Δt of the next step reduced right behind the update                         1
    int band_bytes;         // LSM_BAND_BYTES        byte-mask band kernels in 3-D too (the general path, forced)                0
    int band_no_lists;      // LSM_BAND_NO_LISTS     band kernels over all tiles instead of the compact lists (forced)           0
    int status_spin;        // LSM_STATUS_SPIN       lsm_band_status spins on the status kernel's ticket                         1
    int slab_overlap;       // LSM_SLAB_OVERLAP      slab stages update the interface planes first (read at attach time)         1
    int comm_timeout_ms;    // LSM_COMM_TIMEOUT_MS   a rank's wait for its peers                                              60000
    int layout_align;       // LSM_LAYOUT_ALIGN      rows of the padded layout on 64-byte lines (read by lsm_create)             1
};
const LsmTuning& lsm_tuning_env();
int* lsm_tuning_field(LsmTuning& t, const char* name);      // NULL: no such switch

// term-kind slots inside a fused stage kernel
enum { SLOT_ADV = 0, SLOT_NM = 1, SLOT_CURV = 2, SLOT_EIK = 3, NSLOTS = 4 };

// device view of an LsmCoeff, time factor already evaluated on the host for this stage
struct CoeffArgs {
    int kind;
    int _pad;
    double v[4];
    double tfac;
    const double* f[3];
    const double* sep[3];
};

// everything a stage kernel needs, passed by value
struct StageArgs {
    // geometry (local slab)
    int n[3];          // local interior extent
    int goff[3];       // global index of local (0,0,0)
    int gn[3];         // global node counts
    long long s1, s2;  // strides of dim 1 and dim 2 (elements)
    long long origin;
    double lc[3];
    double h[3];       // meshsize (src/meshes.jl:110), computed on the host in IEEE double
    double h2[3];      // h*h
    double inv_h[3];
    double inv_h2[3];
    double dxmin;      // minimum(meshsize)
    int uniform_h;     // inv_h2 equal in every used dimension (FAST build: the Godunov sums are scaled once, not per dimension)
    // fields
    const double* psi;
    const double* phin;
    double* out;
    double* out2;
    int base_mode;     // LSM_BASE_*
    double base_a, base_b;   // FAST build: base = base_a·ϕⁿ + base_b·ψ (set by the launcher from base_mode)
    int out2_accum;    // 0: out2 starts from psi, 1: out2 accumulates onto itself (multi-pass)
    double cdt, cdt2;
    // terms of this pass, in user order: order[k] is a SLOT_*
    int nterms;
    int order[NSLOTS];
    int natural;       // order[] lists the pass's slots in ascending order (the kernel then needs no look-up)
    int adv_scheme;
    CoeffArgs adv, nm, curv;
    const double* s0;  // Eikonal frozen sign (NULL = current-sign mode)
    unsigned nb[3];    // tiles along x, y (3-D only) and march chunks — set by the launcher
    int mb, me;        // range [mb, me) of the march (last) dimension to update
    int mc;            // planes per march chunk (0 = the tile's compile-time default)
    // graded tail (3-D dense launches; set by the launcher, 0 = off): the first nbig chunk layers have mc planes, the rest of
    // [mb, me) is cut into chunks of mc_tail planes whose workgroups are dispatched LAST on every XCD — the launch drains
    // over the duration of a short workgroup instead of a long one
    unsigned nbig;
    int mc_tail;
    // x ghosts resolved by the loads themselves (set by the whole-grid lsm_advance_* in FAST mode when both x faces copy ONE
    // node: periodic / symmetry / degree-0 extrapolation): a load of a node with x outside [0, n0) goes to the node the boundary
    // condition copies instead, and the ghost fill before the stage skips the x faces.  xkind[side] = LSM_BC_* of the x faces.
    // dynamic tail (set by the launcher when the graded tail is on and the handle lends its counter): the workgroups behind
    // the long chunks take the short chunks in the order they START, through one ticket counter — the eight XCDs get equal
    // numbers of workgroups from the hardware but do not run at the same speed; 30 % spare tail workgroups let the faster
    // ones take more of the tail.  ticket = atomicAdd(tail_ctr, 1); tickets past the last short chunk leave.  Every launch is
    // self-contained: exactly tail_wgs workgroups draw, and the one that draws the last ticket sets the counter back to 0.  The
    // handle lends a RING of LSM_TAIL_SLOTS counters and every launch takes the next one, so a counter is re-used only
    // LSM_TAIL_SLOTS launches later; launches on a stream other than the handle's own get no counter at all (tail_ring ==
    // NULL: static tail) — two launches that could run concurrently never share one, and a launch that failed leaves nothing
    // behind (the slot still holds 0).
    unsigned* tail_ctr;                // this launch's counter (set by the launcher from tail_ring)
    unsigned tail_wgs;
    unsigned* tail_ring;               // LSM_TAIL_SLOTS counters, all 0 between launches (NULL = no dynamic tail)
    unsigned* tail_slot_host;          // host side: the handle's running launch count (selects the slot)
    int xredirect;
    int xkind[2];
    int yredirect;     // the same for dimension 2 of a 3-D grid (the tile's y): its ghost rows are skipped by the fills as well
    int ykind[2];
    int mredirect[2];  // march axis, per face: NeumannBC — the ghost planes ARE the boundary plane, the march clamps there and they are never read
    const unsigned char* mask;         // narrow band: store only where mask != 0 (NULL = dense)
    const unsigned char* tile_active;  // narrow band: per-tile activity flags (NULL = all tiles)
    const int* tile_list;              // narrow band: compact list of the tiles to run (NULL = all tiles get a block)
    unsigned ntile_list;
    const int* brick_list;             // narrow band: the active tiles, one brick of `mc` planes each (stage_brick.h; NULL = none)
    unsigned nbrick_list;
    int f32;                           // psi / phin / out / out2 hold float (LSM_DTYPE_F32); side arrays stay fp64
    unsigned long long* stamp;         // diagnostic build (-DLSM_STAMP, `make stamp`): per-workgroup {Δs_memtime, Δs_memrealtime} of the plane loop
    const LsmTuning* tune;             // host side: the handle's tuning switches (the launchers read them; never NULL)
};

#define LSM_TAIL_SLOTS 16

struct GhostArgs {
    int n[3];
    long long s1, s2, origin;
    int dim;             // dimension being filled
    int kind[2];         // per side
    int degree[2];
    double w[2][LSM_GHOST][8];  // Lagrange weights per side, ghost distance k-1, node j
    void* v;
    int f32;
};

// all dimensions at once (fused ghost fill)
struct GhostAllArgs {
    int n[3];
    long long s1, s2, origin;
    int kind[3][2];
    int degree[3][2];
    const double* w;   // device copy of the Lagrange weights [3][2][LSM_GHOST][8] (no dynamic kernarg indexing)
    void* v;
    int f32;
    int mb, me;        // planes [mb, me) of the last dimension whose lower-dimension ghosts are filled
    int fill_last;     // also fill the (physical) ghosts of the last dimension
    int skip_x;        // leave the ghost nodes at the ends of the interior rows alone (StageArgs::xredirect: nobody reads them)
    int skip_y;        // 3-D: leave the ghost rows of dimension 2 of the interior planes alone (StageArgs::yredirect)
    int depth;         // ghost layers to fill, 1..LSM_GHOST: what the stencils of the step about to run read (a ghost's value
                       // depends on interior nodes only, so the outer layers may stay stale)
};

struct CflArgs {
    int n[3];
    int goff[3];
    int gn[3];
    long long s1, s2, origin;
    double lc[3];
    double h[3];
    int term_kind;
    CoeffArgs coeff;
    double* partial;     // one value per block
    int* nanflag;        // set to 1 if any node produced NaN
    const unsigned char* mask;   // narrow band: only band nodes count (NULL = all nodes)
    const unsigned* rowbits;     // narrow band, list kernel: the band as 32-bit row words, tile-major (word ly + ty·i of tile t = x-line ly of plane i: what
                                 // lsm_band_update's kernels leave behind) — 512 bytes per tile instead of 4096 scattered mask bytes; NULL = read the mask
    const unsigned char* tile_active;   // narrow band: per-tile activity flags of tx × ty × tm bricks (NULL = none)
    int tx, ty, tm;
    unsigned nbx, nby;
    long long* cand;             // pass 2: candidate nodes (i0 + n0·i1 + ncol·m) are appended here
    unsigned* cand_count;
    unsigned cand_cap;
};

// narrow-band kernels (lsm_band.hip)
struct BandArgs {
    int ndim;
    int n[3];
    long long s1, s2, origin;
    int tx, ty, tm;              // tile footprint (x, y [3-D only], last dimension): the stage kernel's bricks
    unsigned nbx, nby, nbm;      // tiles per direction
    const unsigned char* work;   // per-tile flags: tiles to visit (NULL = all)
    const int* list;             // compact list of the tiles to visit (NULL = every tile gets a block)
    unsigned nlist;
    int f32;                     // the value arrays hold float
    int force_bytes;             // LsmTuning::band_bytes: byte-mask kernels in 3-D too
};
void launch_band_copy(const BandArgs& a, const unsigned char* in, unsigned char* out, unsigned char* zero, hipStream_t s);
void launch_band_work(const BandArgs& a, const unsigned char* active, unsigned char* work, unsigned* zero_face, int* zero_flags, hipStream_t s);
void launch_band_cut(const BandArgs& a, const void* v, const unsigned char* old_mask, unsigned char* seed, hipStream_t s);
void launch_band_dilate(const BandArgs& a, const unsigned char* in, unsigned char* out, hipStream_t s);
void launch_band_grow(const BandArgs& a, const void* v, const unsigned char* old_mask, int nl, unsigned char* new_mask,
                      unsigned char* tiles, hipStream_t s);
bool band_grow_fits(const BandArgs& a, int nl);
struct BandEntry {          // a halo node and its nearest band node (16 bytes)
    long long q;            // padded index of the node
    int rel;                // padded index of the nearest band node minus q
    signed char d[4];       // I - P per dimension
};
void launch_band_zero(const BandArgs& a, unsigned char* out, hipStream_t s);
// bit-row path of update_band! (lsm_band.hip)
bool band_bits_fit(const BandArgs& a, int nl);
void launch_band_bits(const BandArgs& a, const void* v, const unsigned char* mask, unsigned* OB, unsigned* LE, unsigned* GE,
                      const unsigned char* flags_src, unsigned char* flags_dst, unsigned* zero0, hipStream_t s);
void launch_band_grow_bits(const BandArgs& a, void* v, unsigned char* mask, int nl, const unsigned char* old_tiles, unsigned char* tiles,
                           const unsigned* OB, const unsigned* LE, const unsigned* GE, unsigned* NB, int* miss, hipStream_t s);
void launch_band_copy_values(const BandArgs& a, const unsigned char* mask, const void* src, void* dst, hipStream_t s);
struct BandStatusCfl {         // the prefetched Δt reductions lsm_band_status finishes: n slots of npartials partial maxima each
    int n, npartials;
    int kind[4];               // LSM_TERM_* of the slot's term
    const double* partial;
    const int* nanflag;
    double dxmin;
};
void launch_band_status(const unsigned* halo_count, const int* miss, const unsigned* lcounts, const BandStatusCfl& pf, double* out, double ticket,
                        hipStream_t s);
void launch_band_extrapolate(const BandArgs& a, const unsigned char* target, unsigned char* halo, const unsigned char* src_mask,
                             const signed char* ring, int nring, int nring_lds, const void* src, void* dst, int* miss,
                             BandEntry* list, unsigned* list_count, unsigned list_cap, hipStream_t s);
void launch_band_halo_bits(const BandArgs& a, const unsigned char* tiles, const unsigned* NB, unsigned char* halo, int* miss, BandEntry* list,
                           unsigned* list_count, unsigned list_cap, hipStream_t s);
void launch_band_apply(const BandArgs& a, const BandEntry* list, const unsigned* list_count, long long n_host, unsigned list_cap,
                       const unsigned char* src_mask, const void* src, void* dst, hipStream_t s);
struct BandBcArgs { int kind[3][2]; int degree[3][2]; };
void launch_band_halo_bc(const BandArgs& a, const BandBcArgs& bc, int d, int r, const unsigned char* band, unsigned char* halo,
                         hipStream_t s);
void launch_band_lists(const BandArgs& a, const unsigned char* active, const unsigned char* work, int* act_list, int* work_list, unsigned* counts,
                       hipStream_t s);
void launch_band_tiles(const BandArgs& a, const unsigned char* mask, unsigned char* tiles, hipStream_t s);
void launch_band_count(const BandArgs& a, const unsigned char* mask, unsigned long long* count, hipStream_t s);
// tile geometry of the stage kernel for a given dimension and march chunk (stage_tu.hip)
void stage_tile_shape(int ndim, int* tx, int* ty);

// reinitialize! (lsm_reinit.hip)
struct ReinitWorkspace;      // device buffers of reinitialize! kept between calls (lsm_reinit.hip); NULL workspace pointer = allocate and free per call
void reinit_workspace_free(ReinitWorkspace* w);
int reinit_run(int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, long long total, const double lc[3],
               const double h[3], int order, int upsample, int maxiters, double xtol, double ftol, void* phi, int f32, const unsigned char* mask,
               void* out_field, hipStream_t stream, long long out_counts[3], const char** err, ReinitWorkspace** workspace);

int interp_run(int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, const double lc[3], const double h[3], int order,
               const void* phi, int f32, long long npts, const double* pts, double* val, double* grad, double* hess, hipStream_t stream, const char** err);

struct SdfObject;
int sdf_build(int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, long long total, const double lc[3],
              const double h[3], int order, int upsample, int maxiters, double xtol, double ftol, const void* phi, int f32,
              const unsigned char* mask, hipStream_t stream, SdfObject** out, long long* nsamples, const char** err);
int sdf_eval(SdfObject* o, long long npts, const double* xs, double* dist, double* cps, long long* nfail, const char** err);
int sdf_samples(SdfObject* o, double* out, const char** err);
void sdf_free(SdfObject* o);

// compile-time description of one instantiated fused kernel
struct Combo {
    int adv;   // 0 none, 1 upwind, 2 weno5
    int nm;    // 0/1
    int curv;  // 0/1
    int eik;   // 0 none, 1 frozen S0, 2 current sign
};
// stage_brick.hip: the band stage with one lane per band node; -1 = not its case (the tiled kernels take the launch)
int launch_stage_brick(const Combo& c, const StageArgs& a, hipStream_t s);

// launchers implemented in stage_{fast,strict}.hip; return 0 if the combo is instantiated
int launch_stage_fast(int ndim, const Combo& c, const StageArgs& a, hipStream_t s);
int launch_stage_strict(int ndim, const Combo& c, const StageArgs& a, hipStream_t s);
bool combo_available(const Combo& c);

// small kernels (lsm_aux.hip)
void launch_ghost_fill(int ndim, const GhostArgs& a, hipStream_t s);
void launch_ghost_fill_all(int ndim, const GhostAllArgs& a, hipStream_t s);
int cfl_blocks(int ndim, const int n[3]);
void launch_cfl(int ndim, const CflArgs& a, int nblocks, int pass, const double* thresh, hipStream_t s);
void launch_cfl_candidates(int ndim, const CflArgs& a, unsigned count, hipStream_t s);
int launch_cfl_band_list(const CflArgs& a, const int* list, unsigned nlist, const unsigned* nlist_dev, int max_partials, hipStream_t s);
void launch_cfl_final(const double* partial, int nblocks, const int* nanflag, double* out, int term_kind, double dxmin,
                      int pass, hipStream_t s);
void launch_extrema(int ndim, const int n[3], long long s1, long long s2, long long origin, const void* v, int f32,
                    double* partial_min, double* partial_max, int nblocks, double* out2, hipStream_t s);
void launch_measure(int mode, int ndim, const int n[3], long long s1, long long s2, long long origin, const double h[3],
                    double dmin, double scale, const void* v, int f32, double* partial, int nblocks, double* out, hipStream_t s,
                    const unsigned char* mask = nullptr);
int launch_band_volume(int ndim, const int n[3], long long s1, long long s2, long long origin, double dmin, double scale, const void* v, int f32,
                       const unsigned char* mask, double* out, hipStream_t s);
void launch_signed_normals(int ndim, const int n[3], long long s1, long long s2, long long origin, const double h[3], double delta,
                           double band_width, double min_norm2, const void* phi, int f32, const double* frozen, double* c0, double* c1,
                           double* c2, hipStream_t s);
void launch_geometry(int what, int ndim, const int n[3], long long s1, long long s2, long long origin, const double h[3], double scale,
                     double band_width, double fill, const void* phi, int f32, double* o0, double* o1, double* o2, double* frozen,
                     hipStream_t s, const unsigned char* mask = nullptr);
void launch_eikonal_sign(int ndim, const int n[3], long long s1, long long s2, long long origin, double dxmin,
                         const void* phi0, int f32, double* s0, hipStream_t s);

}  // namespace lsm
