/*
 * lsm.h — C ABI of libhiplsm: the MI355X (gfx950) grid-update hot path of
 * LevelSetMethods.jl behind plain pointers and sizes.
 *
 * The reference has no FFI: its seam is Julia multiple dispatch on the field type
 * (AbstractMeshField interface, src/meshfield.jl:11-33).  The entry points below are what a
 * device-resident AbstractMeshField subtype would `ccall` from the methods the integrator
 * reaches the field through.  Each declaration cites the reference code whose
 * loop body it replaces.  No torch/HIP types appear in signatures; `stream` is a hipStream_t
 * passed as void* (NULL = the handle's own stream).
 *
 * Conventions
 *  - All field pointers are DEVICE pointers to arrays in the *padded layout* described by
 *    LsmLayout (column-major like Julia's Array, dim 1 fastest, LSM_GHOST ghost layers on each
 *    side of every used dimension).  The caller owns the memory (AMDGPU.jl ROCArray / torch
 *    tensor); the library borrows pointers for the duration of a call.
 *  - Every call returns LSM_OK (0) or a negative error code; lsm_last_error() gives the text.
 *    The library never throws or aborts.
 *  - Calls are asynchronous on the handle's stream unless documented otherwise; calls that
 *    return a host value (lsm_compute_cfl, lsm_download) synchronise.
 */
#ifndef LSM_H
#define LSM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSM_MAX_DIM 3
#define LSM_GHOST 3          /* ghost layers per side: WENO5 needs 3 (src/derivatives.jl:89-121) */
#define LSM_MAX_TERMS 8

/* status codes */
enum {
    LSM_OK = 0,
    LSM_ERR_INVALID = -1,    /* bad argument / unsupported combination */
    LSM_ERR_HIP = -2,        /* a HIP runtime call failed */
    LSM_ERR_NO_DEVICE = -3,
    LSM_ERR_COMM = -4        /* multi-GPU: a peer rank left / aborted / did not answer in time, or RCCL failed; the
                                communicator stays failed (lsm_comm_detach + a new attach to go on) */
};

/* CartesianGrid (src/meshes.jl:1-5): lower/upper corner and node counts of the GLOBAL grid.
 * meshsize h_d = (hc_d - lc_d)/(n_d - 1) (src/meshes.jl:109-110). */
typedef struct LsmGrid {
    int32_t ndim;                 /* 1, 2 or 3 */
    int32_t _pad;
    int64_t n[LSM_MAX_DIM];       /* unused dims must be 1 */
    double lc[LSM_MAX_DIM];
    double hc[LSM_MAX_DIM];
} LsmGrid;

/* BoundaryCondition kinds (src/boundaryconditions.jl:27,40-46,63).  LSM_BC_NONE marks a slab
 * interface of a multi-GPU decomposition: its ghosts come from the halo exchange, the ghost
 * fill skips it. */
enum { LSM_BC_PERIODIC = 0, LSM_BC_EXTRAPOLATION = 1, LSM_BC_SYMMETRY = 2, LSM_BC_NONE = 3 };
typedef struct LsmBc {
    int32_t kind;
    int32_t degree;               /* P of ExtrapolationBC{P}; 0 = NeumannBC, 1 = LinearExtrapolationBC */
} LsmBc;

/* Slab of the last dimension owned by this handle (multi-GPU); lo is 0-based. */
typedef struct LsmSlab {
    int64_t lo;
    int64_t n;
} LsmSlab;

/* Padded device layout of one scalar field. element(i1,i2,i3) (0-based LOCAL interior index,
 * ghosts at -LSM_GHOST..-1 and n..n+LSM_GHOST-1) lives at origin + i1 + i2*stride[1] + i3*stride[2]. */
typedef struct LsmLayout {
    int64_t n[LSM_MAX_DIM];       /* local interior extent */
    int64_t g[LSM_MAX_DIM];       /* ghost width per dim (0 for unused dims) */
    int64_t stride[LSM_MAX_DIM];  /* stride[0] == 1 */
    int64_t origin;               /* offset of interior node (0,0,0) */
    int64_t total;                /* elements to allocate */
} LsmLayout;

/* LevelSetTerm kinds (src/levelsetterms.jl:45,104,139,211) and SpatialScheme (src/derivatives.jl:11,20) */
enum { LSM_TERM_ADVECTION = 0, LSM_TERM_NORMAL_MOTION = 1, LSM_TERM_CURVATURE = 2, LSM_TERM_EIKONAL = 3 };
enum { LSM_SCHEME_UPWIND = 0, LSM_SCHEME_WENO5 = 1 };

/* Coefficient of a term (velocity / speed / b), the device-side counterpart of
 * _eval_field (src/levelsetterms.jl:42-43).  Julia closures cannot run on the device, so:
 *   CONST      value[c]                                         (e.g. (x,t)->SVector(1.0))
 *   ROTATION   u1 = -(w*(x2-c2)), u2 = w*(x1-c1), u3 = 0         with w=value[0], c=value[1..2]
 *   SEPARABLE  u_c = ((T_c1[i1]*T_c2[i2])*T_c3[i3]) * g(t)       tables on the device, GLOBAL index;
 *              sep[c] points to the n1+n2+n3 concatenated doubles of component c
 *              g(t) = 1 (time_kind 0) or cos(pi*t/time_param) (time_kind 1)
 *   FIELD      field[c] = padded device array (same layout as phi), read in-grid only
 * Node coordinates are x_d = lc_d + i_d*h_d (src/meshes.jl:114-117). */
enum { LSM_COEFF_CONST = 0, LSM_COEFF_ROTATION = 1, LSM_COEFF_SEPARABLE = 2, LSM_COEFF_FIELD = 3 };
enum { LSM_TIME_ONE = 0, LSM_TIME_COS = 1 };
typedef struct LsmCoeff {
    int32_t kind;
    int32_t time_kind;
    double time_param;
    double value[4];
    const void* field[LSM_MAX_DIM];
    const double* sep[LSM_MAX_DIM];
} LsmCoeff;

typedef struct LsmTerm {
    int32_t kind;                 /* LSM_TERM_* */
    int32_t scheme;               /* LSM_SCHEME_* (advection only) */
    LsmCoeff coeff;               /* velocity (ndim comps) / speed / b (1 comp); unused for Eikonal */
    const void* s0;               /* Eikonal frozen sign field S0 (padded, src/levelsetterms.jl:217-221) or NULL */
} LsmTerm;

/* How a stage forms its base value before subtracting the terms (src/timestepping.jl:126-202):
 *   PSI     base = psi[I]                          (FE, RK2 predictor, RK3 stage 1; :129,:147,:172)
 *   RK3_S2  base = 0.75*phin[I] + 0.25*psi[I]      (:182)
 *   RK3_S3  base = (phin[I] + 2*psi[I]) / 3        (:193)
 *   OTHER   base = phin[I]                         (RK2 corrector accumulates onto corr; :160)
 */
enum { LSM_BASE_PSI = 0, LSM_BASE_RK3_S2 = 1, LSM_BASE_RK3_S3 = 2, LSM_BASE_OTHER = 3 };

/* arithmetic mode (lsm_create flags) */
enum {
    LSM_MODE_FAST = 0,    /* default: reciprocal-based divisions, FMA contraction, fused WENO weights;
                             within 1e-13*max|phi| of the reference arithmetic per stage */
    LSM_MODE_STRICT = 1   /* literal reference operation order, IEEE division, no contraction */
};
/* Domain of LSM_MODE_FAST (the 1e-13 bound above holds inside it):
 *  - differences between neighbouring nodes must stay below 1e35 in magnitude: the WENO5 weights are formed as products
 *    of squared smoothness indicators with ONE reciprocal (src/derivatives.jl:73-78 divides six times), which leave the
 *    fp64 range beyond that and produce Inf/NaN.  LSM_FAST_MAX_ABS bounds |phi| so that no difference can get there;
 *    lsm_check_range tests a field against it (the host layers call it when an equation is built and every 64 steps, and
 *    refuse to go on: rescale the field or create the handle with LSM_MODE_STRICT, which has no such limit).
 *  - the floor of the WENO5 regularisation eps = 1e-6*max(v_k^2) + 1e-99 (src/derivatives.jl:72) is, in undivided
 *    differences, 1e-99*h^2; FAST raises it to 1e-75 so that the weight denominator needs no guard.  Results change only
 *    where all five differences of a stencil are below ~3e-35, and there by less than those differences: a deviation
 *    from the reference by design, far inside the tolerance; exactly flat data gives exactly 0 in both modes.
 *  - an advection velocity component u_d = +0.0 takes the left-biased stencil where the reference takes the right-biased
 *    one; the term is |0|*W = 0 either way (a NaN velocity gives NaN in both). */
#define LSM_FAST_MAX_ABS 2.5e34
/* storage type of the level-set fields of a handle (ϕ, stage buffers, extension targets).  LSM_DTYPE_F32 is a
 * storage format only: values widen exactly on load, every computation is fp64, results are rounded to nearest
 * on store (the reference's Float32 fields compute in mixed Float32/Float64 by Julia's promotion rules; its
 * results differ from the fp64 ones by O(1e-7) relative, and so do these).  Coefficient fields, frozen signs and
 * frozen masks ("side arrays") are fp64 for either dtype, and so is every reduction. */
enum { LSM_DTYPE_F64 = 0, LSM_DTYPE_F32 = 1 };

typedef struct LsmHandle LsmHandle;

/* Host callback run between stage launches so host-side update_func hooks keep their
 * (stage field, stage time) semantics (src/timestepping.jl:131,149,158,174,185,196).
 * Called with the stream synchronised. Return non-zero to abort the step. */
typedef int (*LsmStageHook)(void* user, int stage, const void* stage_field, double stage_time);

/* ---- lifetime (LevelSetEquation construction, src/levelsetequation.jl:59-78) ---- */
int lsm_create(const LsmGrid* grid, const LsmBc bc[LSM_MAX_DIM][2], const LsmSlab* slab /* NULL = whole grid */,
               int dtype, int mode, int device, LsmHandle** out);
void lsm_destroy(LsmHandle* h);
const char* lsm_last_error(const LsmHandle* h /* NULL = creation errors */);
const char* lsm_version(void);
int lsm_sync(LsmHandle* h);
/* adopt the stream the caller's array library works on (hipStream_t as void*; the handle owns
 * a non-blocking stream of its own until this is called) */
int lsm_set_stream(LsmHandle* h, void* stream);

/* ---- layout + transfers (MeshField <-> device field; values(ϕ), src/meshfield.jl:58) ---- */
int lsm_layout(const LsmHandle* h, LsmLayout* out);
int lsm_upload(LsmHandle* h, void* dev_padded, const void* host_dense);     /* interior only; synchronous */
int lsm_download(LsmHandle* h, const void* dev_padded, void* host_dense);   /* interior only; synchronous */
/* the same for the fp64 side arrays of a handle of either dtype (host dense array of double) */
int lsm_upload_f64(LsmHandle* h, void* dev_padded, const void* host_dense);
int lsm_download_f64(LsmHandle* h, const void* dev_padded, void* host_dense);

/* ---- ghost resolution: _getindexbc + bc_stencil (src/meshfield.jl:248-260,
 *      src/boundaryconditions.jl:107-153) materialised into the ghost layers, dim 1 -> dim N.
 *      dim_mask bit d fills dimension d+1 (7 = all). */
int lsm_fill_ghosts(LsmHandle* h, void* field, int dim_mask, void* stream);

/* ---- one loop body of _advance! (src/timestepping.jl:128-137,143-164,170-202), all terms fused:
 *      out[I]  = (base - cdt*L_1(psi)[I]) - cdt*L_2(psi)[I] ...
 *      out2[I] = (psi[I] - cdt2*L_1) - cdt2*L_2 ...   (RK2's corr accumulator; NULL to skip)
 *      psi needs valid ghosts; out/out2 ghosts are NOT filled. out may alias phin (pointwise).
 *      One padded plane of the field (the whole array in 1-D, a row in 2-D) must be smaller than 2 GiB: the kernel
 *      addresses a plane through a buffer descriptor (LSM_ERR_INVALID otherwise). */
int lsm_stage(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin,
              void* out, void* out2, int base_mode, double cdt, double cdt2, double t_stage, void* stream);

/* The same two calls restricted to planes [m_begin, m_end) of the LAST dimension, so a slab can
 * update and ghost-fill its boundary planes first, start the halo exchange, and update the interior
 * while the planes travel (ndim >= 2).  lsm_fill_ghosts_planes fills the ghosts of dimensions
 * 1..N-1 on those planes and, if fill_last != 0, the physical (non-NONE) ghost planes of dimension N. */
int lsm_stage_planes(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin,
                     void* out, void* out2, int base_mode, double cdt, double cdt2, double t_stage,
                     int64_t m_begin, int64_t m_end, void* stream);
int lsm_fill_ghosts_planes(LsmHandle* h, void* field, int64_t m_begin, int64_t m_end, int fill_last, void* stream);

/* ---- compute_cfl (src/levelsetterms.jl:22-38,90-96,123-127,172-178,250).  *dt_out is the raw
 *      minimum (Inf, NaN and <=0 possible): the CALLER raises the reference's ArgumentError. */
int lsm_compute_cfl(LsmHandle* h, const LsmTerm* terms, int nterms, const void* phi, double t, double* dt_out);

/* The CFL of a catalogued analytic coefficient WITHOUT time factor (ROTATION, SEPARABLE with
 * LSM_TIME_ONE) is the same every step; it is computed once per handle and reused as long as the
 * LsmTerm is byte-identical.  Calling this (with 0 or 1) flushes the cache; pass 0 when table
 * contents are mutated in place between steps. */
int lsm_cfl_cache(LsmHandle* h, int enable);

/* ---- _advance! per integrator.  phi's ghost layers are (re)filled on entry; on return of a whole-grid handle they are
 *      STALE (the interior is the new state): every entry point that reads ghosts fills them itself, and a caller of
 *      lsm_stage calls lsm_fill_ghosts first — one fill per step saved when steps follow each other.  (A slab returns
 *      with its ghost PLANES exchanged and valid, and expects them so on entry.)  In LSM_MODE_FAST without a hook, when both
 *      faces of dimension 1 copy one node (periodic, symmetry, degree-0 extrapolation = NeumannBC), the stage kernels of
 *      these calls resolve the ghosts of dimension 1 in their loads and the fills inside the step leave the row ends alone
 *      (likewise dimension 2 of a 3-D grid): do not read those ghost nodes afterwards without an lsm_fill_ghosts of your own.
 *      hook may be NULL (the reference's default no-op update_func, src/levelsetterms.jl:63).
 *      On a slab handle with a communicator attached these run the slab's step (see "multi-GPU" below). */
int lsm_advance_fe(LsmHandle* h, const LsmTerm* terms, int nterms, void* phi, void* buf1,
                   double tc, double dt, LsmStageHook hook, void* user);
int lsm_advance_rk2(LsmHandle* h, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2,
                    double tc, double dt, LsmStageHook hook, void* user);
int lsm_advance_rk3(LsmHandle* h, const LsmTerm* terms, int nterms, void* phi, void* buf1, void* buf2,
                    double tc, double dt, LsmStageHook hook, void* user);

/* ---- multi-GPU: the grid is cut into slabs of the LAST dimension, one handle per slab (lsm_create with an
 *      LsmSlab and LSM_BC_NONE on the faces towards neighbouring ranks — on BOTH faces of every rank when that dimension
 *      is periodic: the ring closes across the wrap, whose period is n-1, src/boundaryconditions.jl:107-119).  After every
 *      stage the LSM_GHOST full padded planes next to each interface are exchanged with rank±1 (they carry their own ghosts of
 *      the leading dimensions, so the corner composition of _getindexbc, src/meshfield.jl:248-260, is preserved), and Δt of
 *      compute_cfl is min-reduced over the ranks with NaN winning (src/levelsetterms.jl:22-28: `min` propagates NaN).
 *      Two transports:
 *        RCCL   one process per GPU.  Rank 0 calls lsm_comm_unique_id and hands the LSM_COMM_ID_BYTES to every rank out of
 *               band (MPI, a file, torch.distributed's store ...); then every rank calls lsm_comm_attach_rccl (collective).
 *               librccl is opened at run time; the planes travel as grouped ncclSend/ncclRecv over xGMI on a stream of the
 *               communicator's own, so that a stage's interior update overlaps the exchange of its boundary planes.
 *        LOCAL  all ranks are handles of ONE process (any devices): lsm_comm_attach_local(handles, world), one call.
 *               Planes move by peer copies.  lsm_halo_wait / lsm_allreduce_dt / lsm_advance_* block the calling thread until
 *               every rank of the group has made the matching call: drive the ranks from one host thread each, or from one
 *               thread stage by stage (lsm_stage_planes ..., lsm_halo_start on every handle, then lsm_halo_wait on every
 *               handle; reduce Δt yourself).
 *      With a communicator attached to a slab handle, lsm_advance_fe/rk2/rk3 run the slab's stages with the exchange
 *      overlapped (boundary planes first, interior while they travel) and expect phi's ghosts — boundary conditions and
 *      neighbour planes — valid on entry (they are on return; after writing the field from outside call
 *      lsm_fill_ghosts(h, phi, 7, NULL) + lsm_halo_exchange(h, phi)).  lsm_compute_cfl stays local: follow it with
 *      lsm_allreduce_dt.  lsm_destroy detaches. */
#define LSM_COMM_ID_BYTES 128
enum { LSM_COMM_NONE = 0, LSM_COMM_RCCL = 1, LSM_COMM_LOCAL = 2 };
int lsm_comm_unique_id(void* id_out /* LSM_COMM_ID_BYTES */);
int lsm_comm_attach_rccl(LsmHandle* h, const void* unique_id, int rank, int world);
int lsm_comm_attach_local(LsmHandle* const* handles, int world);     /* handles[r] = rank r */
int lsm_comm_detach(LsmHandle* h);
int lsm_comm_info(const LsmHandle* h, int* rank, int* world, int* transport);   /* any pointer may be NULL */
/* boundary-first stages with the exchange overlapped behind the interior update (default; LSM_SLAB_OVERLAP=0 in the
 * environment at attach time, or enable = 0, selects the plain stage -> ghost fill -> exchange order: same results) */
int lsm_comm_set_overlap(LsmHandle* h, int enable);
/* exchange of `field`'s ghost planes: start after the planes next to the interfaces are final on the handle's stream, wait
 * before anything reads the ghost planes (the handle's stream waits; no host synchronisation with RCCL) */
int lsm_halo_start(LsmHandle* h, void* field);
int lsm_halo_wait(LsmHandle* h);
int lsm_halo_exchange(LsmHandle* h, void* field);                    /* start + wait */
int lsm_allreduce_dt(LsmHandle* h, double* dt /* in: local, out: global */);   /* synchronous */
/* Failure: no call above blocks for ever.  When a rank leaves its group (lsm_comm_detach / lsm_destroy), calls lsm_comm_abort,
 * or does not show up within LSM_COMM_TIMEOUT_MS (environment, default 60000), every unfinished wait of the other ranks —
 * lsm_halo_wait, lsm_allreduce_dt, the slab lsm_advance_* — returns LSM_ERR_COMM, and so does every later call on that
 * communicator.  lsm_comm_abort may be called from any thread (e.g. by the rank whose update hook threw). */
int lsm_comm_abort(LsmHandle* h);

/* Slab-decomposed NarrowBandMeshField (BASELINE config 5's decomposition).  The band's operations reach farther across a slab
 * interface than a stencil does (nearest band node within 6 nodes, its slope neighbour, the stencil's 3:
 * src/meshfield.jl:481-530), so every rank creates its slab `overlap` planes larger towards each neighbouring rank (>= 10;
 * LsmSlab.lo / n describe the EXTENDED slab); the extra planes are ordinary planes of the slab, computed redundantly — right
 * on the owned planes — and refreshed from their owners:
 *   lsm_band_overlap_config  declares the overlap depth (after attaching the communicator)
 *   lsm_band_overlap_mask    after every lsm_band_update: the overlap planes of the byte mask from their owners (whole planes),
 *                            after which both sides list the band nodes of the exchanged planes in index order (no index
 *                            travels).  Synchronous.  Follow with lsm_band_overlap_values(vals), lsm_band_retile,
 *                            lsm_band_status, lsm_band_halo.
 *   lsm_band_overlap_values  after every stage (lsm_advance_band_* does it itself): the values of those band nodes, packed —
 *                            ~0.4 MB instead of 48 MB per direction at 768^3.  The handle's stream waits; no host sync with RCCL. */
int lsm_band_overlap_config(LsmHandle* h, int64_t overlap);
int lsm_band_overlap_mask(LsmHandle* h, void* mask);
int lsm_band_overlap_values(LsmHandle* h, void* field);

/* ---- EikonalReinitializationTerm(ϕ₀) constructor map: S0 = v/sqrt(v^2+Δx^2) (src/levelsetterms.jl:217-221) */
int lsm_eikonal_sign(LsmHandle* h, const void* phi0, void* s0_out, void* stream);

/* ---- min/max of the interior, for show (src/meshfield.jl:300-303) ---- */
int lsm_extrema(LsmHandle* h, const void* phi, double* vmin, double* vmax);
/* ---- is `phi` inside the domain of the handle's arithmetic mode?  *ok := 1 for LSM_MODE_STRICT handles and for fields
 *      with max|phi| <= LSM_FAST_MAX_ABS (NaN entries do not count: they propagate in both modes), else 0;
 *      *max_abs := max|phi| over the non-NaN interior values (may be NULL).  One reduction pass; synchronous. */
int lsm_check_range(LsmHandle* h, const void* phi, int* ok, double* max_abs);

/* ---- volume(ϕ) = prod(h)·Σ H(-ϕ) and perimeter(ϕ) = prod(h)·Σ δ(ϕ)‖∇ϕ‖ with the smoothed Heaviside /
 *      Dirac delta of width min(h) (src/levelsetops.jl:27-33,139-149,171-183) over the local slab;
 *      the usual posthook diagnostics.  lsm_perimeter refills phi's ghost layers (centred gradient).
 *      Synchronous. */
int lsm_volume(LsmHandle* h, const void* phi, double* out);
int lsm_perimeter(LsmHandle* h, void* phi, double* out);
/* ---- the same measures of a NarrowBandMeshField (src/levelsetops.jl:34-116,150-166), from the band alone.
 *      lsm_band_volume: prod(h)·(Σ_band H(-ϕ) + the number of off-band nodes inside): per grid line along dimension 1 the
 *      tails and the gaps between consecutive band nodes count when their end nodes are negative; a line without a band
 *      node takes the sign of the band node nearest (in index space) to its point n₁÷2, as the reference's KD-tree query
 *      does.  0 for an empty band.
 *      lsm_band_perimeter: the dense sum over the band nodes (the delta's support lies inside the band); phi must be a
 *      prepared stage input (lsm_band_prepare: the off-band neighbours of band nodes hold the extrapolated values).
 *      Single device; synchronous. */
int lsm_band_volume(LsmHandle* h, const void* phi, const void* mask, double* out);
int lsm_band_perimeter(LsmHandle* h, const void* phi, const void* mask, double* out);

/* ---- extend_along_normals!(F, ϕ; nb_iters, cfl, frozen, interface_band, min_norm)
 *      (src/velocityextension.jl:20-67): extends the speed F off the interface of ϕ by nb_iters
 *      first-order upwind sweeps of ∂τF + sign(ϕ) n·∇F = 0, frozen nodes held fixed.  F, phi and the
 *      work buffers (ndim+1 of them) are padded device arrays; frozen is NULL (band rule
 *      |ϕ| <= interface_band·Δ) or a padded array whose non-zero entries are frozen nodes.
 *      Ghosts of F are resolved with the handle's boundary conditions, as the reference does. */
int lsm_extend_along_normals(LsmHandle* h, void* F, void* phi, const void* frozen, void* work0, void* work1,
                             void* work2, void* work3, int nb_iters, double cfl, double interface_band,
                             double min_norm);

/* ---- curvature(ϕ, I), gradient(ϕ, I), normal(ϕ, I) (src/levelsetops.jl:197-226) at every node of the local slab,
 *      written to fp64 side arrays in the padded layout (out0 for the curvature, out0..out[ndim-1] for the vectors).
 *      phi's ghost layers are refilled on entry (centred differences reach one layer, corners included).
 *      The result is multiplied by `scale`.  band_width >= 0 evaluates only the nodes with |ϕ| <= band_width; the
 *      others get `fill`, and frozen_out (may be NULL; fp64 side array) := 1.0 on the evaluated nodes, 0.0 elsewhere —
 *      the seed-and-freeze loop of the reference's speed update functions (test/test-velocityextension.jl:118-131:
 *      v[I] = -curvature(ϕ, I) where |ϕ[I]| <= 1.5Δ, frozen there) in one launch, ready for
 *      lsm_extend_along_normals.  band_width < 0: every node.  Asynchronous on `stream` (NULL = the handle's). */
enum { LSM_GEOM_CURVATURE = 0, LSM_GEOM_GRADIENT = 1, LSM_GEOM_NORMAL = 2 };
int lsm_geometry(LsmHandle* h, int what, void* phi, double scale, double band_width, double fill, void* out0,
                 void* out1, void* out2, void* frozen_out, void* stream);
/*      The same at the active nodes of a NarrowBandMeshField (the reference's queries work on both field types,
 *      docs/src/geometry-queries.md): phi must be a prepared stage input (lsm_band_prepare: off-band neighbours hold the
 *      extrapolated values, out-of-grid ones the boundary values); nodes off the band get `fill` (frozen_out 0). */
int lsm_band_geometry(LsmHandle* h, int what, const void* phi, const void* mask, double scale, double band_width, double fill,
                      void* out0, void* out1, void* out2, void* frozen_out, void* stream);

/* ---- InterpolatedField(ϕ, order)(x) (src/interpolation.jl:117-151,228-260): value, gradient and Hessian of the piecewise
 *      polynomial interpolant of ϕ (Bernstein patch of the cell that holds x, odd orders interpolate, even orders fit
 *      a stencil one node larger; the interpolant `reinitialize!` measures distances to) at `npoints` points.
 *      points: npoints x ndim doubles on the device (point-major); values: npoints; gradients: npoints x ndim or NULL;
 *      hessians: npoints x ndim x ndim (row-major) or NULL.  order in 1..5.  phi's ghost layers are refilled on entry
 *      (dense fields; single device).  Asynchronous on `stream` (NULL = the handle's). */
int lsm_interpolate(LsmHandle* h, void* phi, int order, int64_t npoints, const void* points, void* values, void* gradients,
                    void* hessians, void* stream);

/* ---- NewtonSDF(ϕ; order, upsample, maxiters, xtol, ftol) (src/sdf.jl:57-127): the interface of a private copy of ϕ
 *      sampled once (the first half of reinitialize!), then signed distances at arbitrary points: exact nearest sample,
 *      Newton–Lagrange closest point on the seed cell's patch (further near samples as fall-back seeds),
 *      sign(dot(x - cp, ∇p(cp)))·‖x - cp‖.  phi: ghosts filled (dense) or a prepared band stage input with its mask.
 *      lsm_sdf_eval: points npoints x ndim doubles on the device; distances npoints; closest_points npoints x ndim or NULL;
 *      *nfail := queries whose solve did not converge (their best iterate is returned); NaN where the field has no
 *      interface sample at all.  lsm_sdf_samples: the nsamples x ndim sample points (get_sample_points).  Synchronous. */
typedef struct LsmSdf LsmSdf;
int lsm_sdf_create(LsmHandle* h, void* phi, const void* mask, int order, int upsample, int maxiters, double xtol, double ftol,
                   LsmSdf** out, int64_t* nsamples);
int lsm_sdf_eval(LsmSdf* s, int64_t npoints, const void* points, void* distances, void* closest_points, int64_t* nfail);
int lsm_sdf_samples(LsmSdf* s, void* points_out);
void lsm_sdf_destroy(LsmSdf* s);

/* ---- NarrowBandMeshField (src/meshfield.jl:314-588) on the device.
 *      The band is a byte mask (1 = active node) over the same padded index space as the values
 *      (allocate LsmLayout.total bytes; ghost entries stay 0).  Values stay in the dense padded array.
 *      PeriodicBC is rejected, as in the reference (:339-340).
 *
 *      lsm_band_update      update_band! (:555-588) and everything derived from the new band in one call:
 *                           cut cells of the band (or of the whole grid when from_dense != 0) seed their
 *                           corners, nlayers L1 dilations grow the new band, newly active nodes get the
 *                           affine extrapolant from the OLD band; mask is replaced; tiles := per-tile
 *                           activity flags (size: lsm_band_tile_count; tile = the stage kernel's brick
 *                           with mc planes; on entry the OLD band's flags unless from_dense); then
 *                           lsm_band_halo on the new band.  scratch_a/b: mask-sized.
 *      lsm_band_halo        halo_mask := the in-grid nodes a stencil centred on a band node reads — up to
 *                           3 nodes along each axis, the 3^N box, and the in-grid nodes out-of-grid
 *                           positions resolve to through _getindexbc (:248-260);  halo_list := one 16-byte
 *                           entry per non-band halo node naming its nearest band node
 *                           (_nearest_band_node, :513-530 — a function of the mask alone, so it is found
 *                           once per band).  halo_count: device uint32, receives the number of entries
 *                           wanted; if it exceeds halo_cap the list is truncated — repeat with a larger one.
 *      lsm_band_fill_list   _extrapolate_to_ghost (:481-511) materialised on the halo from halo_list, so that
 *                           stencils read plain entries; follow with lsm_fill_ghosts for the out-of-grid
 *                           layers.  Called on every stage input.
 *      lsm_band_prepare     lsm_band_fill_list + lsm_fill_ghosts in one call (the ghost fill is skipped while the
 *                           band stays a tile away from every face of the grid).
 *      lsm_band_fill        the same by a fresh search for targets & !band nodes; tiles = NULL visits the whole
 *                           grid (scalar getindex path).
 *      lsm_stage_band       lsm_stage restricted to band nodes (tiles without band nodes are skipped)
 *      lsm_compute_cfl_band compute_cfl over active_nodeindices (tiles/mc: optional tile flags, to step over empty tiles)
 *      lsm_band_count       number of active nodes;  lsm_band_missed: a value was requested farther than
 *                           the search radius (6) from the band since the last call (the reference throws);
 *      lsm_band_status      lsm_band_missed and *halo_count in one synchronisation.  It also brings the lengths
 *                           of the compact tile lists lsm_band_update built on the device to the host: from
 *                           then on band kernels launch one block per listed tile instead of one per tile
 *                           (without the call everything still works, over all tiles). */
int lsm_band_tile_count(LsmHandle* h, int mc, int64_t* ntiles);
int lsm_band_update(LsmHandle* h, void* vals, void* mask, int from_dense, int nlayers, void* scratch_a, void* scratch_b,
                    void* halo_mask, void* tiles, int mc, void* halo_list, int64_t halo_cap, void* halo_count);
int lsm_band_halo(LsmHandle* h, const void* vals, const void* mask, void* halo_mask, const void* tiles, int mc,
                  void* halo_list, int64_t halo_cap, void* halo_count);
int lsm_band_retile(LsmHandle* h, const void* mask, void* tiles, int mc);   /* tile flags + lists of a mask changed from outside */
/* The handle keys what it knows about a band (the compact tile lists of `tiles`, the length of halo_list as lsm_band_status
 * last read it, a prefetched dt) by the ADDRESSES of the caller's buffers.  After rewriting mask / tiles / halo_list /
 * halo_count IN PLACE (copy! of another band into the same buffers) call lsm_band_retile + lsm_band_status — which rebuild that
 * state — or at least lsm_band_invalidate, after which the band kernels run over all tiles and take the list's length from
 * the device until the next lsm_band_status.  (The gather of lsm_band_fill_list is bounded by the device counter in any case.)
 * The prefetched dt of a band obeys lsm_cfl_cache's contract: tables rewritten in place -> lsm_cfl_cache(h, 0). */
int lsm_band_invalidate(LsmHandle* h);
int lsm_band_fill_list(LsmHandle* h, void* vals, const void* mask, const void* halo_list, int64_t halo_cap,
                       const void* halo_count);
int lsm_band_prepare(LsmHandle* h, void* vals, const void* mask, const void* halo_list, int64_t halo_cap,
                     const void* halo_count, const void* tiles, int mc);
int lsm_band_fill(LsmHandle* h, void* vals, const void* mask, const void* targets, const void* tiles, int mc);
int lsm_band_count(LsmHandle* h, const void* mask, int64_t* count);
int lsm_band_missed(LsmHandle* h, int* missed);
int lsm_band_status(LsmHandle* h, const void* halo_count, int64_t* count, int* missed);
int lsm_stage_band(LsmHandle* h, const LsmTerm* terms, int nterms, const void* psi, const void* phin, void* out,
                   void* out2, int base_mode, double cdt, double cdt2, double t_stage, const void* mask,
                   const void* tiles, int mc, void* stream);
int lsm_compute_cfl_band(LsmHandle* h, const LsmTerm* terms, int nterms, const void* phi, const void* mask,
                         const void* tiles, int mc, double t, double* dt_out);

/* ---- _advance!(integrator, ϕ::NarrowBandMeshField, buffers, terms, tc, Δt): the reference steps a band with the same
 *      _advance! as a dense field (src/timestepping.jl:128-137,143-164,170-202), its node loop running over
 *      active_nodeindices (src/meshfield.jl:330-333) and its stencil reads outside the band answered by
 *      _extrapolate_to_ghost (:481-511).  Here: per stage lsm_band_prepare on the stage input, then lsm_stage_band; the
 *      integrator's combinations as in lsm_advance_*.  LsmBand names what lsm_band_update maintains for the field (all
 *      caller-owned device buffers).  phi / buf1 / buf2 are padded value arrays; their off-band entries are scratch (buffers
 *      need no initialisation).  The hook runs before each stage with the PREPARED stage input, stream synchronised.  On a
 *      slab with lsm_band_overlap_config the overlap planes of every stage result are refreshed from their owners
 *      (lsm_band_overlap_values).  Follow an accepted step with lsm_band_update (update_band!, src/timestepping.jl:115). */
typedef struct LsmBand {
    void* mask;                   /* byte mask of the active nodes (LsmLayout.total bytes) */
    void* tiles;                  /* per-tile activity flags (lsm_band_tile_count bytes) */
    int32_t mc;                   /* planes per tile along the last dimension */
    int32_t _pad;
    void* halo_list;              /* lsm_band_update / lsm_band_halo: (halo node, nearest band node) entries, 16 bytes each */
    int64_t halo_cap;
    void* halo_count;             /* device uint32 */
} LsmBand;
int lsm_advance_band_fe(LsmHandle* h, const LsmTerm* terms, int nterms, const LsmBand* band, void* phi, void* buf1,
                        double tc, double dt, LsmStageHook hook, void* user);
int lsm_advance_band_rk2(LsmHandle* h, const LsmTerm* terms, int nterms, const LsmBand* band, void* phi, void* buf1, void* buf2,
                         double tc, double dt, LsmStageHook hook, void* user);
int lsm_advance_band_rk3(LsmHandle* h, const LsmTerm* terms, int nterms, const LsmBand* band, void* phi, void* buf1, void* buf2,
                         double tc, double dt, LsmStageHook hook, void* user);

/* ---- reinitialize!(ϕ; order = 3, upsample = 2, maxiters = 20, xtol, ftol) (src/reinitializer.jl:12-42):
 *      every active node (every node when mask == NULL, the band nodes otherwise) is overwritten with
 *      sign(ϕ)·distance to the zero set of the piecewise-polynomial interpolant of ϕ (NewtonSDF, src/sdf.jl:
 *      interface samples by Newton projection, nearest sample as seed, Newton–Lagrange closest point on the
 *      seed cell's Bernstein patch).  phi: ghosts filled (lsm_fill_ghosts) and, with a mask, the band halo
 *      filled (lsm_band_prepare); work: a field-sized scratch array.  order in 1..5.
 *      Out: candidate cells sampled, nodes whose solve did not converge (the reference warns), nodes left
 *      untouched because the field has no interface sample at all.
 *      The handle keeps the call's device buffers and sizes the next call's from them: a call is enqueued whole, synchronises the
 *      stream once at its end, and repeats itself when the band has outgrown the buffers (ϕ is written by the last round only). */
int lsm_reinitialize(LsmHandle* h, void* phi, const void* mask, void* work, int order, int upsample, int maxiters,
                     double xtol, double ftol, int64_t* ncandidate_cells, int64_t* nfail, int64_t* nfar);

/* ---- tuning switches ----
 * Every switch has the name of an environment variable.  The environment is read ONCE per process, when the first handle is
 * created; every handle starts from those values and lsm_set_tuning changes one handle's (tests and A/B measurements flip
 * them between launches).  None of them changes a result beyond what its description says; the whole GPU test suite passes
 * under each.  Unknown names are refused.  (Experiments that were measured and lost have no switch: their record is DESIGN.md.)
 *
 *   name                    default  meaning
 *   LSM_STAGE_TAIL              16   planes per chunk of the graded tail of a dense 3-D stage launch (0: no tail)
 *   LSM_STAGE_TAIL_DYN          25   % spare workgroups of the dynamic tail (0: the static tail)
 *   LSM_STAGE_MC                 0   planes per march chunk in 3-D (0: 64, shorter on small grids)
 *   LSM_STAGE_MC2                0   rows per march chunk in 2-D (0: 8)
 *   LSM_PAIRS                    1   two nodes per thread for dense single-term stages (0: one node per thread)
 *   LSM_STAGE_GENERIC            0   general stage kernels instead of the plain variants (diagnostic: same results)
 *   LSM_XREDIRECT                1   FAST steps: x / y ghosts of copy-type faces (periodic, symmetry, NeumannBC) are read from the
 *                                    node the boundary condition copies instead of being materialised (a copied -0.0 stays -0.0)
 *   LSM_MREDIRECT                1   ... and NeumannBC faces of the march (last) axis by clamping the march at the boundary plane: a FAST
 *                                    step whose other faces are served by the loads launches no ghost fill at all
 *   LSM_GHOST_FULL_DEPTH         0   ghost fills write all LSM_GHOST layers (default: the layers the step's stencils read)
 *   LSM_BAND_BRICKS              1   narrow band: the stage with one lane per band node (0: the tiled stage; bitwise equal)
 *   LSM_BAND_BITS                1   narrow band: update_band! on bit rows (0: the byte-mask kernels; bitwise equal)
 *   LSM_BAND_CFL_PREFETCH        1   narrow band: dt of the next step reduced right behind lsm_band_update
 *   LSM_BAND_BYTES               0   narrow band: byte-mask kernels in 3-D too (the path of wide bands and bands at a face)
 *   LSM_BAND_NO_LISTS            0   narrow band: launches over all tiles instead of the compact tile lists
 *   LSM_STATUS_SPIN              1   lsm_band_status spins on the status kernel's ticket in pinned memory (0: stream synchronise)
 *   LSM_SLAB_OVERLAP             1   slab steps update the interface planes first (read when a communicator is attached)
 *   LSM_COMM_TIMEOUT_MS      60000   how long a rank waits for its peers before LSM_ERR_COMM
 *   LSM_LAYOUT_ALIGN             1   rows of the padded layout start on 64-byte lines (environment only: fixed by lsm_create) */
int lsm_set_tuning(LsmHandle* h, const char* name, int value);
int lsm_get_tuning(const LsmHandle* h, const char* name, int* value);

/* ---- measurement: HIP-event timing of the stage kernels on the handle's stream ----
 * on = 0: off; on = 1: an event pair around every stage launch; on = N > 1: around every N-th launch (an event costs the
 * stream ≈3.7 µs on MI355X: six per RK3 step are 22 µs — 20 % of a 2048² step, 0.6 % of a 512³ one).  lsm_profile_read
 * returns the number of stage launches since the last read and their total time — with a sampling period, the mean of
 * the sampled launches times that number. */
int lsm_profile_enable(LsmHandle* h, int on);
int lsm_profile_read(LsmHandle* h, int64_t* n_stage_launches, double* stage_ms_total);   /* synchronises; resets */

#ifdef __cplusplus
}
#endif
#endif /* LSM_H */
