// lsm_handle.h — the handle behind the C ABI (include/lsm.h), shared by lsm_api.hip and lsm_comm.hip.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "lsm_internal.h"

struct LsmComm;   // lsm_comm.hip: slab communicator (RCCL or in-process), NULL on a single-device handle

struct LsmHandle {
    LsmGrid grid;
    LsmBc bc[LSM_MAX_DIM][2];
    LsmSlab slab;
    int dtype, mode, device;
    LsmLayout lay;
    int nloc[3], goff[3], gn[3];
    double h[3], h2[3], inv_h[3], inv_h2[3], dxmin;
    double w[3][2][LSM_GHOST][8];
    hipStream_t stream;
    bool own_stream;
    double* d_w;         // device copy of w
    signed char* d_ring; // narrow band: distance-sorted offset ring
    int nring;
    int nring_lds;       // ring entries before the first with a component beyond the LDS apron (3)
    int* d_miss;
    unsigned long long* d_count;
    unsigned char* d_work;             // per-tile work flags (narrow band)
    unsigned char* d_tiles_old;        // the old band's tile flags during an update (the new ones are written in place)
    int64_t work_cap;
    const unsigned char* band_mask;    // set for the duration of a *_band call
    const unsigned char* band_tiles;
    int band_mc;
    const int* band_list;              // compact active-tile list for a stage (NULL = flags only)
    unsigned band_nlist;
    // compact tile lists of the band last updated (built on the device by lsm_band_update, lengths read back by
    // lsm_band_status): with them the band kernels launch one block per listed tile instead of one per tile
    int* d_act_list;
    int* d_work_list;
    unsigned* d_lcounts;
    const void* lists_tiles;           // the tile-flag buffer the lists describe
    int lists_mc;
    bool lists_host_valid;
    const void* halo_n_key;            // the device counter whose value lsm_band_status last read (NULL: unknown on the host)
    long long halo_n;
    unsigned nact, nwork, nface;       // list lengths; work tiles on a face of the grid
    lsm::LsmTuning tune;               // tuning switches: the environment's (read once per process) unless lsm_set_tuning changed them
    bool no_lists;                     // = tune.band_no_lists: always launch over all tiles
    bool band_bytes;                   // = tune.band_bytes: byte-mask band kernels in 3-D too
    double* d_partial;   // 2 * MAXB doubles
    int* d_flag;
    double* d_result;    // 16 doubles: [0..1] reductions, [2..6] lsm_band_status, [8..11] Δt of the next step prefetched by lsm_band_update
    double* h_result;    // pinned, 16 doubles
    double* h_result_dev;   // the same page as the device sees it: lsm_band_status's kernel writes its numbers there directly
    unsigned long long status_ticket;   // lsm_band_status: the kernel's last store is the call's ticket ([13]); the host spins on it
    // Δt of a band, prefetched: when the terms of the last lsm_compute_cfl_band depend neither on t nor on a field (constants,
    // ROTATION, SEPARABLE without time factor, Eikonal), lsm_band_update runs their reductions over the NEW band right behind
    // its own kernels and lsm_band_status brings the results home in the read it does anyway — the next lsm_compute_cfl_band
    // with the same terms on the same band launches nothing and waits for nothing
    struct BandCfl {
        LsmTerm terms[LSM_MAX_TERMS];
        int nterms;
        const void *mask, *tiles;
        int mc;
        bool armed, pending, valid;
        int slot[LSM_MAX_TERMS];           // result slot of a node-dependent term, -1 otherwise
        double dt[LSM_MAX_TERMS];
    } band_cfl;
    int* d_pf_flag;      // NaN flags of the prefetched reductions (4)
    std::string err;
    bool cfl_cache_on;
    bool cfl_prefetched;   // set around lsm_compute_cfl by lsm_compute_cfl_band when band_cfl holds this call's values
    std::vector<std::pair<LsmTerm, double>> cfl_cache;   // time-independent analytic coefficients
    struct CflCand { LsmTerm key; long long* d_cand; unsigned count; };
    std::vector<CflCand> cfl_cand;                       // SEPARABLE × g(t): the arg-max candidates are time-independent
    unsigned* d_cand_count;
    // Δt of a ϕ-independent term (constant / catalogued analytic coefficient) is reduced on a stream of its own, with
    // its own scratch: the host gets it without waiting for the stages queued on the main stream, and can queue the
    // next step behind them
    hipStream_t cfl_stream;
    double *c_partial, *c_result, *ch_result;
    int* c_flag;
    std::vector<const void*> cfl_seen;   // coefficient tables known to have landed
    bool prof;
    int prof_every;              // lsm_profile_enable(h, N): every N-th stage launch is timed
    unsigned long long prof_seen;   // stage launches since lsm_profile_enable / lsm_profile_read
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used;
    lsm::ReinitWorkspace* reinit_ws;   // reinitialize!'s device buffers, kept between calls (grow-only)
    LsmComm* comm;       // multi-GPU: attached by lsm_comm_attach_* (slab handles)
    bool yredirect;                // ... and those of dimension 2 (3-D)
    bool mredirect;                // ... and the march axis' NeumannBC faces are served by clamping the march at the boundary plane: no fill is left
    int ghost_depth;               // ghost layers the fills write and the slab exchange sends: LSM_GHOST, or what the step in progress reads (XRedirect)
    int slab_depth_valid;          // slab steps: ghost layers of ϕ (boundary conditions + neighbours' planes) the last step left valid
    unsigned* d_tail_ctr;          // ring of LSM_TAIL_SLOTS ticket counters of the dynamic tail (each launch resets its own)
    unsigned tail_ticket;          // host: launches that took a slot so far
    bool xredirect;                // set around the stages of a whole-grid lsm_advance_*: x ghosts are resolved by the stage kernel's loads
    unsigned long long* d_stamp;   // diagnostic build (-DLSM_STAMP): 8192 x {Δs_memtime, Δs_memrealtime, start, end} of the stage kernel's workgroups
};

// shared helpers (lsm_api.hip)
int lsm_fail(LsmHandle* h, int code, const std::string& msg);
// lsm_comm.hip: the attached communicator wants boundary-first stages (lsm_comm_set_overlap; LSM_SLAB_OVERLAP=0 at attach time)
bool lsm_comm_overlap(const LsmHandle* h);
int lsm_host_sync(LsmHandle* h, const char* what);   // host wait for h->stream that an RCCL peer's silence cannot hang (lsm_comm.hip)
int lsm_comm_band_overlap(const LsmHandle* h);   // overlap depth declared by lsm_band_overlap_config (0 = none)
