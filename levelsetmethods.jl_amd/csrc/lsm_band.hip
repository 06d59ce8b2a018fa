// lsm_band.hip — device counterpart of NarrowBandMeshField (src/meshfield.jl:314-588):
// the active node set as a byte mask over the dense padded layout, its topological rebuild
// (update_band!), and the affine ghost extrapolation that feeds stencils reaching past the band.
//
// Representation.  Values stay in the dense padded array (288 GB of HBM make the dense backing
// affordable; what the band saves is ARITHMETIC and TRAFFIC: every kernel here and the stage kernel
// work tile by tile — the stage kernel's 32×8×MC bricks — and skip tiles that cannot contain band
// nodes).  `mask[q] != 0` marks band nodes, same index space as the values (mask ghosts are always 0:
// an out-of-grid index is never a band node).
//
//  * update_band!  (:555-588): cut cells (all 2^N corners in the band, values straddling 0) seed
//    their corners; nlayers von-Neumann dilations grow exactly the L¹ ball of the reference's
//    `grow` offsets; newly active nodes get the extrapolated value from the OLD band.
//  * ϕ[I] for an in-grid non-band node (:481-511): nearest band node by the reference's offset
//    ring (sorted by |off|², ties in column-major order), per-axis one-sided slope from the band
//    neighbours of that node (+ side first), sign-preserving clamp.  Materialised ("band halo")
//    for every non-band node within Chebyshev distance 3 of the band before each stage, so that
//    stencils and the boundary-condition fill read plain array entries.
//  * work tiles: a step moves the interface by less than a cell, so the new band, its halo and every
//    node the rebuild reads lie in the old band's tiles or their 26 neighbours; `work` flags those.
// Built with -ffp-contract=off.
#include <cstdlib>
#include "lsm_internal.h"

#define LSM_BAND_LDS 40960   // bytes of LDS for a tile + apron of byte flags

namespace lsm {

// this block's tile: taken from the compact tile list when there is one (grid = list length)
#define LSM_TILE_ID(a) ((a).list ? (unsigned)(a).list[blockIdx.x] : (unsigned)blockIdx.x)
// iterate the nodes of this block's tile; returns false if the tile is skipped
#define LSM_TILE_PROLOGUE(a)                                                                         \
    const unsigned tile = LSM_TILE_ID(a);                                                            \
    if ((a).work && !(a).work[tile]) return;                                                         \
    const int bx_ = tile % (a).nbx, by_ = (tile / (a).nbx) % (a).nby, bm_ = tile / ((a).nbx * (a).nby); \
    const int x0_ = bx_ * (a).tx, y0_ = (a).ndim == 3 ? by_ * (a).ty : 0, m0_ = bm_ * (a).tm;        \
    const int ex_ = (a).tx, ey_ = (a).ndim == 3 ? (a).ty : 1, em_ = (a).ndim >= 2 ? (a).tm : 1;      \
    const int nx_ = (a).n[0], ny_ = (a).ndim == 3 ? (a).n[1] : 1, nm_ = (a).ndim >= 2 ? (a).n[(a).ndim - 1] : 1; \
    const long long sy_ = (a).ndim == 3 ? (a).s1 : 0, sm_ = (a).ndim == 3 ? (a).s2 : ((a).ndim == 2 ? (a).s1 : 0);
#define LSM_TILE_FOR(a, X, Y, M, Q)                                                                  \
    for (int e_ = threadIdx.x; e_ < ex_ * ey_ * em_; e_ += blockDim.x)                               \
        if (const int X = x0_ + e_ % ex_, Y = y0_ + (e_ / ex_) % ey_, M = m0_ + e_ / (ex_ * ey_);     \
            X < nx_ && Y < ny_ && M < nm_)                                                           \
            if (const long long Q = (a).origin + X + Y * sy_ + M * sm_; true)

// LDS box = this block's tile plus an apron of `ap` nodes per side, one byte per node, index
// lx + bx·(ly + by·lm) in tile-local coordinates (apron included); positions outside the grid read as 0.
struct Box { int ap, bx, by, bm, nel; float rbx, rby; };
__device__ __forceinline__ Box make_box(const BandArgs& a, int ex, int ey, int em, int ap) {
    Box b;
    b.ap = ap; b.bx = ex + 2 * ap; b.by = a.ndim == 3 ? ey + 2 * ap : 1; b.bm = a.ndim >= 2 ? em + 2 * ap : 1;
    b.nel = b.bx * b.by * b.bm;
    b.rbx = 1.0f / (float)b.bx; b.rby = 1.0f / (float)b.by;
    return b;
}
// e -> (lx, ly, lm) without integer division (e < 2^16: a float quotient is off by at most one)
__device__ __forceinline__ void box_coords(const Box& b, int e, int& lx, int& ly, int& lm) {
    int t = (int)((float)e * b.rbx), r = e - t * b.bx;
    if (r >= b.bx) { ++t; r -= b.bx; } else if (r < 0) { --t; r += b.bx; }
    int u = (int)((float)t * b.rby), r2 = t - u * b.by;
    if (r2 >= b.by) { ++u; r2 -= b.by; } else if (r2 < 0) { --u; r2 += b.by; }
    lx = r; ly = r2; lm = u;
}
// Staging is latency-bound (a byte per lane per load), so every thread issues K independent loads before
// it stores anything: first(q) for all K, then second(q, first) for all K (which may load again).
template <int K, typename F1, typename F2>
__device__ __forceinline__ void stage_box(const BandArgs& a, const Box& b, int x0, int y0, int m0, int nx, int ny, int nm,
                                          long long sy, long long sm, unsigned char* lds, F1 first, F2 second) {
    const int apy = a.ndim == 3 ? b.ap : 0, apm = a.ndim >= 2 ? b.ap : 0;
    for (int e0 = threadIdx.x; e0 < b.nel; e0 += blockDim.x * K) {
        long long q[K];
        bool ok[K];
        unsigned char u[K], v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = e0 + k * blockDim.x;
            int lx, ly, lm;
            box_coords(b, e < b.nel ? e : 0, lx, ly, lm);
            const int gx = x0 - b.ap + lx, gy = y0 - apy + ly, gm = m0 - apm + lm;
            ok[k] = e < b.nel && gx >= 0 && gx < nx && gy >= 0 && gy < ny && gm >= 0 && gm < nm;
            q[k] = ok[k] ? a.origin + gx + gy * sy + gm * sm : a.origin;   // always a valid address: loads need no branch
        }
#pragma unroll
        for (int k = 0; k < K; ++k) u[k] = first(q[k]);
#pragma unroll
        for (int k = 0; k < K; ++k) u[k] = ok[k] ? u[k] : (unsigned char)0;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = second(q[k], u[k]);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (e0 + k * blockDim.x < b.nel) lds[e0 + k * blockDim.x] = v[k];
    }
}

// update_band! (src/meshfield.jl:555-588) for one tile, entirely in LDS: node flags (in the old band,
// value <= 0, value >= 0) on the tile + apron nl+1 -> cut cells -> seeds (corners of cut cells) -> nl
// von-Neumann dilations -> new band on the tile, and the tile's activity flag.
__global__ void __launch_bounds__(256) band_grow_kernel(BandArgs a, const void* v, const unsigned char* old_mask, int nl,
                                                        unsigned char* new_mask, unsigned char* tiles) {
    if (a.work && !a.work[LSM_TILE_ID(a)]) {
        if (threadIdx.x == 0) tiles[LSM_TILE_ID(a)] = 0;
        return;
    }
    LSM_TILE_PROLOGUE(a)
    extern __shared__ unsigned char f[];
    const Box b = make_box(a, ex_, ey_, em_, nl + 1);
    const int nel = b.nel;
    stage_box<8>(a, b, x0_, y0_, m0_, nx_, ny_, nm_, sy_, sm_, f,
                 [&](long long q) -> unsigned char { return old_mask ? old_mask[q] : (unsigned char)1; },
                 [&](long long q, unsigned char inband) -> unsigned char {
                     const double x = ld_val(v, inband ? q : a.origin, a.f32);      // unconditional load from a valid address
                     return inband ? (unsigned char)(1 | (x <= 0.0 ? 2 : 0) | (x >= 0.0 ? 4 : 0)) : (unsigned char)0;
                 });
    __syncthreads();
    const int sby = b.bx, sbm = b.bx * b.by;
    const int cy = a.ndim == 3 ? 1 : 0, cm = a.ndim >= 2 ? 1 : 0;   // cell extent per tile direction
    // cut cells: lower corner l, all corners in the old band, values straddling 0 (bit 3)
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        int lx, ly, lm;
        box_coords(b, e, lx, ly, lm);
        if (lx + 1 >= b.bx || ly + cy >= b.by || lm + cm >= b.bm) continue;
        unsigned all = 1, any = 0;
        for (int dm = 0; dm <= cm; ++dm)
            for (int dy = 0; dy <= cy; ++dy)
                for (int dx = 0; dx <= 1; ++dx) {
                    const unsigned c = f[e + dx + dy * sby + dm * sbm];
                    all &= c; any |= c;
                }
        if ((all & 1) && (any & 2) && (any & 4)) f[e] |= 8;
    }
    __syncthreads();
    // seeds: every corner of a cut cell (bit 4)
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        int lx, ly, lm;
        box_coords(b, e, lx, ly, lm);
        unsigned sd = 0;
        for (int dm = 0; dm <= cm; ++dm)
            for (int dy = 0; dy <= cy; ++dy)
                for (int dx = 0; dx <= 1; ++dx) {
                    if (lx - dx < 0 || ly - dy < 0 || lm - dm < 0) continue;
                    sd |= f[e - dx - dy * sby - dm * sbm] & 8;
                }
        if (sd) f[e] |= 16;
    }
    __syncthreads();
    // nl dilation steps, ping-pong between bits 4 and 5
    unsigned cur = 16, nxt = 32;
    for (int it = 0; it < nl; ++it) {
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            int lx, ly, lm;
            box_coords(b, e, lx, ly, lm);
            unsigned r = f[e];
            if (lx > 0) r |= f[e - 1];
            if (lx + 1 < b.bx) r |= f[e + 1];
            if (cy) {
                if (ly > 0) r |= f[e - sby];
                if (ly + 1 < b.by) r |= f[e + sby];
            }
            if (cm) {
                if (lm > 0) r |= f[e - sbm];
                if (lm + 1 < b.bm) r |= f[e + sbm];
            }
            f[e] = (unsigned char)((f[e] & ~nxt) | ((r & cur) ? nxt : 0));
        }
        __syncthreads();
        const unsigned t = cur; cur = nxt; nxt = t;
    }
    int any = 0;
    LSM_TILE_FOR(a, x, y, m, q) {
        const int lx = x - x0_ + b.ap, ly = a.ndim == 3 ? y - y0_ + b.ap : 0, lm = a.ndim >= 2 ? m - m0_ + b.ap : 0;
        const unsigned char on = (f[lx + b.bx * (ly + b.by * lm)] & cur) ? 1 : 0;
        new_mask[q] = on;
        any |= on;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) tiles[tile] = any ? 1 : 0;
}

__global__ void __launch_bounds__(256) band_cut_kernel(BandArgs a, const void* v, const unsigned char* old_mask,
                                                       unsigned char* seed) {
    LSM_TILE_PROLOGUE(a)
    const int nc = 1 << a.ndim;
    LSM_TILE_FOR(a, x, y, m, q) {
        // the cell with lower corner I must lie inside the grid
        if (x + 1 >= nx_ || (a.ndim == 3 && y + 1 >= ny_) || (a.ndim >= 2 && m + 1 >= nm_)) continue;
        double vmin = __builtin_inf(), vmax = -__builtin_inf();
        bool ok = true;
        for (int c = 0; c < nc; ++c) {
            // corner offsets in the field's own dimensions: bit 0 -> dim 1, bit 1 -> dim 2, bit 2 -> dim 3
            const long long qc = q + (c & 1) + ((c >> 1) & 1) * a.s1 + ((c >> 2) & 1) * a.s2;
            if (old_mask && !old_mask[qc]) { ok = false; break; }
            const double xv = ld_val(v, qc, a.f32);
            vmin = xv < vmin ? xv : vmin;
            vmax = xv > vmax ? xv : vmax;
        }
        if (!(ok && vmin <= 0.0 && 0.0 <= vmax)) continue;
        for (int c = 0; c < nc; ++c) seed[q + (c & 1) + ((c >> 1) & 1) * a.s1 + ((c >> 2) & 1) * a.s2] = 1;
    }
}

// one von-Neumann (L¹) dilation step on the interior; ghosts of `in` are 0
__global__ void __launch_bounds__(256) band_dilate_kernel(BandArgs a, const unsigned char* in, unsigned char* out) {
    LSM_TILE_PROLOGUE(a)
    LSM_TILE_FOR(a, x, y, m, q) {
        unsigned char r = in[q] | in[q - 1] | in[q + 1];
        if (a.ndim > 1) r |= in[q - a.s1] | in[q + a.s1];
        if (a.ndim > 2) r |= in[q - a.s2] | in[q + a.s2];
        out[q] = r ? 1 : 0;
    }
}

// out := in on the visited tiles; `zero` (may be NULL) is cleared on the same tiles in the same pass (the halo mask of the update)
__global__ void __launch_bounds__(256) band_copy_kernel(BandArgs a, const unsigned char* in, unsigned char* out, unsigned char* zero) {
    LSM_TILE_PROLOGUE(a)
    if (ex_ % 8 == 0 && x0_ + ex_ <= nx_) {
        // whole x-rows of the tile inside the grid: 8 mask bytes per access (unaligned), one access per thread and round
        // instead of eight rounds of byte copies (the kernel is bound by the latency of its rounds)
        typedef unsigned long long w8 __attribute__((aligned(1)));
        const int ppr = ex_ / 8;
        for (int e_ = threadIdx.x; e_ < ppr * ey_ * em_; e_ += blockDim.x) {
            const int x = x0_ + 8 * (e_ % ppr), y = y0_ + (e_ / ppr) % ey_, m = m0_ + e_ / (ppr * ey_);
            if (y >= ny_ || m >= nm_) continue;
            const long long q = a.origin + x + y * sy_ + m * sm_;
            *reinterpret_cast<w8*>(out + q) = *reinterpret_cast<const w8*>(in + q);
            if (zero) *reinterpret_cast<w8*>(zero + q) = 0ull;
        }
        return;
    }
    LSM_TILE_FOR(a, x, y, m, q) { out[q] = in[q]; if (zero) zero[q] = 0; }
}

// dst := src at the band nodes of the visited tiles (the copy!(ϕ, dst) that ends a ForwardEuler step on a band: only band entries matter)
__global__ void __launch_bounds__(256) band_copy_values_kernel(BandArgs a, const unsigned char* mask, const void* src, void* dst) {
    LSM_TILE_PROLOGUE(a)
    LSM_TILE_FOR(a, x, y, m, q)
        if (mask[q]) st_val(dst, q, a.f32, ld_val(src, q, a.f32));
}

// zero a byte mask on the visited tiles only (the halo mask of a band update: nothing reads it outside the work tiles)
__global__ void __launch_bounds__(256) band_zero_kernel(BandArgs a, unsigned char* out) {
    LSM_TILE_PROLOGUE(a)
    if (ex_ % 8 == 0 && x0_ + ex_ <= nx_) {
        typedef unsigned long long w8 __attribute__((aligned(1)));
        const int ppr = ex_ / 8;
        for (int e_ = threadIdx.x; e_ < ppr * ey_ * em_; e_ += blockDim.x) {
            const int x = x0_ + 8 * (e_ % ppr), y = y0_ + (e_ / ppr) % ey_, m = m0_ + e_ / (ppr * ey_);
            if (y >= ny_ || m >= nm_) continue;
            *reinterpret_cast<w8*>(out + (a.origin + x + y * sy_ + m * sm_)) = 0ull;
        }
        return;
    }
    LSM_TILE_FOR(a, x, y, m, q) out[q] = 0;
}

// the numbers lsm_band_status hands to the host, gathered into one pinned-memory copy: {halo entries wanted, search misses,
// active tiles, work tiles, face tiles}
// `out` is the host's pinned page: the last word written, behind a system-scope fence, is the call's ticket — the host spins on it
// instead of paying a stream synchronisation's wake-up (tens of microseconds of idle GPU per step).
// pf.n > 0: Δt of the next step, prefetched (LsmHandle::BandCfl) — this kernel is also the second stage of those reductions
// (cfl_final_kernel's arithmetic on the partial maxima of cfl_band_list_kernel): one small launch less per step.
__global__ void __launch_bounds__(256) band_status_kernel(const unsigned* halo_count, const int* miss, const unsigned* lcounts, BandStatusCfl pf, double* out,
                                                          double ticket) {
    __shared__ double wmax[4];
    for (int s = 0; s < pf.n; ++s) {
        double best = 0.0;
        for (int i = threadIdx.x; i < pf.npartials; i += 256) { const double v = pf.partial[s * pf.npartials + i]; best = v > best ? v : best; }
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(best, off, 64); best = o > best ? o : best; }
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) best = wmax[w] > best ? wmax[w] : best;
            const double cfl = pf.kind[s] == LSM_TERM_CURVATURE ? (pf.dxmin * pf.dxmin) / (2 * best) : 1 / best;
            out[6 + s] = pf.nanflag[s] ? __builtin_nan("") : cfl;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (double)halo_count[0];
        out[1] = (double)miss[0];
        out[2] = lcounts ? (double)lcounts[0] : 0.0;
        out[3] = lcounts ? (double)lcounts[1] : 0.0;
        out[4] = lcounts ? (double)lcounts[2] : 0.0;
        out[5] = 0.0;
        __threadfence_system();
        *reinterpret_cast<volatile double*>(out + 11) = ticket;
    }
}

// _nearest_band_node continued in global memory: ring entries [r0, nring) in order, bounds-checked
__device__ __forceinline__ bool ring_scan_global(const BandArgs& a, const unsigned char* src_mask, const signed char* ring, int r0,
                                                 int nring, const int I[3], int P[3]) {
    for (int r = r0; r < nring; ++r) {
        const int p0 = I[0] + ring[3 * r], p1 = I[1] + ring[3 * r + 1], p2 = I[2] + ring[3 * r + 2];
        if (p0 < 0 || p0 >= a.n[0] || p1 < 0 || p1 >= a.n[1] || p2 < 0 || p2 >= a.n[2]) continue;
        if (src_mask[a.origin + p0 + p1 * a.s1 + p2 * a.s2]) { P[0] = p0; P[1] = p1; P[2] = p2; return true; }
    }
    return false;
}

// what follows the search for node I (padded index q) with nearest band node P: record the pair and / or
// write the affine extrapolant (_extrapolate_to_ghost, src/meshfield.jl:494-511)
__device__ __forceinline__ void finish_node(const BandArgs& a, long long q, const int I[3], const int P[3], bool found,
                                            const unsigned char* src_mask, const void* src, void* dst, int* miss, BandEntry* list,
                                            unsigned* list_count, unsigned list_cap) {
    if (!found) { atomicOr(miss, 1); return; }   // the reference throws: farther than the search radius
    const long long qp = a.origin + P[0] + P[1] * a.s1 + P[2] * a.s2;
    if (list) {
        // one atomic per wave: the lanes that reach this point together take consecutive slots
        const unsigned long long bal = __ballot(1);
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)bal) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(list_count, (unsigned)__popcll(bal));
        const unsigned k = __shfl(base, leader, 64) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (k < list_cap) {
            BandEntry e;
            e.q = q; e.rel = (int)(qp - q);
            e.d[0] = (signed char)(I[0] - P[0]); e.d[1] = (signed char)(I[1] - P[1]); e.d[2] = (signed char)(I[2] - P[2]); e.d[3] = 0;
            list[k] = e;
        }
    }
    if (!dst) return;
    const double phiP = ld_val(src, qp, a.f32);
    double val = phiP;
    for (int d = 0; d < a.ndim; ++d) {
        const int delta = I[d] - P[d];
        if (delta == 0) continue;
        const long long sd = d == 0 ? 1 : (d == 1 ? a.s1 : a.s2);
        double slope = 0.0;                                   // _axis_slope: + neighbour first, then -
        if (P[d] + 1 < a.n[d] && src_mask[qp + sd]) slope = ld_val(src, qp + sd, a.f32) - phiP;
        else if (P[d] - 1 >= 0 && src_mask[qp - sd]) slope = phiP - ld_val(src, qp - sd, a.f32);
        val += (double)delta * slope;
    }
    // never let extrapolation invent a sign change far from the band
    const double sv = val > 0 ? 1.0 : (val < 0 ? -1.0 : val), sp = phiP > 0 ? 1.0 : (phiP < 0 ? -1.0 : phiP);
    st_val(dst, q, a.f32, (phiP == 0.0 || sv == sp) ? val : phiP);
}

// _extrapolate_to_ghost (src/meshfield.jl:494-511).
//  * target != NULL: for every node with target[q] && !src_mask[q], value written to dst.
//  * halo   != NULL (band-halo mode): the targets are the nodes a stencil centred on a band node reads — up
//    to LSM_GHOST nodes along each axis and the 3^N box, found here from the LDS copy of the mask — plus the
//    nodes already marked in halo[] (boundary-condition sources, band_halo_bc_kernel); halo[] becomes the
//    full halo mask (band included) and the (node, nearest band node) pairs are appended to `list` for
//    band_apply_kernel: the nearest node depends on the mask only, so the search runs once per band update
//    and every stage input is then filled by a plain gather.
// The ring search probes mask bytes in order of distance; the tile's mask with an apron of RL nodes is staged
// in LDS (out-of-grid entries read as 0), which serves the first nring_lds ring entries (those with all
// components <= RL); the rare longer searches continue in global memory.
constexpr int RL = 3;
__global__ void __launch_bounds__(256) band_extrapolate_kernel(BandArgs a, const unsigned char* target, unsigned char* halo,
                                                               const unsigned char* src_mask, const signed char* ring, int nring,
                                                               int nring_lds, const void* src, void* dst, int* miss,
                                                               BandEntry* list, unsigned* list_count, unsigned list_cap) {
    LSM_TILE_PROLOGUE(a)
    if (target) {   // nothing to do in most tiles: skip them before staging anything
        int any = 0;
        LSM_TILE_FOR(a, x, y, m, q) any |= (target[q] && !src_mask[q]) ? 1 : 0;
        if (!__syncthreads_or(any)) return;
    }
    extern __shared__ unsigned char lmask[];
    const Box b = make_box(a, ex_, ey_, em_, RL);
    const bool staged = b.nel <= LSM_BAND_LDS;     // the launcher sized the dynamic LDS accordingly
    if (staged) {
        stage_box<8>(a, b, x0_, y0_, m0_, nx_, ny_, nm_, sy_, sm_, lmask, [&](long long q) -> unsigned char { return src_mask[q]; },
                     [&](long long, unsigned char u) -> unsigned char { return u; });
        __syncthreads();
    }
    const int cy = a.ndim == 3 ? 1 : 0, cm = a.ndim >= 2 ? 1 : 0;
    const int sby = cy ? b.bx : 0, sbm = cm ? b.bx * b.by : 0;      // LDS strides of the tile directions
    LSM_TILE_FOR(a, x, y, m, q) {
        const int lx = x - x0_ + RL, ly = cy ? y - y0_ + RL : 0, lm = cm ? m - m0_ + RL : 0;
        const int c = lx + b.bx * (ly + b.by * lm);
        if (halo) {
            unsigned char t;
            if (staged) {
                if (lmask[c]) { halo[q] = 1; continue; }
                t = halo[q];
                if (!t) {
                    for (int k = -LSM_GHOST; k <= LSM_GHOST; ++k) t |= lmask[c + k] | lmask[c + k * sby] | lmask[c + k * sbm];
                    for (int dm = -cm; dm <= cm; ++dm)
                        for (int dy = -cy; dy <= cy; ++dy) {
                            const int cc = c + dy * sby + dm * sbm;
                            t |= lmask[cc - 1] | lmask[cc] | lmask[cc + 1];
                        }
                }
            } else {
                // mask at tile-direction offset (dx, dy, dm) from this node
                auto at = [&](int dx, int dy, int dm) -> unsigned char {
                    const int X = x + dx, Y = y + dy, M = m + dm;
                    if (X < 0 || X >= nx_ || Y < 0 || Y >= ny_ || M < 0 || M >= nm_) return 0;
                    return src_mask[a.origin + X + Y * sy_ + M * sm_];
                };
                if (at(0, 0, 0)) { halo[q] = 1; continue; }
                t = halo[q];
                if (!t) {
                    for (int k = -LSM_GHOST; k <= LSM_GHOST; ++k) {
                        t |= at(k, 0, 0);
                        if (cy) t |= at(0, k, 0);
                        if (cm) t |= at(0, 0, k);
                    }
                    for (int dm = -cm; dm <= cm; ++dm)
                        for (int dy = -cy; dy <= cy; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) t |= at(dx, dy, dm);
                }
            }
            if (!t) continue;
            halo[q] = 1;
        } else if (!target[q] || src_mask[q]) {
            continue;
        }
        int I[3] = {x, 0, 0};
        if (a.ndim == 2) I[1] = m;
        if (a.ndim == 3) { I[1] = y; I[2] = m; }
        // _nearest_band_node: first hit along the distance-sorted offset ring (offsets in field dimensions)
        int P[3] = {0, 0, 0};
        bool found = false;
        int r = 0;
        if (staged) {
            for (; r < nring_lds; ++r) {
                const int o0 = ring[3 * r], o1 = ring[3 * r + 1], o2 = ring[3 * r + 2];
                const int oy = a.ndim == 3 ? o1 : 0, om = a.ndim == 3 ? o2 : (a.ndim == 2 ? o1 : 0);
                if (lmask[c + o0 + oy * sby + om * sbm]) {
                    P[0] = I[0] + o0; P[1] = I[1] + o1; P[2] = I[2] + o2; found = true; break;
                }
            }
        }
        if (!found) found = ring_scan_global(a, src_mask, ring, r, nring, I, P);
        finish_node(a, q, I, P, found, src_mask, src, dst, miss, list, list_count, list_cap);
    }
}

// ------------------------------------------------------------------------------------------------
// 3-D fast path (32×8×mc tiles): the same two kernels with the masks held as 64-bit row words (bit lx of
// row (ly, lm) = node lx of that x-line of the tile + apron, built with wave ballots), so that cut
// cells, dilations, the halo test and the nearest-node search handle a whole x-line per operation.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
#ifndef LSM_BAND_GK
#define LSM_BAND_GK 16
#endif
constexpr int GK = LSM_BAND_GK;   // x-line loads a wave keeps in flight while staging

// Stage one flag word per x-line of the box: a wave takes a contiguous run of rows, lanes along x.  The row
// bookkeeping (coordinates, validity, base address) is computed for 64 rows at once, one row per LANE, and
// broadcast row by row with v_readlane; the ballots are gathered back one row per lane and stored with a single
// LDS write per 64 rows.  (A version that did the row arithmetic on the scalar unit was bound by it: ~35 scalar
// instructions per row against ~8 here.)
// load(q, ok, rowB) -> flag bits of this lane's node (bit wd goes to out[wd][row]); q is always a valid address;
// rowB = Bin[row] (the band word of the row, for the second pass) or 0.
__device__ __forceinline__ u64 readlane64(u64 v, int k) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), k);
    return ((u64)hi << 32) | lo;
}
template <int U, int NW, typename F>
__device__ __forceinline__ void stage_rows(const BandArgs& a, int ap, int bx, int by, int bm, int x0, int y0, int m0, u64* const (&out)[NW],
                                           const u64* Bin, F load) {
    const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrows = by * bm;
    const int rpw = (nrows + nwv - 1) / nwv;
    const int rbeg = wv * rpw, rend = rbeg + rpw < nrows ? rbeg + rpw : nrows;
    const int gx = x0 - ap + lane;
    const bool xok = (lane < bx) & ((unsigned)gx < (unsigned)a.n[0]);
    const long long xoff = xok ? gx : 0;
    const float rby = 1.0f / (float)by;
    for (int c0 = rbeg; c0 < rend; c0 += 64) {
        const int cnt = rend - c0 < 64 ? rend - c0 : 64;      // wave-uniform
        // lane j prepares row c0 + j
        const int row = c0 + lane < nrows ? c0 + lane : nrows - 1;
        int lm = (int)((float)row * rby), ly = row - lm * by;
        if (ly >= by) { ++lm; ly -= by; } else if (ly < 0) { --lm; ly += by; }
        const int gy = y0 - ap + ly, gm = m0 - ap + lm;
        const int rok = (lane < cnt) & ((unsigned)gy < (unsigned)a.n[1]) & ((unsigned)gm < (unsigned)a.n[2]);
        const u64 base = (u64)(a.origin + (rok ? gy * a.s1 + gm * a.s2 : 0ll));
        const u64 bin = Bin ? Bin[row] : 0ull;
        u64 word[NW];
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) word[wd] = 0;
        for (int k0 = 0; k0 < cnt; k0 += U) {
            unsigned f[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                      // U independent loads in flight
                const int k = k0 + u < cnt ? k0 + u : cnt - 1;
                const long long q = (long long)readlane64(base, k) + xoff;
                const bool ok = xok & (__builtin_amdgcn_readlane(rok, k) != 0) & (k0 + u < cnt);
                const u64 rowB = readlane64(bin, k);
                // second pass: an x-line without a band node has nothing to fetch (wave-uniform: the row word is a scalar) —
                // about half of the rows of a work tile's box; the kernel is bound by the latency of these row loads
                f[u] = (Bin && rowB == 0) ? 0u : load(q, ok, rowB);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int wd = 0; wd < NW; ++wd) {
                    const u64 bal = __ballot((f[u] >> wd) & 1u);
                    if (lane == k0 + u) word[wd] = bal;       // v_cndmask, no exec juggling
                }
        }
        if (lane < cnt)
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) out[wd][c0 + lane] = word[wd];
    }
}

// The row words of a BYTE mask, eight mask bytes per lane: a lane takes piece p (bytes 8p .. 8p+7) of row r, so that one
// wave instruction covers 64 / ceil(bx/8) rows (12 for the 40-node rows of a 32-wide tile with apron 4) instead of one —
// the kernels that stage masks are bound by the latency of these loads, not by their bytes.  Unaligned 8-byte loads
// (rows start at arbitrary byte offsets); bytes outside [0, n0) of the row or beyond the box are dropped; a row outside
// the grid is empty.  mask == NULL: every in-grid node counts as set (update_band! from a dense field).
typedef u64 u64_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ void stage_mask_rows8(const BandArgs& a, int ap, int bx, int by, int bm, int x0, int y0, int m0, u64* out,
                                                 const unsigned char* mask) {
    const int nrows = by * bm;
    for (int t = threadIdx.x; t < nrows; t += blockDim.x) out[t] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int ppr = (bx + 7) >> 3, rpi = 64 / ppr;
    const int p = lane % ppr, rl = lane / ppr;
    const int gx0 = x0 - ap + 8 * p;
    // bytes [lo, hi) of this lane's piece lie inside the row of the grid and inside the box
    int lo = gx0 < 0 ? -gx0 : 0, hi = 8;
    if (a.n[0] - gx0 < hi) hi = a.n[0] - gx0;
    if (bx - 8 * p < hi) hi = bx - 8 * p;
    const u64 keep = hi > lo ? ((~0ull >> (64 - 8 * (hi - lo))) << (8 * lo)) : 0ull;
    const float rby = 1.0f / (float)by;
    unsigned char* ob = reinterpret_cast<unsigned char*>(out);
    for (int r0 = wv * rpi; r0 < nrows; r0 += nwv * rpi) {
        const int row = r0 + rl;
        if (rl >= rpi || row >= nrows || keep == 0) continue;
        int lm = (int)((float)row * rby), ly = row - lm * by;
        if (ly >= by) { ++lm; ly -= by; } else if (ly < 0) { --lm; ly += by; }
        const int gy = y0 - ap + ly, gm = m0 - ap + lm;
        if ((unsigned)gy >= (unsigned)a.n[1] || (unsigned)gm >= (unsigned)a.n[2]) continue;
        u64 w = 0x0101010101010101ull;
        if (mask) w = *reinterpret_cast<const u64_unaligned*>(mask + (a.origin + gy * a.s1 + gm * a.s2 + gx0));
        w &= keep;
        // non-zero bytes -> one bit each, gathered into the low byte
        u64 t = ((w & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | w;
        t = (t >> 7) & 0x0101010101010101ull;
        ob[row * 8 + p] = (unsigned char)((t * 0x0102040810204080ull) >> 56);
    }
}

// update_band! for one 32×8×mc tile (see band_grow_kernel)
__global__ void __launch_bounds__(256) band_grow3_kernel(BandArgs a, const void* v, const unsigned char* old_mask, int nl,
                                                         unsigned char* new_mask, unsigned char* tiles) {
    const unsigned tile = LSM_TILE_ID(a);
    if (a.work && !a.work[tile]) {
        if (threadIdx.x == 0) tiles[tile] = 0;
        return;
    }
    const int x0 = (tile % a.nbx) * a.tx, y0 = ((tile / a.nbx) % a.nby) * a.ty, m0 = (tile / (a.nbx * a.nby)) * a.tm;
    const int ap = nl + 1, bx = a.tx + 2 * ap, by = a.ty + 2 * ap, bm = a.tm + 2 * ap, nrows = by * bm;
    extern __shared__ u64 w3[];
    u64 *B = w3, *LE = w3 + nrows, *GE = w3 + 2 * nrows, *S0 = w3 + 3 * nrows, *S1 = w3 + 4 * nrows;
    // pass 1: the old band's row words; tiles whose box holds no band node have no cut cell -> empty
    stage_mask_rows8(a, ap, bx, by, bm, x0, y0, m0, B, old_mask);
    __syncthreads();
    // tile is 32 × 8 × tm; the block's 256-thread groups take planes pg, pg + npg, ...
    const int tx_ = threadIdx.x & 31, ty_ = (threadIdx.x >> 5) & 7, pg = threadIdx.x >> 8, npg = blockDim.x >> 8;
    const int x = x0 + tx_, y = y0 + ty_;
    {
        int anyb = 0;
        for (int t = threadIdx.x; t < nrows; t += blockDim.x) anyb |= B[t] ? 1 : 0;
        if (!__syncthreads_or(anyb)) {
            if (x < a.n[0] && y < a.n[1])
                for (int i = pg; i < a.tm && m0 + i < a.n[2]; i += npg) new_mask[a.origin + x + y * a.s1 + (m0 + i) * a.s2] = 0;
            if (threadIdx.x == 0) tiles[tile] = 0;
            return;
        }
    }
    // pass 2: values of the band nodes -> (<= 0) and (>= 0) row words
    u64* const out2[2] = {LE, GE};
    stage_rows<GK, 2>(a, ap, bx, by, bm, x0, y0, m0, out2, B, [&](long long q, bool, u64 rowB) -> unsigned {
        const bool on = (rowB >> (threadIdx.x & 63)) & 1ull;
        const double xv = ld_val(v, on ? q : a.origin, a.f32);
        return on ? ((xv <= 0.0 ? 1u : 0u) | (xv >= 0.0 ? 2u : 0u)) : 0u;
    });
    __syncthreads();
    const float rby = 1.0f / (float)by;
    auto rowcoords = [&](int t, int& ly, int& lm) {
        lm = (int)((float)t * rby); ly = t - lm * by;
        if (ly >= by) { ++lm; ly -= by; } else if (ly < 0) { --lm; ly += by; }
    };
    // cut cells: bit lx of row (ly, lm) = the cell with lower corner (lx, ly, lm)
    for (int t = threadIdx.x; t < nrows; t += blockDim.x) {
        int ly, lm;
        rowcoords(t, ly, lm);
        u64 cut = 0;
        if (ly + 1 < by && lm + 1 < bm) {
            u64 all = B[t] & B[t + 1] & B[t + by] & B[t + by + 1];
            u64 le = LE[t] | LE[t + 1] | LE[t + by] | LE[t + by + 1];
            u64 ge = GE[t] | GE[t + 1] | GE[t + by] | GE[t + by + 1];
            all &= all >> 1; le |= le >> 1; ge |= ge >> 1;
            cut = all & le & ge;
        }
        S0[t] = cut;
    }
    __syncthreads();
    // seeds: every corner of a cut cell
    for (int t = threadIdx.x; t < nrows; t += blockDim.x) {
        int ly, lm;
        rowcoords(t, ly, lm);
        u64 c = S0[t];
        if (ly > 0) c |= S0[t - 1];
        if (lm > 0) c |= S0[t - by];
        if (ly > 0 && lm > 0) c |= S0[t - by - 1];
        S1[t] = c | (c << 1);
    }
    __syncthreads();
    // nl von-Neumann dilations (bits shifted past the box only relay paths that exist in the full grid)
    u64 *cur = S1, *nxt = S0;
    for (int it = 0; it < nl; ++it) {
        for (int t = threadIdx.x; t < nrows; t += blockDim.x) {
            int ly, lm;
            rowcoords(t, ly, lm);
            const u64 c = cur[t];
            u64 r = c | (c << 1) | (c >> 1);
            if (ly > 0) r |= cur[t - 1];
            if (ly + 1 < by) r |= cur[t + 1];
            if (lm > 0) r |= cur[t - by];
            if (lm + 1 < bm) r |= cur[t + by];
            nxt[t] = r;
        }
        __syncthreads();
        u64* tmp = cur; cur = nxt; nxt = tmp;
    }
    int any = 0;
    if (x0 + a.tx <= a.n[0]) {
        // whole x-rows inside the grid: a thread expands 8 bits of a row word to 8 mask bytes and stores them at once
        typedef u64 w8 __attribute__((aligned(1)));
        const int ppr = a.tx / 8;                                  // 4 pieces per 32-node row
        for (int e = threadIdx.x; e < ppr * a.ty * a.tm; e += blockDim.x) {
            const int p = e % ppr, ry = (e / ppr) % a.ty, i = e / (ppr * a.ty);
            if (y0 + ry >= a.n[1] || m0 + i >= a.n[2]) continue;
            const u64 bits = (cur[(ry + ap) + by * (i + ap)] >> (8 * p + ap)) & 0xffull;
            u64 w = (bits * 0x0101010101010101ull) & 0x8040201008040201ull;      // byte k = bit k (as 1 << k)
            w = ((((w & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | w) >> 7) & 0x0101010101010101ull;
            *reinterpret_cast<w8*>(new_mask + (a.origin + x0 + 8 * p + (y0 + ry) * a.s1 + (m0 + i) * a.s2)) = w;
            any |= bits ? 1 : 0;
        }
    } else if (x < a.n[0] && y < a.n[1]) {
        for (int i = pg; i < a.tm && m0 + i < a.n[2]; i += npg) {
            const unsigned char on = (unsigned char)((cur[(ty_ + ap) + by * (i + ap)] >> (tx_ + ap)) & 1ull);
            new_mask[a.origin + x + y * a.s1 + (m0 + i) * a.s2] = on;
            any |= on;
        }
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) tiles[tile] = any ? 1 : 0;
}

// band_extrapolate_kernel for 32×8×mc tiles; the block's 256-thread groups share the planes, J per thread and round
template <int J>
__global__ void __launch_bounds__(256) band_search3_kernel(BandArgs a, const unsigned char* target, unsigned char* halo,
                                                           const unsigned char* src_mask, const signed char* ring, int nring,
                                                           int nring_lds, const void* src, void* dst, int* miss,
                                                           BandEntry* list, unsigned* list_count, unsigned list_cap) {
    const unsigned tile = LSM_TILE_ID(a);
    if (a.work && !a.work[tile]) return;
    const int x0 = (tile % a.nbx) * a.tx, y0 = ((tile / a.nbx) % a.nby) * a.ty, m0 = (tile / (a.nbx * a.nby)) * a.tm;
    const int tx_ = threadIdx.x & 31, ty_ = (threadIdx.x >> 5) & 7, pg = threadIdx.x >> 8, npg = blockDim.x >> 8;
    const int x = x0 + tx_, y = y0 + ty_;
    const bool inxy = x < a.n[0] && y < a.n[1];
    const long long qxy = a.origin + x + y * a.s1;
    if (target) {   // nothing to do in most tiles: skip them before staging anything
        int any = 0;
        if (inxy)
            for (int i = pg; i < a.tm && m0 + i < a.n[2]; i += npg) {
                const long long q = qxy + (m0 + i) * a.s2;
                any |= (target[q] && !src_mask[q]) ? 1 : 0;
            }
        if (!__syncthreads_or(any)) return;
    }
    const int bx = a.tx + 2 * RL, by = a.ty + 2 * RL, bm = a.tm + 2 * RL, nrows = by * bm;
    extern __shared__ u64 w3[];
    u64 *B = w3, *T = w3 + nrows;            // T: halo words of the tile rows (ty × tm)
    // xkey[pattern]: (o_x² << 9) | (o_x + RL) of the set bit nearest to the centre of a 7-bit line pattern, the lower x
    // winning ties; an empty pattern gets a key no real candidate can reach
    unsigned* xkey = reinterpret_cast<unsigned*>(T + a.ty * a.tm);
    for (int p = threadIdx.x; p < 128; p += blockDim.x) {
        unsigned k = 1u << 20;
        for (int d = RL; d >= 0; --d) {          // farthest first, so the nearest (and on ties the lower x) is written last
            if (p & (1 << (RL + d))) k = ((unsigned)(d * d) << 9) | (unsigned)(RL + d);
            if (p & (1 << (RL - d))) k = ((unsigned)(d * d) << 9) | (unsigned)(RL - d);
        }
        xkey[p] = k;
    }
    stage_mask_rows8(a, RL, bx, by, bm, x0, y0, m0, B, src_mask);
    __syncthreads();
    if (halo) {
        // what stencils centred on band nodes read: LSM_GHOST nodes along each axis and the 3^3 box
        for (int t = threadIdx.x; t < a.ty * a.tm; t += blockDim.x) {
            const int ly = t % a.ty + RL, lm = t / a.ty + RL, r = ly + by * lm;
            u64 c = B[r];
            u64 h = c | (c << 1) | (c >> 1) | (c << 2) | (c >> 2) | (c << 3) | (c >> 3);
            for (int k = 1; k <= LSM_GHOST; ++k) h |= B[r + k] | B[r - k] | B[r + k * by] | B[r - k * by];
            for (int dm = -1; dm <= 1; ++dm)
                for (int dy = -1; dy <= 1; ++dy) {
                    const u64 d = B[r + dy + dm * by];
                    h |= d | (d << 1) | (d >> 1);
                }
            T[t] = h;
        }
        __syncthreads();
    }
    // The (node, nearest band node) pairs of the tile are appended with ONE global atomic per block and
    // J·npg planes: lanes take tile-local slots through an LDS counter first.
    __shared__ unsigned s_cnt, s_base;
    const int lx = tx_ + RL, ly = ty_ + RL, lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < a.tm; i0 += J * npg) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        unsigned rec[J], slot[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int i = i0 + j * npg + pg, m = m0 + i, lm = i + RL, r = ly + by * lm;
            const long long q = qxy + m * a.s2;
            rec[j] = 0; slot[j] = 0;
            bool want = inxy && i < a.tm && m < a.n[2];
            if (want) {
                const bool inband = (B[r] >> lx) & 1ull;
                if (halo) {
                    if (inband) { halo[q] = 1; want = false; }
                    else if (!(halo[q] | (unsigned char)((T[ty_ + a.ty * i] >> lx) & 1ull))) want = false;
                    else halo[q] = 1;
                } else if (!target[q] || inband) {
                    want = false;
                }
            }
            if (want) {
                const int I[3] = {x, y, m};
                int P[3] = {0, 0, 0};
                // _nearest_band_node inside the (2·RL+1)^3 cube: the first ring hit is the set bit with the smallest
                // (|off|², o_z, o_y, o_x) — the ring is the column-major offset list stably sorted by |off|².
                // Per x-line: the nearest set bit to lx, the lower x winning ties.
                // (the per-line part of the key comes from a 128-entry table over the 7-bit pattern around lx)
                unsigned best = 0xffffffffu;
#pragma unroll
                for (int dm = -RL; dm <= RL; ++dm)
#pragma unroll
                    for (int dy = -RL; dy <= RL; ++dy) {
                        const unsigned pat = (unsigned)(B[r + dy + dm * by] >> (lx - RL)) & 0x7fu;   // bit j <-> o_x = j - RL
                        const unsigned key = xkey[pat] + (((unsigned)(dy * dy + dm * dm) << 9) | (unsigned)((dm + RL) * 49 + (dy + RL) * 7));
                        best = key < best ? key : best;
                    }
                bool found = false;
                if (best < (16u << 9)) {       // closer than any offset with a component beyond RL: final
                    const int code = best & 511;
                    P[0] = x + code % 7 - RL; P[1] = y + (code / 7) % 7 - RL; P[2] = m + code / 49 - RL;
                    found = true;
                } else {
                    found = ring_scan_global(a, src_mask, ring, nring_lds, nring, I, P);
                }
                finish_node(a, q, I, P, found, src_mask, src, dst, miss, nullptr, nullptr, 0);
                if (found) rec[j] = 0x8000u | (unsigned)(I[0] - P[0] + 8) | ((unsigned)(I[1] - P[1] + 8) << 4) | ((unsigned)(I[2] - P[2] + 8) << 8);
            }
            if (list) {
                const u64 bal = __ballot(rec[j] != 0);
                if (bal) {
                    const int leader = __ffsll((long long)bal) - 1;
                    unsigned base = 0;
                    if (lane == leader) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
                    slot[j] = __shfl(base, leader, 64) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                }
            }
        }
        if (list) {
            __syncthreads();
            if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(list_count, s_cnt) : 0u;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (!rec[j]) continue;
                const unsigned k = s_base + slot[j];
                if (k >= list_cap) continue;
                const int d0 = (int)(rec[j] & 15u) - 8, d1 = (int)((rec[j] >> 4) & 15u) - 8, d2 = (int)((rec[j] >> 8) & 15u) - 8;
                BandEntry e;
                e.q = qxy + (long long)(m0 + i0 + j * npg + pg) * a.s2;
                e.rel = -(int)(d0 + d1 * a.s1 + d2 * a.s2);
                e.d[0] = (signed char)d0; e.d[1] = (signed char)d1; e.d[2] = (signed char)d2; e.d[3] = 0;
                list[k] = e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bit-row path of update_band! (3-D, 32×8×tm tiles, nlayers <= 3, a band that moves by less than a tile per step — the
// conditions of the "listed" / "local" update of lsm_band_update).  The kernels above stage the byte mask and the values of a
// tile's whole box (tile + apron: 5 times the tile) — one 40-byte mask row per lane and one 160-byte value row per wave
// instruction, 17 k work tiles per step at 768³: they are bound by the latency of those rows.  Here every node is read ONCE:
//   band_bits_kernel       per ACTIVE tile: mask bytes and values of the tile's own nodes -> three bit words per x-line
//                          (in the band / value <= 0 / value >= 0), tile-major in a scratch array (64 words per tile and kind);
//   band_grow_bits_kernel  per work tile: the box's row words assembled from the words of the 27 neighbouring tiles (one round
//                          of independent 4-byte loads per thread; tiles without a band node are known from their flag and
//                          not read), cut cells -> seeds -> dilations as in band_grow3_kernel, then — nothing else reads the
//                          byte mask any more — the new band is written IN PLACE (only x-lines that changed), the newly
//                          active nodes get their extrapolated value from the old band's words in LDS (_extrapolate_to_ghost,
//                          src/meshfield.jl:481-511; no second kernel, no scratch mask, no copy pass), and the new band's words
//                          are left for
//   band_halo_bits_kernel  per work tile: halo mask and (node, nearest band node) list of the NEW band from its row words
//                          (band_search3_kernel's search), with the slope neighbours of the nearest node resolved here —
//                          band_apply_kernel then needs no mask reads.
// ------------------------------------------------------------------------------------------------
constexpr int BAP = 4;     // apron of the row-word boxes: nlayers + 1 <= 4 for the dilations, RL + 1 for the slope neighbours of a nearest node

// flags of the 27 tiles around `tile` (0 outside the tile grid) -> nbflag[(dx+1) + 3 (dy+1) + 9 (dm+1)]
__device__ __forceinline__ void load_nbflags(const BandArgs& a, unsigned tile, const unsigned char* flags, unsigned char* nbflag) {
    if (threadIdx.x < 27) {
        const int bx = tile % a.nbx, by = (tile / a.nbx) % a.nby, bm = tile / (a.nbx * a.nby);
        const int X = bx + (int)(threadIdx.x % 3) - 1, Y = by + (int)((threadIdx.x / 3) % 3) - 1, M = bm + (int)(threadIdx.x / 9) - 1;
        const bool ok = (unsigned)X < a.nbx && (unsigned)Y < a.nby && (unsigned)M < a.nbm;
        nbflag[threadIdx.x] = ok ? flags[X + a.nbx * (Y + a.nby * M)] : (unsigned char)0;
    }
}
// row words of the box (tile + apron BAP) of up to three bit arrays, assembled from the tile-major words of the neighbours
template <int NA>
__device__ __forceinline__ void stage_bit_rows(const BandArgs& a, unsigned tile, int by, int bm, const unsigned char* nbflag,
                                               const unsigned* const (&src)[NA], u64* const (&dst)[NA]) {
    const int bxT = tile % a.nbx, byT = (tile / a.nbx) % a.nby, bmT = tile / (a.nbx * a.nby);
    const int y0 = byT * a.ty, m0 = bmT * a.tm, nrows = by * bm, wpt = a.ty * a.tm;
    const int bxw = a.tx + 2 * BAP;
    const float rby = 1.0f / (float)by;
    for (int t = threadIdx.x; t < nrows; t += blockDim.x) {
        int lm = (int)((float)t * rby), ly = t - lm * by;
        if (ly >= by) { ++lm; ly -= by; } else if (ly < 0) { --lm; ly += by; }
        const int gy = y0 - BAP + ly, gm = m0 - BAP + lm;
        u64 w[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) w[k] = 0;
        if ((unsigned)gy < (unsigned)a.n[1] && (unsigned)gm < (unsigned)a.n[2]) {
            const int tyi = gy / a.ty, tmi = gm / a.tm;
            const unsigned slot = (unsigned)((gy - tyi * a.ty) + a.ty * (gm - tmi * a.tm));
            const int fb = 3 * (tyi - byT + 1) + 9 * (tmi - bmT + 1);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int X = bxT + dx;
                if ((unsigned)X >= a.nbx || !nbflag[fb + dx + 1]) continue;
                const size_t sidx = (size_t)((unsigned)X + a.nbx * ((unsigned)tyi + a.nby * (unsigned)tmi)) * (size_t)wpt + slot;
#pragma unroll
                for (int k = 0; k < NA; ++k) {
                    const u64 x = src[k][sidx];
                    w[k] |= dx < 0 ? x >> (32 - BAP) : (dx == 0 ? x << BAP : x << (32 + BAP));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) dst[k][t] = w[k] & ((1ull << bxw) - 1ull);
    }
}
// 8 bits -> 8 mask bytes (byte k = bit k)
__device__ __forceinline__ u64 bits8_to_bytes(u64 bits) {
    u64 w = (bits * 0x0101010101010101ull) & 0x8040201008040201ull;
    return ((((w & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | w) >> 7) & 0x0101010101010101ull;
}
// xkey table of band_search3_kernel (nearest set bit of a 7-bit line pattern, the lower x winning ties)
__device__ __forceinline__ void fill_xkey(unsigned* xkey) {
    for (int p = threadIdx.x; p < 128; p += blockDim.x) {
        unsigned k = 1u << 20;
        for (int d = RL; d >= 0; --d) {
            if (p & (1 << (RL + d))) k = ((unsigned)(d * d) << 9) | (unsigned)(RL + d);
            if (p & (1 << (RL - d))) k = ((unsigned)(d * d) << 9) | (unsigned)(RL - d);
        }
        xkey[p] = k;
    }
}
// _nearest_band_node inside the 7^3 cube around box position (lx, r = ly + by·lm): smallest (|off|², o_z, o_y, o_x) among the set bits.
// The 49 x-lines of the cube are visited in shells of growing dy² + dm²; once every searching lane of the wave holds a hit nearer than
// the next shell can be (its |off|² >= that shell's dy² + dm²; ties included: they are only possible inside visited shells) the wave
// stops — a halo node of the first layer after 5 lines, one of the third after 29 at most.
struct ShellTab {
    signed char dy[49], dm[49];
    int end[10], next_r2[10];
    constexpr ShellTab() : dy{}, dm{}, end{}, next_r2{} {
        const int r2s[10] = {0, 1, 2, 4, 5, 8, 9, 10, 13, 18};
        int n = 0;
        for (int g = 0; g < 10; ++g) {
            for (int m = -RL; m <= RL; ++m)
                for (int y = -RL; y <= RL; ++y)
                    if (y * y + m * m == r2s[g]) { dy[n] = (signed char)y; dm[n] = (signed char)m; ++n; }
            end[g] = n;
            next_r2[g] = g < 9 ? r2s[g + 1] : (1 << 20);
        }
    }
};
__device__ __forceinline__ unsigned nearest_key(const u64* B, const unsigned* xkey, int by, int r, int lx) {
    constexpr ShellTab S{};
    static_assert(S.end[9] == 49, "the shells cover the 7 x 7 lines of the cube");
    unsigned best = 0xffffffffu;
    bool done = false;
#pragma unroll
    for (int g = 0; g < 10; ++g) {
        if (!done) {
#pragma unroll
            for (int i = (g == 0 ? 0 : S.end[g - 1]); i < S.end[g]; ++i) {
                const int dy = S.dy[i], dm = S.dm[i];
                const unsigned pat = (unsigned)(B[r + dy + dm * by] >> (lx - RL)) & 0x7fu;   // bit j <-> o_x = j - RL
                const unsigned key = xkey[pat] + (((unsigned)(dy * dy + dm * dm) << 9) | (unsigned)((dm + RL) * 49 + (dy + RL) * 7));
                best = key < best ? key : best;
            }
            // wave-uniform: nobody can still find a nearer (or an equally near, earlier) node in the shells to come
            done = __ballot((best >> 9) >= (unsigned)S.next_r2[g]) == 0ull;
        }
    }
    return best;
}

__global__ void __launch_bounds__(256) band_bits_kernel(BandArgs a, const void* v, const unsigned char* mask, unsigned* OB, unsigned* LE,
                                                        unsigned* GE, unsigned ntile_blocks, const unsigned char* flags_src,
                                                        unsigned char* flags_dst, unsigned nflags, unsigned* zero0) {
    if (blockIdx.x >= ntile_blocks) {
        // the workgroups behind the tiles' copy the old band's tile flags aside (the update writes the new ones in place) and clear the
        // halo list's counter: two small launches less per step
        const unsigned c = (blockIdx.x - ntile_blocks) * 4096u + threadIdx.x * 16u;
        for (unsigned k = c; k < c + 16u && k < nflags; ++k) flags_dst[k] = flags_src[k];
        if (blockIdx.x == ntile_blocks && threadIdx.x == 0 && zero0) *zero0 = 0u;
        return;
    }
    const unsigned tile = LSM_TILE_ID(a);
    if (a.work && !a.work[tile]) return;
    const int x0 = (tile % a.nbx) * a.tx, y0 = ((tile / a.nbx) % a.nby) * a.ty, m0 = (tile / (a.nbx * a.nby)) * a.tm;
    const int p = threadIdx.x & 3, ry = (threadIdx.x >> 2) & 7, wpt = a.ty * a.tm;
    const int gx0 = x0 + 8 * p, gy = y0 + ry;
    int hi = a.n[0] - gx0;
    hi = hi > 8 ? 8 : (hi < 0 ? 0 : hi);
    const u64 keep = hi > 0 ? (~0ull >> (64 - 8 * hi)) : 0ull;
    for (int i = threadIdx.x >> 5; i < a.tm; i += 8) {
        const int gm = m0 + i;
        unsigned b8 = 0, le8 = 0, ge8 = 0;
        if (gy < a.n[1] && gm < a.n[2] && keep) {
            const long long q0 = a.origin + gx0 + (long long)gy * a.s1 + (long long)gm * a.s2;
            u64 w = *reinterpret_cast<const u64_unaligned*>(mask + q0) & keep;
            u64 t = ((w & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | w;
            t = (t >> 7) & 0x0101010101010101ull;
            b8 = (unsigned)((t * 0x0102040810204080ull) >> 56);
            if (b8) {
                // the piece's eight values in two (float) or four (double) 16-byte loads, band node or not — 32 resp. 64 contiguous
                // bytes per lane, four lanes to a row — instead of eight predicated element loads: the kernel is bound by the
                // cache lines its load instructions touch, not by their bytes.  (hi < 8: the piece hangs over the row's end — element loads.)
                double xv[8];
                if (hi == 8 && a.f32) {
                    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
                    const f4* pf = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(v) + q0);
                    const f4 u0 = pf[0], u1 = pf[1];
                    xv[0] = u0.x; xv[1] = u0.y; xv[2] = u0.z; xv[3] = u0.w; xv[4] = u1.x; xv[5] = u1.y; xv[6] = u1.z; xv[7] = u1.w;
                } else if (hi == 8) {
                    typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
                    const d2* pd = reinterpret_cast<const d2*>(reinterpret_cast<const double*>(v) + q0);
                    const d2 u0 = pd[0], u1 = pd[1], u2 = pd[2], u3 = pd[3];
                    xv[0] = u0.x; xv[1] = u0.y; xv[2] = u1.x; xv[3] = u1.y; xv[4] = u2.x; xv[5] = u2.y; xv[6] = u3.x; xv[7] = u3.y;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) xv[k] = ld_val(v, q0 + (((b8 >> k) & 1u) ? k : 0), a.f32);   // always a valid address
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned on = (b8 >> k) & 1u;
                    le8 |= (on && xv[k] <= 0.0) ? (1u << k) : 0u;
                    ge8 |= (on && xv[k] >= 0.0) ? (1u << k) : 0u;
                }
            }
        }
        unsigned wb = b8 << (8 * p), wl = le8 << (8 * p), wg = ge8 << (8 * p);
        wb |= __shfl_xor(wb, 1, 64); wl |= __shfl_xor(wl, 1, 64); wg |= __shfl_xor(wg, 1, 64);
        wb |= __shfl_xor(wb, 2, 64); wl |= __shfl_xor(wl, 2, 64); wg |= __shfl_xor(wg, 2, 64);
        if (p == 0) {
            const size_t sidx = (size_t)tile * wpt + ry + a.ty * i;
            OB[sidx] = wb; LE[sidx] = wl; GE[sidx] = wg;
        }
    }
}

__global__ void __launch_bounds__(256) band_grow_bits_kernel(BandArgs a, void* v, unsigned char* mask, int nl, const unsigned char* old_tiles,
                                                             unsigned char* tiles, const unsigned* OB, const unsigned* LEw, const unsigned* GEw,
                                                             unsigned* NB, int* miss) {
    const unsigned tile = LSM_TILE_ID(a);
    if (a.work && !a.work[tile]) return;
    const int x0 = (tile % a.nbx) * a.tx, y0 = ((tile / a.nbx) % a.nby) * a.ty, m0 = (tile / (a.nbx * a.nby)) * a.tm;
    const int ap = BAP, by = a.ty + 2 * ap, bm = a.tm + 2 * ap, nrows = by * bm, wpt = a.ty * a.tm;
    extern __shared__ u64 w3[];
    u64 *B = w3, *LE = w3 + nrows, *GE = w3 + 2 * nrows, *S0 = w3 + 3 * nrows, *S1 = w3 + 4 * nrows;
    __shared__ unsigned char nbflag[32];
    load_nbflags(a, tile, old_tiles, nbflag);
    __syncthreads();
    {
        const unsigned* const src[3] = {OB, LEw, GEw};
        u64* const dst[3] = {B, LE, GE};
        stage_bit_rows<3>(a, tile, by, bm, nbflag, src, dst);
    }
    __syncthreads();
    {
        int anyb = 0;
        for (int t = threadIdx.x; t < nrows; t += blockDim.x) anyb |= B[t] ? 1 : 0;
        if (!__syncthreads_or(anyb)) return;      // no band node in the box: the tile was and stays empty (flag, mask and words untouched = 0)
    }
    const float rby = 1.0f / (float)by;
    auto rowcoords_slow = [&](int t, int& ly, int& lm) {
        lm = (int)((float)t * rby); ly = t - lm * by;
        if (ly >= by) { ++lm; ly -= by; } else if (ly < 0) { --lm; ly += by; }
    };
    int ly_own, lm_own, ly_own2, lm_own2;       // a thread's (at most two, tm <= 24) box rows: their coordinates once for all passes
    rowcoords_slow((int)threadIdx.x, ly_own, lm_own);
    rowcoords_slow((int)threadIdx.x + 256, ly_own2, lm_own2);
    auto rowcoords = [&](int t, int& ly, int& lm) {
        if (t == (int)threadIdx.x) { ly = ly_own; lm = lm_own; }
        else if (t == (int)threadIdx.x + 256) { ly = ly_own2; lm = lm_own2; }
        else rowcoords_slow(t, ly, lm);
    };
    for (int t = threadIdx.x; t < nrows; t += blockDim.x) {      // cut cells (see band_grow3_kernel)
        int ly, lm;
        rowcoords(t, ly, lm);
        u64 cut = 0;
        if (ly + 1 < by && lm + 1 < bm) {
            u64 all = B[t] & B[t + 1] & B[t + by] & B[t + by + 1];
            u64 le = LE[t] | LE[t + 1] | LE[t + by] | LE[t + by + 1];
            u64 ge = GE[t] | GE[t + 1] | GE[t + by] | GE[t + by + 1];
            all &= all >> 1; le |= le >> 1; ge |= ge >> 1;
            cut = all & le & ge;
        }
        S0[t] = cut;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nrows; t += blockDim.x) {      // seeds
        int ly, lm;
        rowcoords(t, ly, lm);
        u64 c = S0[t];
        if (ly > 0) c |= S0[t - 1];
        if (lm > 0) c |= S0[t - by];
        if (ly > 0 && lm > 0) c |= S0[t - by - 1];
        S1[t] = c | (c << 1);
    }
    __syncthreads();
    u64 *cur = S1, *nxt = S0;
    for (int it = 0; it < nl; ++it) {                             // nl von-Neumann dilations
        for (int t = threadIdx.x; t < nrows; t += blockDim.x) {
            int ly, lm;
            rowcoords(t, ly, lm);
            const u64 c = cur[t];
            u64 r = c | (c << 1) | (c >> 1);
            if (ly > 0) r |= cur[t - 1];
            if (ly + 1 < by) r |= cur[t + 1];
            if (lm > 0) r |= cur[t - by];
            if (lm + 1 < bm) r |= cur[t + by];
            nxt[t] = r;
        }
        __syncthreads();
        u64* tmp = cur; cur = nxt; nxt = tmp;
    }
    // outputs: the new band's words, the byte mask where an x-line changed, the tile flag; NEWB = newly active nodes of the tile rows
    unsigned* NEWB = reinterpret_cast<unsigned*>(nxt);
    unsigned* xkey = NEWB + wpt;
    const int nxv = a.n[0] - x0;
    const unsigned xvalid = nxv >= 32 ? 0xffffffffu : ((1u << nxv) - 1u);
    int any = 0, anynew = 0;
    for (int e = threadIdx.x; e < 4 * wpt; e += blockDim.x) {
        const int p = e & 3, row = e >> 2, ry = row % a.ty, i = row / a.ty;
        const bool ing = y0 + ry < a.n[1] && m0 + i < a.n[2];
        const int r = (ry + ap) + by * (i + ap);
        const unsigned nw = ing ? ((unsigned)(cur[r] >> ap) & xvalid) : 0u, ow = (unsigned)(B[r] >> ap);
        if (p == 0) { NB[(size_t)tile * wpt + row] = nw; NEWB[row] = nw & ~ow; }
        any |= nw ? 1 : 0;
        anynew |= (nw & ~ow) ? 1 : 0;
        const unsigned n8 = (nw >> (8 * p)) & 0xffu, o8 = (ow >> (8 * p)) & 0xffu;
        if (ing && n8 != o8) {
            const long long q0 = a.origin + x0 + 8 * p + (long long)(y0 + ry) * a.s1 + (long long)(m0 + i) * a.s2;
            if (x0 + 8 * p + 8 <= a.n[0]) *reinterpret_cast<u64_unaligned*>(mask + q0) = bits8_to_bytes(n8);
            else for (int k = 0; k < 8 && x0 + 8 * p + k < a.n[0]; ++k) mask[q0 + k] = (unsigned char)((n8 >> k) & 1u);
        }
    }
    any = __syncthreads_or(any);
    anynew = __syncthreads_or(anynew);
    if (threadIdx.x == 0) tiles[tile] = any ? 1 : 0;
    if (!anynew) return;
    // newly active nodes: affine extrapolant from the nearest node of the OLD band (B), in place — reads old band nodes only, writes new ones only
    fill_xkey(xkey);
    __syncthreads();
    const int tx_ = threadIdx.x & 31, ty_ = (threadIdx.x >> 5) & 7, pg = threadIdx.x >> 8, npg = blockDim.x >> 8;
    const int x = x0 + tx_, y = y0 + ty_, lx = tx_ + ap;
    for (int i = pg; i < a.tm; i += npg) {
        if (!((NEWB[ty_ + a.ty * i] >> tx_) & 1u)) continue;
        const int m = m0 + i, r = (ty_ + ap) + by * (i + ap);
        const unsigned best = nearest_key(B, xkey, by, r, lx);
        if (best >= (16u << 9)) { atomicOr(miss, 1); continue; }      // cannot happen for nl <= 3: a new node lies within nl L1-steps of the old band
        const int code = best & 511, ox = code % 7 - RL, oy = (code / 7) % 7 - RL, oz = code / 49 - RL;
        const int rP = r + oy + oz * by, lxP = lx + ox;
        const long long q = a.origin + x + (long long)y * a.s1 + (long long)m * a.s2;
        const long long qp = q + ox + (long long)oy * a.s1 + (long long)oz * a.s2;
        const double phiP = ld_val(v, qp, a.f32);
        double val = phiP;
        const int delta[3] = {-ox, -oy, -oz};
        const int rstep[3] = {0, 1, by};
        const long long sd[3] = {1, a.s1, a.s2};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (delta[d] == 0) continue;
            const bool plus = d == 0 ? ((B[rP] >> (lxP + 1)) & 1ull) : ((B[rP + rstep[d]] >> lxP) & 1ull);
            const bool minus = d == 0 ? ((B[rP] >> (lxP - 1)) & 1ull) : ((B[rP - rstep[d]] >> lxP) & 1ull);
            double slope = 0.0;                                   // _axis_slope: + neighbour first, then -
            if (plus) slope = ld_val(v, qp + sd[d], a.f32) - phiP;
            else if (minus) slope = phiP - ld_val(v, qp - sd[d], a.f32);
            val += (double)delta[d] * slope;
        }
        const double sv = val > 0 ? 1.0 : (val < 0 ? -1.0 : val), sp = phiP > 0 ? 1.0 : (phiP < 0 ? -1.0 : phiP);
        st_val(v, q, a.f32, (phiP == 0.0 || sv == sp) ? val : phiP);
    }
}

// halo mask and (node, nearest band node) list of the new band from its row words; interior bands only (no boundary-condition
// sources pre-marked in halo[]): the halo bytes of every visited tile are written whole, no clearing pass
__global__ void __launch_bounds__(256) band_halo_bits_kernel(BandArgs a, const unsigned char* tiles, const unsigned* NB, unsigned char* halo,
                                                             int* miss, BandEntry* list, unsigned* list_count, unsigned list_cap) {
    const unsigned tile = LSM_TILE_ID(a);
    if (a.work && !a.work[tile]) return;
    const int x0 = (tile % a.nbx) * a.tx, y0 = ((tile / a.nbx) % a.nby) * a.ty, m0 = (tile / (a.nbx * a.nby)) * a.tm;
    const int ap = BAP, by = a.ty + 2 * ap, bm = a.tm + 2 * ap, nrows = by * bm, wpt = a.ty * a.tm;
    extern __shared__ u64 w3[];
    u64 *B = w3, *T = w3 + nrows;
    unsigned* xkey = reinterpret_cast<unsigned*>(T + wpt);
    __shared__ unsigned char nbflag[32];
    __shared__ unsigned s_cnt, s_base;
    load_nbflags(a, tile, tiles, nbflag);
    fill_xkey(xkey);
    __syncthreads();
    {
        const unsigned* const src[1] = {NB};
        u64* const dst[1] = {B};
        stage_bit_rows<1>(a, tile, by, bm, nbflag, src, dst);
    }
    __syncthreads();
    // what stencils centred on band nodes read: LSM_GHOST nodes along each axis and the 3^3 box
    int anyh = 0;
    for (int t = threadIdx.x; t < wpt; t += blockDim.x) {
        const int ly = t % a.ty + ap, lm = t / a.ty + ap, r = ly + by * lm;
        const u64 c = B[r];
        u64 h = c | (c << 1) | (c >> 1) | (c << 2) | (c >> 2) | (c << 3) | (c >> 3);
        for (int k = 1; k <= LSM_GHOST; ++k) h |= B[r + k] | B[r - k] | B[r + k * by] | B[r - k * by];
        for (int dm = -1; dm <= 1; ++dm)
            for (int dy = -1; dy <= 1; ++dy) {
                const u64 d = B[r + dy + dm * by];
                h |= d | (d << 1) | (d >> 1);
            }
        T[t] = h;
        anyh |= ((unsigned)(h >> ap)) ? 1 : 0;
    }
    anyh = __syncthreads_or(anyh);
    // halo bytes of the tile (band nodes included), whole x-lines
    const int nxv = a.n[0] - x0;
    const unsigned xvalid = nxv >= 32 ? 0xffffffffu : ((1u << nxv) - 1u);
    for (int e = threadIdx.x; e < 4 * wpt; e += blockDim.x) {
        const int p = e & 3, row = e >> 2, ry = row % a.ty, i = row / a.ty;
        if (y0 + ry >= a.n[1] || m0 + i >= a.n[2]) continue;
        const unsigned hw = anyh ? ((unsigned)(T[row] >> ap) & xvalid) : 0u, h8 = (hw >> (8 * p)) & 0xffu;
        const long long q0 = a.origin + x0 + 8 * p + (long long)(y0 + ry) * a.s1 + (long long)(m0 + i) * a.s2;
        if (x0 + 8 * p + 8 <= a.n[0]) *reinterpret_cast<u64_unaligned*>(halo + q0) = bits8_to_bytes(h8);
        else for (int k = 0; k < 8 && x0 + 8 * p + k < a.n[0]; ++k) halo[q0 + k] = (unsigned char)((h8 >> k) & 1u);
    }
    if (!anyh) return;
    // The wanted nodes (in the halo, not in the band) are few — 130 of a tile's 2048 on average at 768³ — and scattered: one
    // thread per NODE with lanes waiting for the few that search ran the 49-row scan for almost every wave.  They are listed
    // first (row-major, x ascending: the order of the tile's entries is deterministic) and then searched by full waves.
    unsigned* cnt = reinterpret_cast<unsigned*>(xkey + 128);                      // wpt + 1 words
    unsigned short* wl = reinterpret_cast<unsigned short*>(cnt + 2 * wpt + 2);   // up to 32·wpt entries: (row << 5) | x
    for (int t = threadIdx.x; t < wpt; t += blockDim.x) {
        const int ry = t % a.ty, i = t / a.ty;
        const bool ing = y0 + ry < a.n[1] && m0 + i < a.n[2];
        const int r = (ry + ap) + by * (i + ap);
        const unsigned w = ing ? ((unsigned)(T[t] >> ap) & ~(unsigned)(B[r] >> ap) & xvalid) : 0u;
        cnt[t] = w;
    }
    __syncthreads();
    // offsets of the rows' entries: one wave scans the counts, two rows per lane (wpt <= 128)
    unsigned* offs = cnt + wpt + 1;
    if (threadIdx.x < 64) {
        const int l = (int)threadIdx.x;
        const unsigned c0 = 2 * l < wpt ? (unsigned)__builtin_popcount(cnt[2 * l]) : 0u, c1 = 2 * l + 1 < wpt ? (unsigned)__builtin_popcount(cnt[2 * l + 1]) : 0u;
        unsigned inc = c0 + c1;
        for (int o = 1; o < 64; o <<= 1) { const unsigned vv = __shfl_up(inc, o, 64); if (l >= o) inc += vv; }
        const unsigned ex = inc - (c0 + c1);
        if (2 * l < wpt) offs[2 * l] = ex;
        if (2 * l + 1 < wpt) offs[2 * l + 1] = ex + c0;
        if (l == 63) s_cnt = inc;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < wpt; t += blockDim.x) {
        unsigned w = cnt[t], off = offs[t];
        while (w) {
            const int bpos = __builtin_ctz(w);
            wl[off++] = (unsigned short)((t << 5) | bpos);
            w &= w - 1;
        }
    }
    __syncthreads();
    const unsigned tot = s_cnt;
    if (tot == 0) return;
    if (threadIdx.x == 0) s_base = list ? atomicAdd(list_count, tot) : 0u;
    __syncthreads();
    const unsigned base = s_base;
    for (unsigned k = threadIdx.x; k < tot; k += blockDim.x) {
        const unsigned code16 = wl[k];
        const int row = (int)(code16 >> 5), tx_ = (int)(code16 & 31u), ry = row % a.ty, i = row / a.ty;
        const int lx = tx_ + ap, r = (ry + ap) + by * (i + ap);
        const unsigned best = nearest_key(B, xkey, by, r, lx);
        if (best >= (16u << 9)) { atomicOr(miss, 1); continue; }      // a halo node lies within Chebyshev distance 3 of the band: cannot happen
        const int code = best & 511, ox = code % 7 - RL, oy = (code / 7) % 7 - RL, oz = code / 49 - RL;
        const int rP = r + oy + oz * by, lxP = lx + ox;
        // _axis_slope's choice per axis with a non-zero offset (+ neighbour first, then -): 1 = plus, 2 = minus
        unsigned sc = 0;
        if (ox) sc |= ((B[rP] >> (lxP + 1)) & 1ull) ? 1u : (((B[rP] >> (lxP - 1)) & 1ull) ? 2u : 0u);
        if (oy) sc |= (((B[rP + 1] >> lxP) & 1ull) ? 1u : (((B[rP - 1] >> lxP) & 1ull) ? 2u : 0u)) << 2;
        if (oz) sc |= (((B[rP + by] >> lxP) & 1ull) ? 1u : (((B[rP - by] >> lxP) & 1ull) ? 2u : 0u)) << 4;
        if (!list || base + k >= list_cap) continue;
        BandEntry e;
        e.q = a.origin + (x0 + tx_) + (long long)(y0 + ry) * a.s1 + (long long)(m0 + i) * a.s2;
        e.rel = (int)(ox + oy * a.s1 + oz * a.s2);
        e.d[0] = (signed char)(-ox); e.d[1] = (signed char)(-oy); e.d[2] = (signed char)(-oz);      // I - P
        e.d[3] = (signed char)(0x40u | sc);                                // 0x40: the slope neighbours are resolved (2 bits per axis)
        list[base + k] = e;
    }
}

// _extrapolate_to_ghost from a precomputed (node, nearest band node) list: one thread per entry.  n_host >= 0: the list's
// length as the host knows it (lsm_band_status) — every wave reading the device counter is 16 k requests for ONE cache line.
template <int NDIM, int U>
__global__ void __launch_bounds__(256) band_apply_kernel(BandArgs a, const BandEntry* list, const unsigned* __restrict__ list_count, long long n_host,
                                                         unsigned list_cap, const unsigned char* src_mask, const void* src, void* dst) {
    // the host's length sizes the launch (lsm_band_status read it anyway); the device counter still bounds the entries (one
    // scalar load per wave): a host that rewrote the list in place and kept a stale length cannot make the kernel gather
    // through entries that do not exist
    const unsigned n_dev = *list_count;
    unsigned n = n_host >= 0 && (unsigned long long)n_host < n_dev ? (unsigned)n_host : n_dev;
    n = n < list_cap ? n : list_cap;
    // U entries per thread and round, a block apart (coalesced): the entries, then all their value loads, are in flight together —
    // the kernel is a chain of two memory round trips per entry and nothing else
    for (unsigned i0 = blockIdx.x * (blockDim.x * U) + threadIdx.x; i0 < n; i0 += gridDim.x * blockDim.x * U) {
        BandEntry e[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * blockDim.x;
            ok[u] = i < n;
            e[u] = list[ok[u] ? i : i0];
        }
        double phiP[U], nbv[U][NDIM];
        const long long sdv[3] = {1, a.s1, a.s2};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long qp = e[u].q + e[u].rel;
            const unsigned sc = (unsigned)(unsigned char)e[u].d[3];
            phiP[u] = ld_val(src, qp, a.f32);
#pragma unroll
            for (int d = 0; d < NDIM; ++d) {
                unsigned c = (sc >> (2 * d)) & 3u;
                if (!(sc & 0x40u)) {
                    // entries of the byte-mask search: _axis_slope's choice is made here (mask ghosts are 0: no bounds test needed)
                    c = e[u].d[d] == 0 ? 0u : (src_mask[qp + sdv[d]] ? 1u : (src_mask[qp - sdv[d]] ? 2u : 0u));
                    e[u].d[3] = (signed char)((unsigned char)e[u].d[3] | (c << (2 * d)));
                }
                // only the slopes the entry uses are fetched (an axis-aligned halo node needs one): the gathers are bound by the L1's look-ups
                nbv[u][d] = phiP[u];
                if (c != 0u && e[u].d[d] != 0) nbv[u][d] = ld_val(src, c == 1u ? qp + sdv[d] : qp - sdv[d], a.f32);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const unsigned sc = (unsigned)(unsigned char)e[u].d[3];
            double val = phiP[u];
#pragma unroll
            for (int d = 0; d < NDIM; ++d) {
                const unsigned c = (sc >> (2 * d)) & 3u;
                const int delta = e[u].d[d];
                if (delta == 0) continue;
                const double slope = c == 1u ? nbv[u][d] - phiP[u] : (c == 2u ? phiP[u] - nbv[u][d] : 0.0);
                val += (double)delta * slope;
            }
            const double sv = val > 0 ? 1.0 : (val < 0 ? -1.0 : val), sp = phiP[u] > 0 ? 1.0 : (phiP[u] < 0 ? -1.0 : phiP[u]);
            st_val(dst, e[u].q, a.f32, (phiP[u] == 0.0 || sv == sp) ? val : phiP[u]);
        }
    }
}

// is the (possibly out-of-grid) position p read by a stencil centred on a band node?
__device__ __forceinline__ bool band_reads(const BandArgs& a, int r, const unsigned char* band, const int p[3]) {
    auto probe = [&](int x, int y, int z) -> bool {
        if (x < 0 || x >= a.n[0] || y < 0 || y >= a.n[1] || z < 0 || z >= a.n[2]) return false;
        return band[a.origin + x + y * a.s1 + z * a.s2] != 0;
    };
    for (int k = -r; k <= r; ++k) {
        if (probe(p[0] + k, p[1], p[2])) return true;
        if (a.ndim > 1 && probe(p[0], p[1] + k, p[2])) return true;
        if (a.ndim > 2 && probe(p[0], p[1], p[2] + k)) return true;
    }
    for (int k2 = (a.ndim > 2 ? -1 : 0); k2 <= (a.ndim > 2 ? 1 : 0); ++k2)
        for (int k1 = (a.ndim > 1 ? -1 : 0); k1 <= (a.ndim > 1 ? 1 : 0); ++k1)
            for (int k0 = -1; k0 <= 1; ++k0)
                if (probe(p[0] + k0, p[1] + k1, p[2] + k2)) return true;
    return false;
}

// Out-of-grid stencil positions resolve through _getindexbc (src/meshfield.jl:248-260), highest
// dimension first, to in-grid nodes: the boundary node's line for ExtrapolationBC{P} (nodes 0..P), the
// mirror node for SymmetryBC.  Those nodes must hold valid values too.  One launch per dimension d, from
// the last to the first: every needed position that is out of range in d (any position in the lower
// dimensions, in range in the higher ones) marks its sources, which the launches of the lower
// dimensions resolve further.  halo[] is indexed over the padded space; only in-grid marks are consumed.
__global__ void __launch_bounds__(256) band_halo_bc_kernel(BandArgs a, BandBcArgs bc, int d, int r, const unsigned char* band,
                                                           unsigned char* halo) {
    const int G = LSM_GHOST;
    long long ext[3];
    for (int k = 0; k < 3; ++k) ext[k] = k >= a.ndim ? 1 : (k < d ? a.n[k] + 2 * G : (k == d ? 2 * G : a.n[k]));
    const long long total = ext[0] * ext[1] * ext[2];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c[3] = {(int)(e % ext[0]), (int)((e / ext[0]) % ext[1]), (int)(e / (ext[0] * ext[1]))};
        int p[3];
        for (int k = 0; k < 3; ++k) p[k] = k >= a.ndim ? 0 : (k < d ? c[k] - G : (k == d ? (c[k] < G ? c[k] - G : a.n[d] + c[k] - G) : c[k]));
        const long long q = a.origin + p[0] + p[1] * a.s1 + p[2] * a.s2;
        if (!(halo[q] || band_reads(a, r, band, p))) continue;
        const int side = p[d] < 0 ? 0 : 1;
        const int k = side == 0 ? -p[d] : p[d] - (a.n[d] - 1);
        const long long sd = d == 0 ? 1 : (d == 1 ? a.s1 : a.s2);
        const long long q0 = q - p[d] * sd;                      // same line, node 0 of dimension d
        const int kind = bc.kind[d][side];
        if (kind == LSM_BC_EXTRAPOLATION) {
            for (int j = 0; j <= bc.degree[d][side]; ++j) halo[q0 + (side == 0 ? j : a.n[d] - 1 - j) * sd] = 1;
        } else if (kind == LSM_BC_SYMMETRY) {
            halo[q0 + (side == 0 ? k : a.n[d] - 1 - k) * sd] = 1;
        }
    }
}

// tile activity: tile is active iff it contains a band node (skipped tiles are written 0)
__global__ void __launch_bounds__(256) band_tiles_kernel(BandArgs a, const unsigned char* mask, unsigned char* tiles) {
    if (a.work && !a.work[LSM_TILE_ID(a)]) {
        if (threadIdx.x == 0) tiles[LSM_TILE_ID(a)] = 0;
        return;
    }
    LSM_TILE_PROLOGUE(a)
    int any = 0;
    LSM_TILE_FOR(a, x, y, m, q) any |= mask[q] ? 1 : 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) tiles[tile] = any ? 1 : 0;
}

// work[t] = OR of active over the 3^N tile neighbourhood of t.  (Stage pieces of two bricks marched by one workgroup were tried in
// round 3 — the launch 6 % slower, the step 1 % faster at best — and are gone: a stage launch takes the active-tile list.)
__global__ void __launch_bounds__(256) band_work_kernel(BandArgs a, const unsigned char* active, unsigned char* work, unsigned* zero_face,
                                                        int* zero_flags) {
    const unsigned nt = a.nbx * a.nby * a.nbm;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {                       // counters the list / Δt kernels behind this one accumulate into
        if (zero_face) *zero_face = 0u;
        if (zero_flags) { zero_flags[0] = 0; zero_flags[1] = 0; zero_flags[2] = 0; zero_flags[3] = 0; }
    }
    if (t >= nt) return;
    const int bx = t % a.nbx, by = (t / a.nbx) % a.nby, bm = t / (a.nbx * a.nby);
    unsigned char r = 0;
    for (int dm = -1; dm <= 1; ++dm)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int X = bx + dx, Y = by + dy, M = bm + dm;
                if (X < 0 || X >= (int)a.nbx || Y < 0 || Y >= (int)a.nby || M < 0 || M >= (int)a.nbm) continue;
                r |= active[X + a.nbx * (Y + a.nby * M)];
            }
    work[t] = r ? 1 : 0;
}

// compact, ordered lists of the active tiles and of the work tiles (active or next to one), so that the
// next launches cover those tiles only; counts[0..1] = list lengths, counts[2] = number of work tiles on a
// face of the grid (0: no band node can sit within a tile of the boundary before the next update, so no
// stencil reaches outside the grid; zeroed by the launcher, accumulated here).
// One workgroup per chunk of 8192 tiles (8 flags = one 64-bit word per thread, coalesced).  A chunk's offset into the
// lists is the number of set flags before it, which the workgroup counts itself (one word per thread and preceding
// chunk: 3 MB of reads in all at 768³) — no pass between workgroups, one launch.  (The first version was ONE workgroup
// whose threads each walked a contiguous run of flags: 0.10 ms at 768³, every load its own cache line.)
constexpr unsigned LISTS_CHUNK = 8192;
__device__ __forceinline__ unsigned nz_bytes(unsigned long long w) {       // number of non-zero bytes of a word
    w |= w >> 4; w |= w >> 2; w |= w >> 1;
    return (unsigned)__builtin_popcountll(w & 0x0101010101010101ull);
}
__global__ void __launch_bounds__(1024) band_lists_kernel(BandArgs a, const unsigned char* active, const unsigned char* work, int* act_list,
                                                          int* work_list, unsigned* counts) {
    __shared__ unsigned red[2][16], wsum[2][16];
    const unsigned ntiles = a.nbx * a.nby * a.nbm;
    const bool words = (((unsigned long long)active | (unsigned long long)work) & 7ull) == 0;
    auto flags8 = [&](const unsigned char* p, unsigned t) -> unsigned long long {   // flags of tiles t .. t+7 (t % 8 == 0)
        if (t >= ntiles) return 0ull;
        if (words && t + 8 <= ntiles) return *(const unsigned long long*)(p + t);
        unsigned long long r = 0;
        for (unsigned k = 0; k < 8 && t + k < ntiles; ++k) r |= (unsigned long long)p[t + k] << (8 * k);
        return r;
    };
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto wave_sum = [&](unsigned v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    // offset of this chunk = set flags in the chunks before it
    unsigned pa = 0, pw = 0;
    for (unsigned cb = 0; cb < blockIdx.x; ++cb) {
        const unsigned t = cb * LISTS_CHUNK + threadIdx.x * 8;
        pa += nz_bytes(flags8(active, t));
        pw += nz_bytes(flags8(work, t));
    }
    pa = wave_sum(pa); pw = wave_sum(pw);
    if (lane == 0) { red[0][wave] = pa; red[1][wave] = pw; }
    // own chunk: one word per thread, exclusive scan over the workgroup (wave scan + 16 wave totals)
    const unsigned t0 = blockIdx.x * LISTS_CHUNK + threadIdx.x * 8;
    const unsigned long long fa = flags8(active, t0), fw = flags8(work, t0);
    const unsigned na = nz_bytes(fa), nw = nz_bytes(fw);
    unsigned ia = na, iw = nw;                                    // inclusive wave scans
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned va = __shfl_up(ia, off, 64), vw = __shfl_up(iw, off, 64);
        if ((int)lane >= off) { ia += va; iw += vw; }
    }
    if (lane == 63) { wsum[0][wave] = ia; wsum[1][wave] = iw; }
    __syncthreads();
    unsigned offa = 0, offw = 0, tota = 0, totw = 0;
    for (unsigned w = 0; w < 16; ++w) {
        offa += red[0][w]; offw += red[1][w];
        if (w < wave) { offa += wsum[0][w]; offw += wsum[1][w]; }
        tota += wsum[0][w]; totw += wsum[1][w];
    }
    ia = offa + ia - na; iw = offw + iw - nw;
    unsigned nbd = 0;
    for (unsigned k = 0; k < 8; ++k) {
        if ((fa >> (8 * k)) & 0xff) act_list[ia++] = (int)(t0 + k);
        if ((fw >> (8 * k)) & 0xff) {
            work_list[iw++] = (int)(t0 + k);
            const unsigned tt = t0 + k, bx = tt % a.nbx, by = (tt / a.nbx) % a.nby, bm = tt / (a.nbx * a.nby);
            const bool face = bx == 0 || bx == a.nbx - 1 || (a.ndim == 3 && (by == 0 || by == a.nby - 1)) ||
                              (a.ndim >= 2 && (bm == 0 || bm == a.nbm - 1));
            nbd += face ? 1 : 0;
        }
    }
    nbd = wave_sum(nbd);
    if (lane == 0 && nbd) atomicAdd(&counts[2], nbd);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        unsigned ba = 0, bw = 0;
        for (unsigned w = 0; w < 16; ++w) { ba += red[0][w]; bw += red[1][w]; }
        counts[0] = ba + tota;
        counts[1] = bw + totw;
    }
}

// (One launch for work flags and lists — every chunk publishing its totals and waiting for the chunks before it — was tried in round 3:
// 0.710 against 0.683 ms per step at 768³; the two launches below stay.)
__global__ void __launch_bounds__(256) band_count_kernel(BandArgs a, const unsigned char* mask, unsigned long long* count) {
    LSM_TILE_PROLOGUE(a)
    unsigned long long c = 0;
    LSM_TILE_FOR(a, x, y, m, q) c += mask[q] ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

static dim3 tile_grid(const BandArgs& a) { return dim3(a.list ? a.nlist : a.nbx * a.nby * a.nbm); }
// the row-word kernels need 32×8 tiles whose x-lines (with apron) fit a 64-bit word, and `words` words per row in LDS
static bool fast3(const BandArgs& a, int ap, int words) {
    return !a.force_bytes && a.ndim == 3 && a.tx == 32 && a.ty == 8 && a.tx + 2 * ap <= 64 &&
           (long long)words * 8 * (a.ty + 2 * ap) * (a.tm + 2 * ap) + 8 * a.ty * a.tm <= 65536;
}
static bool no_tiles(const BandArgs& a) { return a.list && a.nlist == 0; }
static long long box_bytes(const BandArgs& a, long long ap) {
    return (a.tx + 2 * ap) * (a.ndim == 3 ? a.ty + 2 * ap : 1) * (a.ndim >= 2 ? a.tm + 2 * ap : 1);
}
void launch_band_cut(const BandArgs& a, const void* v, const unsigned char* old_mask, unsigned char* seed, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_cut_kernel, tile_grid(a), dim3(256), 0, s, a, v, old_mask, seed);
}
void launch_band_dilate(const BandArgs& a, const unsigned char* in, unsigned char* out, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_dilate_kernel, tile_grid(a), dim3(256), 0, s, a, in, out);
}
void launch_band_grow(const BandArgs& a, const void* v, const unsigned char* old_mask, int nl, unsigned char* new_mask,
                      unsigned char* tiles, hipStream_t s) {
    if (fast3(a, nl + 1, 5)) {
        const size_t lds = (size_t)5 * 8 * (a.ty + 2 * (nl + 1)) * (a.tm + 2 * (nl + 1));
        if (no_tiles(a)) return;
        hipLaunchKernelGGL(band_grow3_kernel, tile_grid(a), dim3(256), lds, s, a, v, old_mask, nl, new_mask, tiles);
        return;
    }
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_grow_kernel, tile_grid(a), dim3(256), (size_t)box_bytes(a, nl + 1), s, a, v, old_mask, nl, new_mask, tiles);
}
bool band_grow_fits(const BandArgs& a, int nl) { return fast3(a, nl + 1, 5) || box_bytes(a, nl + 1) <= LSM_BAND_LDS; }
void launch_band_copy(const BandArgs& a, const unsigned char* in, unsigned char* out, unsigned char* zero, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_copy_kernel, tile_grid(a), dim3(256), 0, s, a, in, out, zero);
}
void launch_band_copy_values(const BandArgs& a, const unsigned char* mask, const void* src, void* dst, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_copy_values_kernel, tile_grid(a), dim3(256), 0, s, a, mask, src, dst);
}
void launch_band_zero(const BandArgs& a, unsigned char* out, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_zero_kernel, tile_grid(a), dim3(256), 0, s, a, out);
}
void launch_band_status(const unsigned* halo_count, const int* miss, const unsigned* lcounts, const BandStatusCfl& pf, double* out, double ticket,
                        hipStream_t s) {
    hipLaunchKernelGGL(band_status_kernel, dim3(1), dim3(256), 0, s, halo_count, miss, lcounts, pf, out, ticket);
}
void launch_band_extrapolate(const BandArgs& a, const unsigned char* target, unsigned char* halo, const unsigned char* src_mask,
                             const signed char* ring, int nring, int nring_lds, const void* src, void* dst, int* miss,
                             BandEntry* list, unsigned* list_count, unsigned list_cap, hipStream_t s) {
    if (fast3(a, RL, 1)) {
        const size_t lds = (size_t)8 * ((a.ty + 2 * RL) * (a.tm + 2 * RL) + a.ty * a.tm) + 128 * sizeof(unsigned);
        if (no_tiles(a)) return;
        hipLaunchKernelGGL(band_search3_kernel<8>, tile_grid(a), dim3(256), lds, s, a, target, halo, src_mask, ring, nring, nring_lds, src,
                           dst, miss, list, list_count, list_cap);
        return;
    }
    const long long bytes = box_bytes(a, RL);
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_extrapolate_kernel, tile_grid(a), dim3(256), (size_t)(bytes <= LSM_BAND_LDS ? bytes : 0), s, a, target, halo,
                       src_mask, ring, nring, nring_lds, src, dst, miss, list, list_count, list_cap);
}
bool band_bits_fit(const BandArgs& a, int nl) {
    return fast3(a, BAP, 5) && nl >= 0 && nl + 1 <= BAP && a.tm >= BAP && a.ty * a.tm <= 128;
}
void launch_band_bits(const BandArgs& a, const void* v, const unsigned char* mask, unsigned* OB, unsigned* LE, unsigned* GE,
                      const unsigned char* flags_src, unsigned char* flags_dst, unsigned* zero0, hipStream_t s) {
    const unsigned nflags = a.nbx * a.nby * a.nbm, ntile_blocks = a.list ? a.nlist : nflags;
    hipLaunchKernelGGL(band_bits_kernel, dim3(ntile_blocks + (nflags + 4095u) / 4096u), dim3(256), 0, s, a, v, mask, OB, LE, GE, ntile_blocks,
                       flags_src, flags_dst, nflags, zero0);
}
void launch_band_grow_bits(const BandArgs& a, void* v, unsigned char* mask, int nl, const unsigned char* old_tiles, unsigned char* tiles,
                           const unsigned* OB, const unsigned* LE, const unsigned* GE, unsigned* NB, int* miss, hipStream_t s) {
    if (no_tiles(a)) return;
    const size_t lds = (size_t)5 * 8 * (a.ty + 2 * BAP) * (a.tm + 2 * BAP);
    hipLaunchKernelGGL(band_grow_bits_kernel, tile_grid(a), dim3(256), lds, s, a, v, mask, nl, old_tiles, tiles, OB, LE, GE, NB, miss);
}
void launch_band_halo_bits(const BandArgs& a, const unsigned char* tiles, const unsigned* NB, unsigned char* halo, int* miss, BandEntry* list,
                           unsigned* list_count, unsigned list_cap, hipStream_t s) {
    if (no_tiles(a)) return;
    const size_t wpt = (size_t)a.ty * a.tm;
    const size_t lds = (size_t)8 * ((a.ty + 2 * BAP) * (a.tm + 2 * BAP) + wpt) + 128 * sizeof(unsigned) + (2 * wpt + 4) * sizeof(unsigned) + 64 * wpt;
    hipLaunchKernelGGL(band_halo_bits_kernel, tile_grid(a), dim3(256), lds, s, a, tiles, NB, halo, miss, list, list_count, list_cap);
}
void launch_band_apply(const BandArgs& a, const BandEntry* list, const unsigned* list_count, long long n_host, unsigned list_cap,
                       const unsigned char* src_mask, const void* src, void* dst, hipStream_t s) {
    if (n_host == 0) return;
    constexpr int U = 1;      // entries per thread and round (4 measured at 768³: 0.724 against 0.694 ms per step — the gathers saturate, more of them in flight only queue;
                              // fewer, longer-lived workgroups — 8192 / 4096 / 2048 / 1024 instead of 9.2 k — are slower too: 0.536 / 0.538 / 0.541 / 0.549 against 0.534)
    // the length known on the host: U entries per thread, one round; else a grid-stride loop over the capacity
    unsigned blocks = n_host > 0 ? (unsigned)((n_host + 256 * U - 1) / (256 * U)) : (list_cap + 256 * U - 1) / (256 * U);
    const unsigned cap = n_host > 0 ? 65536u : 4096u;
    blocks = blocks > cap ? cap : (blocks < 1 ? 1 : blocks);
    if (a.ndim == 3) hipLaunchKernelGGL((band_apply_kernel<3, U>), dim3(blocks), dim3(256), 0, s, a, list, list_count, n_host, list_cap, src_mask, src, dst);
    else if (a.ndim == 2) hipLaunchKernelGGL((band_apply_kernel<2, U>), dim3(blocks), dim3(256), 0, s, a, list, list_count, n_host, list_cap, src_mask, src, dst);
    else hipLaunchKernelGGL((band_apply_kernel<1, U>), dim3(blocks), dim3(256), 0, s, a, list, list_count, n_host, list_cap, src_mask, src, dst);
}
void launch_band_halo_bc(const BandArgs& a, const BandBcArgs& bc, int d, int r, const unsigned char* band, unsigned char* halo,
                         hipStream_t s) {
    long long total = 2 * LSM_GHOST;
    for (int k = 0; k < a.ndim; ++k)
        if (k != d) total *= k < d ? a.n[k] + 2 * LSM_GHOST : a.n[k];
    long long blocks = (total + 255) / 256;
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(band_halo_bc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, bc, d, r, band, halo);
}
void launch_band_tiles(const BandArgs& a, const unsigned char* mask, unsigned char* tiles, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_tiles_kernel, tile_grid(a), dim3(256), 0, s, a, mask, tiles);
}
void launch_band_work(const BandArgs& a, const unsigned char* active, unsigned char* work, unsigned* zero_face, int* zero_flags, hipStream_t s) {
    const unsigned nt = a.nbx * a.nby * a.nbm;
    hipLaunchKernelGGL(band_work_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, a, active, work, zero_face, zero_flags);
}
void launch_band_lists(const BandArgs& a, const unsigned char* active, const unsigned char* work, int* act_list, int* work_list, unsigned* counts,
                       hipStream_t s) {
    const unsigned ntiles = a.nbx * a.nby * a.nbm;      // counts[2] (face tiles) is accumulated: cleared by the band_work launch before this one
    hipLaunchKernelGGL(band_lists_kernel, dim3((ntiles + LISTS_CHUNK - 1) / LISTS_CHUNK), dim3(1024), 0, s, a, active, work, act_list, work_list, counts);
}
void launch_band_count(const BandArgs& a, const unsigned char* mask, unsigned long long* count, hipStream_t s) {
    if (no_tiles(a)) return;
    hipLaunchKernelGGL(band_count_kernel, tile_grid(a), dim3(256), 0, s, a, mask, count);
}

}  // namespace lsm
