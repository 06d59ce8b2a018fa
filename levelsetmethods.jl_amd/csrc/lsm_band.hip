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
#include "lsm_internal.h"

#define LSM_BAND_LDS 20480   // bytes of LDS for a tile's mask + search apron ((32+12)·(8+12)·(8+12) = 17600 in 3-D)

namespace lsm {

// iterate the nodes of this block's tile; returns false if the tile is skipped
#define LSM_TILE_PROLOGUE(a)                                                                         \
    const unsigned tile = blockIdx.x;                                                                \
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

__global__ void __launch_bounds__(256) band_cut_kernel(BandArgs a, const double* v, const unsigned char* old_mask,
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
            const double xv = v[qc];
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

// Chebyshev dilation of radius r (<= LSM_GHOST) along ONE axis; N passes give the box dilation
__global__ void __launch_bounds__(256) band_box_dilate_kernel(BandArgs a, int dim, int r, const unsigned char* in, unsigned char* out) {
    LSM_TILE_PROLOGUE(a)
    const long long sd = dim == 0 ? 1 : (dim == 1 ? a.s1 : a.s2);
    LSM_TILE_FOR(a, x, y, m, q) {
        unsigned char o = 0;
        for (int k = -r; k <= r; ++k) o |= in[q + k * sd];
        out[q] = o ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) band_copy_kernel(BandArgs a, const unsigned char* in, unsigned char* out) {
    LSM_TILE_PROLOGUE(a)
    LSM_TILE_FOR(a, x, y, m, q) out[q] = in[q];
}

// _extrapolate_to_ghost (src/meshfield.jl:494-511) for every node with target[q] && !src_mask[q]
// With `list` the (node, nearest band node) pairs are appended for band_apply_kernel: the nearest node
// depends on the mask only, so the search runs once per band update and every stage input is then
// filled by a plain gather.
__global__ void __launch_bounds__(256) band_extrapolate_kernel(BandArgs a, const unsigned char* target, const unsigned char* src_mask,
                                                               const signed char* ring, int nring, const double* src, double* dst,
                                                               int* miss, BandEntry* list, unsigned* list_count, unsigned list_cap) {
    LSM_TILE_PROLOGUE(a)
    {   // nothing to do in most tiles: skip them before staging anything
        int any = 0;
        LSM_TILE_FOR(a, x, y, m, q) any |= (target[q] && !src_mask[q]) ? 1 : 0;
        if (!__syncthreads_or(any)) return;
    }
    // The ring search probes up to (2R+1)^N mask bytes per node; stage the tile's mask with an apron of R
    // nodes in LDS once (out-of-grid entries read as 0) so that every probe is an LDS byte read.
    constexpr int R = 6;
    __shared__ unsigned char lmask[LSM_BAND_LDS];
    const int ax = ex_ + 2 * R, ay = a.ndim == 3 ? ey_ + 2 * R : 1, am = a.ndim >= 2 ? em_ + 2 * R : 1;
    const bool staged = ax * ay * am <= LSM_BAND_LDS;
    if (staged) {
        for (int e = threadIdx.x; e < ax * ay * am; e += blockDim.x) {
            const int lx = e % ax, ly = (e / ax) % ay, lm = e / (ax * ay);
            const int gx = x0_ - R + lx, gy = a.ndim == 3 ? y0_ - R + ly : 0, gm = a.ndim >= 2 ? m0_ - R + lm : 0;
            unsigned char v = 0;
            if (gx >= 0 && gx < nx_ && gy >= 0 && gy < ny_ && gm >= 0 && gm < nm_) v = src_mask[a.origin + gx + gy * sy_ + gm * sm_];
            lmask[e] = v;
        }
        __syncthreads();
    }
    LSM_TILE_FOR(a, x, y, m, q) {
        if (!target[q] || src_mask[q]) continue;
        int I[3] = {x, 0, 0};
        if (a.ndim == 2) I[1] = m;
        if (a.ndim == 3) { I[1] = y; I[2] = m; }
        // _nearest_band_node: first hit along the distance-sorted offset ring
        int P[3] = {0, 0, 0};
        bool found = false;
        if (staged) {
            // tile-local coordinates of I inside the staged block (ring offsets are in field dimensions)
            const int lx = x - x0_ + R, ly = a.ndim == 3 ? y - y0_ + R : 0, lm = a.ndim >= 2 ? m - m0_ + R : 0;
            for (int r = 0; r < nring; ++r) {
                const int o0 = ring[3 * r], o1 = ring[3 * r + 1], o2 = ring[3 * r + 2];
                const int oy = a.ndim == 3 ? o1 : 0, om = a.ndim == 3 ? o2 : (a.ndim == 2 ? o1 : 0);
                if (lmask[(lx + o0) + ax * ((ly + oy) + ay * (lm + om))]) {
                    P[0] = I[0] + o0; P[1] = I[1] + o1; P[2] = I[2] + o2; found = true; break;
                }
            }
        } else {
            for (int r = 0; r < nring; ++r) {
                const int p0 = I[0] + ring[3 * r], p1 = I[1] + ring[3 * r + 1], p2 = I[2] + ring[3 * r + 2];
                if (p0 < 0 || p0 >= a.n[0] || p1 < 0 || p1 >= a.n[1] || p2 < 0 || p2 >= a.n[2]) continue;
                if (src_mask[a.origin + p0 + p1 * a.s1 + p2 * a.s2]) { P[0] = p0; P[1] = p1; P[2] = p2; found = true; break; }
            }
        }
        if (!found) { atomicOr(miss, 1); continue; }   // the reference throws: farther than the search radius
        const long long qp = a.origin + P[0] + P[1] * a.s1 + P[2] * a.s2;
        if (list) {
            const unsigned k = atomicAdd(list_count, 1u);
            if (k < list_cap) {
                BandEntry e;
                e.q = q; e.rel = (int)(qp - q);
                e.d[0] = (signed char)(I[0] - P[0]); e.d[1] = (signed char)(I[1] - P[1]); e.d[2] = (signed char)(I[2] - P[2]); e.d[3] = 0;
                list[k] = e;
            }
        }
        if (!dst) continue;
        const double phiP = src[qp];
        double val = phiP;
        for (int d = 0; d < a.ndim; ++d) {
            const int delta = I[d] - P[d];
            if (delta == 0) continue;
            const long long sd = d == 0 ? 1 : (d == 1 ? a.s1 : a.s2);
            double slope = 0.0;                                   // _axis_slope: + neighbour first, then -
            if (P[d] + 1 < a.n[d] && src_mask[qp + sd]) slope = src[qp + sd] - phiP;
            else if (P[d] - 1 >= 0 && src_mask[qp - sd]) slope = phiP - src[qp - sd];
            val += (double)delta * slope;
        }
        // never let extrapolation invent a sign change far from the band
        const double sv = val > 0 ? 1.0 : (val < 0 ? -1.0 : val), sp = phiP > 0 ? 1.0 : (phiP < 0 ? -1.0 : phiP);
        dst[q] = (phiP == 0.0 || sv == sp) ? val : phiP;
    }
}

// _extrapolate_to_ghost from a precomputed (node, nearest band node) list: one thread per entry
__global__ void __launch_bounds__(256) band_apply_kernel(BandArgs a, const BandEntry* list, const unsigned* list_count, unsigned list_cap,
                                                         const unsigned char* src_mask, const double* src, double* dst) {
    unsigned n = *list_count;
    n = n < list_cap ? n : list_cap;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const BandEntry e = list[i];
        const long long qp = e.q + e.rel;
        const double phiP = src[qp];
        double val = phiP;
        for (int d = 0; d < a.ndim; ++d) {
            const int delta = e.d[d];
            if (delta == 0) continue;
            const long long sd = d == 0 ? 1 : (d == 1 ? a.s1 : a.s2);
            double slope = 0.0;                                   // mask ghosts are 0: no bounds test needed
            if (src_mask[qp + sd]) slope = src[qp + sd] - phiP;
            else if (src_mask[qp - sd]) slope = phiP - src[qp - sd];
            val += (double)delta * slope;
        }
        const double sv = val > 0 ? 1.0 : (val < 0 ? -1.0 : val), sp = phiP > 0 ? 1.0 : (phiP < 0 ? -1.0 : phiP);
        dst[e.q] = (phiP == 0.0 || sv == sp) ? val : phiP;
    }
}

// halo of the band = what stencils centred on band nodes can read: up to r nodes along each axis
// (WENO5 / ENO2 lines) and the 3^N box (curvature's edge diagonals).  In-grid part here; the in-grid
// nodes that out-of-grid stencil positions resolve to are added by band_halo_bc_kernel.
__global__ void __launch_bounds__(256) band_cross_kernel(BandArgs a, int r, const unsigned char* in, unsigned char* out) {
    LSM_TILE_PROLOGUE(a)
    LSM_TILE_FOR(a, x, y, m, q) {
        unsigned char o = 0;
        for (int k = -r; k <= r; ++k) {
            o |= in[q + k];
            if (a.ndim > 1) o |= in[q + k * a.s1];
            if (a.ndim > 2) o |= in[q + k * a.s2];
        }
        for (int k2 = -1; k2 <= 1; ++k2)
            for (int k1 = -1; k1 <= 1; ++k1)
                for (int k0 = -1; k0 <= 1; ++k0) {
                    if ((a.ndim < 2 && k1) || (a.ndim < 3 && k2)) continue;
                    o |= in[q + k0 + k1 * a.s1 + k2 * a.s2];
                }
        out[q] = o ? 1 : 0;
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
    if (a.work && !a.work[blockIdx.x]) {
        if (threadIdx.x == 0) tiles[blockIdx.x] = 0;
        return;
    }
    LSM_TILE_PROLOGUE(a)
    int any = 0;
    LSM_TILE_FOR(a, x, y, m, q) any |= mask[q] ? 1 : 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) tiles[tile] = any ? 1 : 0;
}

// work[t] = OR of active over the 3^N tile neighbourhood of t
__global__ void __launch_bounds__(256) band_work_kernel(BandArgs a, const unsigned char* active, unsigned char* work) {
    const unsigned nt = a.nbx * a.nby * a.nbm;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
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

__global__ void __launch_bounds__(256) band_count_kernel(BandArgs a, const unsigned char* mask, unsigned long long* count) {
    LSM_TILE_PROLOGUE(a)
    unsigned long long c = 0;
    LSM_TILE_FOR(a, x, y, m, q) c += mask[q] ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

static dim3 tile_grid(const BandArgs& a) { return dim3(a.nbx * a.nby * a.nbm); }
void launch_band_cut(const BandArgs& a, const double* v, const unsigned char* old_mask, unsigned char* seed, hipStream_t s) {
    hipLaunchKernelGGL(band_cut_kernel, tile_grid(a), dim3(256), 0, s, a, v, old_mask, seed);
}
void launch_band_dilate(const BandArgs& a, const unsigned char* in, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(band_dilate_kernel, tile_grid(a), dim3(256), 0, s, a, in, out);
}
void launch_band_box_dilate(const BandArgs& a, int dim, int r, const unsigned char* in, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(band_box_dilate_kernel, tile_grid(a), dim3(256), 0, s, a, dim, r, in, out);
}
void launch_band_copy(const BandArgs& a, const unsigned char* in, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(band_copy_kernel, tile_grid(a), dim3(256), 0, s, a, in, out);
}
void launch_band_extrapolate(const BandArgs& a, const unsigned char* target, const unsigned char* src_mask, const signed char* ring,
                             int nring, const double* src, double* dst, int* miss, BandEntry* list, unsigned* list_count,
                             unsigned list_cap, hipStream_t s) {
    hipLaunchKernelGGL(band_extrapolate_kernel, tile_grid(a), dim3(256), 0, s, a, target, src_mask, ring, nring, src, dst, miss, list,
                       list_count, list_cap);
}
void launch_band_apply(const BandArgs& a, const BandEntry* list, const unsigned* list_count, unsigned list_cap,
                       const unsigned char* src_mask, const double* src, double* dst, hipStream_t s) {
    unsigned blocks = (list_cap + 255) / 256;
    blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(band_apply_kernel, dim3(blocks), dim3(256), 0, s, a, list, list_count, list_cap, src_mask, src, dst);
}
void launch_band_cross(const BandArgs& a, int r, const unsigned char* in, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(band_cross_kernel, tile_grid(a), dim3(256), 0, s, a, r, in, out);
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
    hipLaunchKernelGGL(band_tiles_kernel, tile_grid(a), dim3(256), 0, s, a, mask, tiles);
}
void launch_band_work(const BandArgs& a, const unsigned char* active, unsigned char* work, hipStream_t s) {
    const unsigned nt = a.nbx * a.nby * a.nbm;
    hipLaunchKernelGGL(band_work_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, a, active, work);
}
void launch_band_count(const BandArgs& a, const unsigned char* mask, unsigned long long* count, hipStream_t s) {
    hipLaunchKernelGGL(band_count_kernel, tile_grid(a), dim3(256), 0, s, a, mask, count);
}

}  // namespace lsm
