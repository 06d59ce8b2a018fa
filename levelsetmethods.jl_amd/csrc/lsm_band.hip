// lsm_band.hip — device counterpart of NarrowBandMeshField (src/meshfield.jl:314-588):
// the active node set as a byte mask over the dense padded layout, its topological rebuild
// (update_band!), and the affine ghost extrapolation that feeds stencils reaching past the band.
//
// Representation.  Values stay in the dense padded array (288 GB of HBM make the dense backing
// affordable; what the band saves is ARITHMETIC: the stage kernel skips tiles without band nodes
// and stores only on band nodes).  `mask[q] != 0` marks band nodes, same index space as the
// values (mask ghosts are always 0: an out-of-grid index is never a band node).
//
//  * update_band!  (:555-588): cut cells (all 2^N corners in the band, values straddling 0) seed
//    their corners; nlayers von-Neumann dilations grow exactly the L¹ ball of the reference's
//    `grow` offsets; newly active nodes get the extrapolated value from the OLD band.
//  * ϕ[I] for an in-grid non-band node (:481-511): nearest band node by the reference's offset
//    ring (sorted by |off|², ties in column-major order), per-axis one-sided slope from the band
//    neighbours of that node (+ side first), sign-preserving clamp.  Materialised ("band halo")
//    for every non-band node within Chebyshev distance 3 of the band before each stage, so that
//    stencils and the boundary-condition fill read plain array entries.
// Built with -ffp-contract=off.
#include "lsm_internal.h"

namespace lsm {

__global__ void __launch_bounds__(256) band_cut_kernel(BandArgs a, const double* v, const unsigned char* old_mask,
                                                       unsigned char* seed) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    const int nc = 1 << a.ndim;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % a.n[0]), i1 = (int)((t / a.n[0]) % a.n[1]), i2 = (int)(t / ((long long)a.n[0] * a.n[1]));
        // the cell with lower corner I must lie inside the grid
        if (i0 + 1 >= a.n[0] || (a.ndim > 1 && i1 + 1 >= a.n[1]) || (a.ndim > 2 && i2 + 1 >= a.n[2])) continue;
        const long long q = a.origin + i0 + i1 * a.s1 + i2 * a.s2;
        double vmin = __builtin_inf(), vmax = -__builtin_inf();
        bool ok = true;
        for (int c = 0; c < nc; ++c) {
            const long long qc = q + (c & 1) + ((c >> 1) & 1) * a.s1 + ((c >> 2) & 1) * a.s2;
            if (old_mask && !old_mask[qc]) { ok = false; break; }
            const double x = v[qc];
            vmin = x < vmin ? x : vmin;
            vmax = x > vmax ? x : vmax;
        }
        if (!(ok && vmin <= 0.0 && 0.0 <= vmax)) continue;
        for (int c = 0; c < nc; ++c) seed[q + (c & 1) + ((c >> 1) & 1) * a.s1 + ((c >> 2) & 1) * a.s2] = 1;
    }
}

// one von-Neumann (L¹) dilation step on the interior; ghosts of `in` are 0
__global__ void __launch_bounds__(256) band_dilate_kernel(BandArgs a, const unsigned char* in, unsigned char* out) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % a.n[0]), i1 = (int)((t / a.n[0]) % a.n[1]), i2 = (int)(t / ((long long)a.n[0] * a.n[1]));
        const long long q = a.origin + i0 + i1 * a.s1 + i2 * a.s2;
        unsigned char m = in[q] | in[q - 1] | in[q + 1];
        if (a.ndim > 1) m |= in[q - a.s1] | in[q + a.s1];
        if (a.ndim > 2) m |= in[q - a.s2] | in[q + a.s2];
        out[q] = m ? 1 : 0;
    }
}

// Chebyshev dilation of radius r (<= LSM_GHOST) along ONE axis; three passes give the box dilation
__global__ void __launch_bounds__(256) band_box_dilate_kernel(BandArgs a, int dim, int r, const unsigned char* in, unsigned char* out) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    const long long sd = dim == 0 ? 1 : (dim == 1 ? a.s1 : a.s2);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % a.n[0]), i1 = (int)((t / a.n[0]) % a.n[1]), i2 = (int)(t / ((long long)a.n[0] * a.n[1]));
        const long long q = a.origin + i0 + i1 * a.s1 + i2 * a.s2;
        unsigned char m = 0;
        for (int k = -r; k <= r; ++k) m |= in[q + k * sd];
        out[q] = m ? 1 : 0;
    }
}

// _extrapolate_to_ghost (src/meshfield.jl:494-511) for every node with target[q] && !src_mask[q]
__global__ void __launch_bounds__(256) band_extrapolate_kernel(BandArgs a, const unsigned char* target, const unsigned char* src_mask,
                                                               const signed char* ring, int nring, const double* src, double* dst,
                                                               int* miss) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int I[3] = {(int)(t % a.n[0]), (int)((t / a.n[0]) % a.n[1]), (int)(t / ((long long)a.n[0] * a.n[1]))};
        const long long q = a.origin + I[0] + I[1] * a.s1 + I[2] * a.s2;
        if (!target[q] || src_mask[q]) continue;
        // _nearest_band_node: first hit along the distance-sorted offset ring
        int P[3] = {0, 0, 0};
        bool found = false;
        for (int r = 0; r < nring; ++r) {
            const int p0 = I[0] + ring[3 * r], p1 = I[1] + ring[3 * r + 1], p2 = I[2] + ring[3 * r + 2];
            if (p0 < 0 || p0 >= a.n[0] || p1 < 0 || p1 >= a.n[1] || p2 < 0 || p2 >= a.n[2]) continue;
            if (src_mask[a.origin + p0 + p1 * a.s1 + p2 * a.s2]) { P[0] = p0; P[1] = p1; P[2] = p2; found = true; break; }
        }
        if (!found) { atomicOr(miss, 1); continue; }   // the reference throws: farther than the search radius
        const long long qp = a.origin + P[0] + P[1] * a.s1 + P[2] * a.s2;
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

// tile activity for the stage kernel: tile (bx, by, bm) is active iff it contains a band node
__global__ void __launch_bounds__(256) band_tiles_kernel(BandArgs a, int tx, int ty, int mc, unsigned nbx, unsigned nby, unsigned nbm,
                                                         const unsigned char* mask, unsigned char* tiles) {
    const unsigned tile = blockIdx.x;
    if (tile >= nbx * nby * nbm) return;
    const int bx = tile % nbx, by = (tile / nbx) % nby, bm = tile / (nbx * nby);
    const int x0 = bx * tx, y0 = a.ndim == 3 ? by * ty : 0, m0 = bm * mc;
    const int nx = a.n[0], ny = a.ndim == 3 ? a.n[1] : 1, nm = a.ndim >= 2 ? a.n[a.ndim - 1] : 1;
    const long long sy = a.ndim == 3 ? a.s1 : 0, sm = a.ndim == 3 ? a.s2 : (a.ndim == 2 ? a.s1 : 0);
    const int ex = tx, ey = a.ndim == 3 ? ty : 1, em = a.ndim >= 2 ? mc : 1;
    int any = 0;
    for (int e = threadIdx.x; e < ex * ey * em; e += blockDim.x) {
        const int x = x0 + e % ex, y = y0 + (e / ex) % ey, m = m0 + e / (ex * ey);
        if (x < nx && y < ny && m < nm && mask[a.origin + x + y * sy + m * sm]) any = 1;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) tiles[tile] = any ? 1 : 0;
}

__global__ void __launch_bounds__(256) band_count_kernel(BandArgs a, const unsigned char* mask, unsigned long long* count) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    unsigned long long c = 0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % a.n[0]), i1 = (int)((t / a.n[0]) % a.n[1]), i2 = (int)(t / ((long long)a.n[0] * a.n[1]));
        c += mask[a.origin + i0 + i1 * a.s1 + i2 * a.s2] ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

static int nblocks(const BandArgs& a) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    long long b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
void launch_band_cut(const BandArgs& a, const double* v, const unsigned char* old_mask, unsigned char* seed, hipStream_t s) {
    hipLaunchKernelGGL(band_cut_kernel, dim3(nblocks(a)), dim3(256), 0, s, a, v, old_mask, seed);
}
void launch_band_dilate(const BandArgs& a, const unsigned char* in, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(band_dilate_kernel, dim3(nblocks(a)), dim3(256), 0, s, a, in, out);
}
void launch_band_box_dilate(const BandArgs& a, int dim, int r, const unsigned char* in, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(band_box_dilate_kernel, dim3(nblocks(a)), dim3(256), 0, s, a, dim, r, in, out);
}
void launch_band_extrapolate(const BandArgs& a, const unsigned char* target, const unsigned char* src_mask, const signed char* ring,
                             int nring, const double* src, double* dst, int* miss, hipStream_t s) {
    hipLaunchKernelGGL(band_extrapolate_kernel, dim3(nblocks(a)), dim3(256), 0, s, a, target, src_mask, ring, nring, src, dst, miss);
}
void launch_band_tiles(const BandArgs& a, int tx, int ty, int mc, unsigned nbx, unsigned nby, unsigned nbm, const unsigned char* mask,
                       unsigned char* tiles, hipStream_t s) {
    hipLaunchKernelGGL(band_tiles_kernel, dim3(nbx * nby * nbm), dim3(256), 0, s, a, tx, ty, mc, nbx, nby, nbm, mask, tiles);
}
void launch_band_count(const BandArgs& a, const unsigned char* mask, unsigned long long* count, hipStream_t s) {
    hipLaunchKernelGGL(band_count_kernel, dim3(nblocks(a)), dim3(256), 0, s, a, mask, count);
}

}  // namespace lsm
