// lsm_aux.hip — the small kernels around the stage kernel: ghost-layer fill, CFL reduction,
// extrema, Eikonal sign map.  Built with -ffp-contract=off: ghost values and the CFL minimum are
// bit-for-bit those of the reference arithmetic (they cost nothing next to a stage).
#include "lsm_internal.h"

namespace lsm {

// ---------------------------------------------------------------------------------------------
// Ghost fill of ONE dimension: _getindexbc + bc_stencil (src/meshfield.jl:248-260,
// src/boundaryconditions.jl:107-153).  Filling dims 1..N in order, each pass covering the ghosts
// already written by the lower dimensions, reproduces the recursion's corner composition exactly:
//   ghost(i_g, j_g) = Σ_j w2_j · ( Σ_i w1_i ϕ[I_i, J_j] ).
// One thread per transverse position; it writes the 2·LSM_GHOST ghosts of its line.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ghost_fill_kernel(const GhostArgs a, int ndim) {
    const int d = a.dim;
    // transverse extents: lower dims incl. ghosts, higher dims interior
    int lo[3], ext[3];
    for (int e = 0; e < 3; ++e) {
        const int g = e < ndim ? LSM_GHOST : 0;
        if (e < d) { lo[e] = -g; ext[e] = a.n[e] + 2 * g; }
        else       { lo[e] = 0;  ext[e] = a.n[e]; }
    }
    ext[d] = 1;
    const long long total = (long long)ext[0] * ext[1] * ext[2];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int I[3];
    I[0] = lo[0] + (int)(t % ext[0]);
    I[1] = lo[1] + (int)((t / ext[0]) % ext[1]);
    I[2] = lo[2] + (int)(t / ((long long)ext[0] * ext[1]));
    const long long sd = d == 0 ? 1 : (d == 1 ? a.s1 : a.s2);
    I[d] = 0;
    const long long line0 = a.origin + I[0] + I[1] * a.s1 + I[2] * a.s2;   // node 0 of this line
    auto line = [&](long long off) { return ld_val(a.v, line0 + off, a.f32); };
    const int n = a.n[d];
    for (int side = 0; side < 2; ++side) {
        const int kind = a.kind[side];
        if (kind == LSM_BC_NONE) continue;
        const int b = side == 0 ? 0 : n - 1;
        const int dir = side == 0 ? 1 : -1;
        for (int k = 1; k <= LSM_GHOST; ++k) {
            double acc = 0.0;
            if (kind == LSM_BC_PERIODIC) {
                const int j = side == 0 ? (n - 1) - k : k;          // period n-1 (src/boundaryconditions.jl:107-119)
                acc += 1.0 * line(j * sd);
            } else if (kind == LSM_BC_EXTRAPOLATION) {
                for (int j = 0; j <= a.degree[side]; ++j) acc += a.w[side][k - 1][j] * line((b + dir * j) * sd);
            } else {
                acc += 1.0 * line((b + dir * k) * sd);              // mirror about the boundary node
            }
            st_val(a.v, line0 + (b - dir * k) * sd, a.f32, acc);
        }
    }
}

void launch_ghost_fill(int ndim, const GhostArgs& a, hipStream_t s) {
    long long total = 1;
    for (int e = 0; e < 3; ++e) {
        if (e == a.dim) continue;
        const int g = e < ndim ? LSM_GHOST : 0;
        total *= (e < a.dim) ? a.n[e] + 2 * g : a.n[e];
    }
    const int block = 256;
    const unsigned grid = (unsigned)((total + block - 1) / block);
    hipLaunchKernelGGL(ghost_fill_kernel, dim3(grid), dim3(block), 0, s, a, ndim);
}

// ---------------------------------------------------------------------------------------------
// Fused ghost fill: ALL dimensions in one launch, one thread per ghost node.  A ghost that is out
// of range in several dimensions evaluates the nested sums of the recursion directly,
//   ghost = Σ_jz w_jz ( Σ_jy w_jy ( Σ_jx w_jx ϕ[interior] ) ),
// in exactly the order of _getindexbc (highest dimension outermost, acc = 0; acc += w·value), so
// the values are bit-identical to three sequential per-dimension passes — but every thread reads
// interior nodes only, hence no dependency between threads and a single launch.
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double ghost_resolve(const GhostAllArgs& a, int I0, int I1, int I2) {
    if constexpr (D < 0) {
        return ld_val(a.v, a.origin + I0 + I1 * a.s1 + I2 * a.s2, a.f32);
    } else {
        const int i = D == 0 ? I0 : (D == 1 ? I1 : I2);
        const int n = a.n[D];
        if (i >= 0 && i < n) return ghost_resolve<D - 1>(a, I0, I1, I2);
        const int side = i < 0 ? 0 : 1;
        const int k = side == 0 ? -i : i - (n - 1);
        const int b = side == 0 ? 0 : n - 1;
        const int dir = side == 0 ? 1 : -1;
        const int kind = side ? a.kind[D][1] : a.kind[D][0];
        const int deg = side ? a.degree[D][1] : a.degree[D][0];
        const double* w = a.w + ((D * 2 + side) * LSM_GHOST + (k - 1)) * 8;
        auto at = [&](int j) {
            return D == 0 ? ghost_resolve<D - 1>(a, j, I1, I2) : (D == 1 ? ghost_resolve<D - 1>(a, I0, j, I2) : ghost_resolve<D - 1>(a, I0, I1, j));
        };
        double acc = 0.0;
        if (kind == LSM_BC_PERIODIC) {
            acc += 1.0 * at(side == 0 ? (n - 1) - k : k);
        } else if (kind == LSM_BC_EXTRAPOLATION) {
            for (int j = 0; j <= deg; ++j) acc += w[j] * at(b + dir * j);
        } else {
            acc += 1.0 * at(b + dir * k);
        }
        return acc;
    }
}

// IT = the integer type of the ghost-node enumeration: unsigned (32-bit divisions; every grid whose ghost count fits —
// 4.8 M at 512³) or long long.  With 64-bit divisions the index arithmetic, not the 120 MB the kernel moves, set its time.
template <int NDIM, class IT>
__global__ void __launch_bounds__(256) ghost_fill_all_kernel(const GhostAllArgs a) {
    const int G = a.depth;       // layers to fill (<= LSM_GHOST, the layout's padding)
    const int P0 = a.n[0] + 2 * G, P1 = NDIM > 1 ? a.n[1] + 2 * G : 1;
    // region sizes: (A) ghosts of the last dim over the full padded lower dims, (B) [3-D only] y ghosts
    // over padded x and interior z, (C) x ghosts over interior y,z
    const bool lastL = a.fill_last && a.kind[NDIM - 1][0] != LSM_BC_NONE, lastR = a.fill_last && a.kind[NDIM - 1][1] != LSM_BC_NONE;
    const int nlastg = (lastL ? G : 0) + (lastR ? G : 0);
    const int np = a.me - a.mb;   // planes (rows in 2-D) of the last dimension handled by regions B and C
    const bool cL = true;
    const int ncx = 2 * G;
    IT nA, nB, nC;
    if (NDIM == 1) { nA = nlastg; nB = 0; nC = 0; }
    else if (NDIM == 2) { nA = (IT)nlastg * P0; nB = 0; nC = (IT)np * ncx; }
    else { nA = (IT)nlastg * P0 * P1; nB = (IT)np * 2 * G * P0; nC = (IT)np * a.n[1] * ncx; }
    if (a.skip_x && NDIM > 1) nC = 0;
    if (a.skip_y && NDIM == 3) nB = 0;
    const IT t = (IT)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nA + nB + nC) return;
    int I0 = 0, I1 = 0, I2 = 0;
    auto ghost_index = [&](int g, int n, bool left_on) {   // g-th ghost of a dim: left block first (if present)
        if (left_on && g < G) return g - G;
        return n + (g - (left_on ? G : 0));
    };
    if (t < nA) {
        if (NDIM == 1) { I0 = ghost_index((int)t, a.n[0], lastL); }
        else if (NDIM == 2) { I0 = (int)(t % P0) - G; I1 = ghost_index((int)(t / P0), a.n[1], lastL); }
        else { I0 = (int)(t % P0) - G; I1 = (int)((t / P0) % P1) - G; I2 = ghost_index((int)(t / ((IT)P0 * P1)), a.n[2], lastL); }
    } else if (t < nA + nB) {   // 3-D only
        const IT u = t - nA;
        I0 = (int)(u % P0) - G;
        const int g = (int)((u / P0) % (2 * G));
        I1 = g < G ? g - G : a.n[1] + (g - G);
        I2 = a.mb + (int)(u / ((IT)P0 * 2 * G));
    } else {
        const IT u = t - nA - nB;
        const int g = (int)(u % (ncx > 0 ? ncx : 1));
        I0 = ghost_index(g, a.n[0], cL);
        const int nc1 = ncx > 0 ? ncx : 1;
        if (NDIM == 2) { I1 = a.mb + (int)(u / nc1); }
        else { I1 = (int)((u / nc1) % a.n[1]); I2 = a.mb + (int)(u / ((IT)nc1 * a.n[1])); }
    }
    const double val = ghost_resolve<NDIM - 1>(a, I0, I1, I2);
    st_val(a.v, a.origin + I0 + I1 * a.s1 + I2 * a.s2, a.f32, val);
}

// Row-based form of the same fill (2-D and 3-D): a block is a 256-wide piece of ONE row of ghost nodes, found from
// blockIdx with compares and a division by 6 — no division by a runtime extent in any lane (the flat enumeration above spends
// most of its 40 µs at 512³ on them).  blockIdx.y = line:
//   (A) the ghost planes of the last dimension: nlastg x P1 rows of P0 nodes (the padded row, corners included),
//   (B) 3-D: the 2G ghost rows of dimension 1 of every plane mb..me-1, P0 nodes each,
//   (C) the 2G ghost nodes at the ends of the interior rows: one line per plane (3-D) or one line in all (2-D); its blocks
//       walk the rows, 42 rows x 6 nodes at a time.
// COPY: every face that is resolved copies ONE node (periodic wrap, symmetry mirror, degree-0 extrapolation = NeumannBC):
// the nested sums collapse to 0.0 + 1.0·ϕ[source] (the "+ 0.0" is the reference's `acc = 0; acc += w·value`: it turns -0.0
// into +0.0 and nothing else), one load per ghost node; bit-identical to the flat kernel, which keeps the weighted faces.
template <int D>
__device__ __forceinline__ int ghost_copy_source(const GhostAllArgs& a, int i) {
    const int n = a.n[D];
    if (i >= 0 && i < n) return i;
    const int side = i < 0 ? 0 : 1;
    const int k = side == 0 ? -i : i - (n - 1);
    const int kind = side ? a.kind[D][1] : a.kind[D][0];
    if (kind == LSM_BC_PERIODIC) return side == 0 ? (n - 1) - k : k;
    if (kind == LSM_BC_EXTRAPOLATION) return side == 0 ? 0 : n - 1;      // degree 0
    return side == 0 ? k : (n - 1) - k;                                  // symmetry
}

template <int NDIM, bool COPY>
__global__ void __launch_bounds__(256) ghost_rows_kernel(const GhostAllArgs a) {
    const int G = a.depth;       // layers to fill (<= LSM_GHOST)
    const int P0 = a.n[0] + 2 * G, P1 = NDIM == 3 ? a.n[1] + 2 * G : 1;
    const int nl = a.n[NDIM - 1];
    const bool lastL = a.fill_last && a.kind[NDIM - 1][0] != LSM_BC_NONE, lastR = a.fill_last && a.kind[NDIM - 1][1] != LSM_BC_NONE;
    const int nlastg = (lastL ? G : 0) + (lastR ? G : 0);
    const int np = a.me - a.mb;
    const int nA = nlastg * P1, nB = NDIM == 3 && !a.skip_y ? np * 2 * G : 0;
    auto put = [&](int I0, int I1, int I2) {
        static_assert(COPY, "the row form serves copy-type faces");
        const int s0 = ghost_copy_source<0>(a, I0), s1 = ghost_copy_source<1>(a, I1), s2 = NDIM == 3 ? ghost_copy_source<2>(a, I2) : 0;
        const double val = 0.0 + 1.0 * ld_val(a.v, a.origin + s0 + s1 * a.s1 + s2 * a.s2, a.f32);
        st_val(a.v, a.origin + I0 + I1 * a.s1 + I2 * a.s2, a.f32, val);
    };
    int line = (int)blockIdx.y;
    if (line < nA + nB) {
        int I1 = 0, I2 = 0;
        if (line < nA) {
            int gz = 0;
            while (line >= P1) { line -= P1; ++gz; }                     // nlastg <= 6 turns of a scalar loop
            const int il = (lastL && gz < G) ? gz - G : nl + (gz - (lastL ? G : 0));
            if (NDIM == 3) { I1 = line - G; I2 = il; } else { I1 = il; }
        } else {
            const int l = line - nA, z = l / (2 * G), g = l % (2 * G);
            I1 = g < G ? g - G : a.n[1] + (g - G);
            I2 = a.mb + z;
        }
        for (int x = (int)blockIdx.x * 256 + (int)threadIdx.x; x < P0; x += 256 * (int)gridDim.x) put(x - G, I1, I2);
    } else {
        const int l = line - nA - nB;                                    // 3-D: plane; 2-D: 0
        const int nrows = NDIM == 3 ? a.n[1] : np;
        const int g = (int)threadIdx.x % (2 * G);
        const int I0 = g < G ? g - G : a.n[0] + (g - G);
        const int rpb = 256 / (2 * G);                                   // rows per block: 42 with three layers
        if ((int)threadIdx.x >= rpb * 2 * G) return;
        for (int r = (int)blockIdx.x * rpb + (int)threadIdx.x / (2 * G); r < nrows; r += rpb * (int)gridDim.x) {
            if (NDIM == 3) put(I0, r, a.mb + l); else put(I0, a.mb + r, 0);
        }
    }
}

static bool ghost_rows_launch(int ndim, const GhostAllArgs& a, hipStream_t s) {
    if (ndim < 2) return false;
    const int G = a.depth;
    const long long P0 = a.n[0] + 2 * G, P1 = ndim == 3 ? a.n[1] + 2 * G : 1;
    const int nlastg = a.fill_last ? (a.kind[ndim - 1][0] != LSM_BC_NONE ? G : 0) + (a.kind[ndim - 1][1] != LSM_BC_NONE ? G : 0) : 0;
    const long long np = a.me - a.mb;
    const long long lines = nlastg * P1 + (ndim == 3 ? (a.skip_y ? 0 : np * 2 * G) + (a.skip_x ? 0 : np) : (np > 0 && !a.skip_x ? 1 : 0));
    if (lines <= 0) return true;
    if (lines > 65535) return false;
    bool copy = true;
    for (int d = 0; d < ndim; ++d)
        for (int sd = 0; sd < 2; ++sd) {
            const int k = a.kind[d][sd];
            if (k == LSM_BC_NONE) continue;
            if (k == LSM_BC_EXTRAPOLATION && a.degree[d][sd] != 0) copy = false;
        }
    // weighted faces (extrapolation of degree >= 1) stay with the flat kernel.  Tried in row form twice: with the nested sums
    // inlined as a recursion into both loops (166 registers, spills: 289 µs against 75 µs at 512³ with ExtrapolationBC(2)) and, in
    // round 3, as runtime loops with the face ghosts' loads issued together (low registers: 65 µs against the flat kernel's 57 µs
    // at two layers) — the time is the scattered sectors of the row ends, not the index arithmetic.
    if (!copy) return false;
    const dim3 grid((unsigned)((P0 + 255) / 256), (unsigned)lines);
    if (ndim == 2) hipLaunchKernelGGL((ghost_rows_kernel<2, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ghost_rows_kernel<3, true>), grid, dim3(256), 0, s, a);
    return true;
}

void launch_ghost_fill_all(int ndim, const GhostAllArgs& a, hipStream_t s) {
    if (ghost_rows_launch(ndim, a, s)) return;
    const int G = a.depth;
    const long long P0 = a.n[0] + 2 * G, P1 = ndim > 1 ? a.n[1] + 2 * G : 1;
    const int nlastg = a.fill_last ? (a.kind[ndim - 1][0] != LSM_BC_NONE ? G : 0) + (a.kind[ndim - 1][1] != LSM_BC_NONE ? G : 0) : 0;
    const long long np = a.me - a.mb;
    long long total;
    const int ncx = 2 * G;
    const long long ncxs = a.skip_x ? 0 : ncx;
    if (ndim == 1) total = nlastg;
    else if (ndim == 2) total = nlastg * P0 + np * ncxs;
    else total = nlastg * P0 * P1 + (a.skip_y ? 0 : np * 2 * G * P0) + np * a.n[1] * ncxs;
    if (total <= 0) return;
    const unsigned grid = (unsigned)((total + 255) / 256);
    const bool narrow = total < (1ll << 31) - 512;
    if (ndim == 1) hipLaunchKernelGGL((ghost_fill_all_kernel<1, unsigned>), dim3(grid), dim3(256), 0, s, a);
    else if (ndim == 2) {
        if (narrow) hipLaunchKernelGGL((ghost_fill_all_kernel<2, unsigned>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ghost_fill_all_kernel<2, long long>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (narrow) hipLaunchKernelGGL((ghost_fill_all_kernel<3, unsigned>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ghost_fill_all_kernel<3, long long>), dim3(grid), dim3(256), 0, s, a);
    }
}

// ---------------------------------------------------------------------------------------------
// NaN-propagating min reduction helpers (Julia's min(x, NaN) = NaN, src/levelsetterms.jl:31-38).
// The minimum over non-NaN values and an "any NaN" flag are reduced separately.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

#define CFL_MAX_CHUNK 512   // planes per march chunk staged in LDS (launcher keeps chunk <= this)

// Per-node CFL of one term (src/levelsetterms.jl:90-96,123-127,172-178).  The node formulas are
//   advection 1/Σ_d(|u_d|/h_d),  normal motion 1/Σ_d(|v|/h_d),  curvature Δx²/(2|b|),
// i.e. a correctly-rounded, monotonically DEcreasing function f of a per-node quantity s
// (s = Σ|u_d|/h_d, resp. |b|).  Because IEEE division is monotone, min_I f(s_I) == f(max_I s_I)
// bit for bit, so the kernel reduces max s (plus an any-NaN flag: Julia's min propagates NaN) and
// the final kernel applies f once.
//
// The three IEEE divisions per node dominate this kernel (VALU-bound), so analytic coefficients
// use TWO passes: pass 0 reduces an approximate s~ = Σ|u_d|·(1/h_d) (|s~ - s| <= 1e-15·s); pass 1
// recomputes s~ and evaluates the exact, division-based s only at nodes with s~ >= (1-1e-13)·max s~.
// The true arg-max is always among those candidates, so the result is still exact; typically a
// handful of nodes take the slow branch.  FIELD coefficients (HBM-bound) use the exact pass alone.
template <int NDIM, int TKIND, int CKIND>
__global__ void __launch_bounds__(256) cfl_kernel(const CflArgs a, int pass, const double* thresh_ptr) {
    constexpr int NCOMP = TKIND == LSM_TERM_ADVECTION ? NDIM : 1;
    const double thresh = pass >= 1 && thresh_ptr ? *thresh_ptr : 0.0;   // pass 2 = pass 1 + record the candidates
    const double ih0 = 1.0 / a.h[0], ih1 = 1.0 / a.h[1], ih2 = 1.0 / a.h[2];
    // one thread per column of the LAST dimension (x fastest across lanes: coalesced FIELD reads);
    // everything that does not depend on the last index is hoisted out of the march.
    const int nlast = a.n[NDIM - 1];
    const long long ncol = NDIM == 1 ? a.n[0] : (NDIM == 2 ? a.n[0] : (long long)a.n[0] * a.n[1]);
    const long long slast = NDIM == 1 ? 1 : (NDIM == 2 ? a.s1 : a.s2);
    double best = 0.0;   // s >= 0
    int sawnan = 0;
    // the march range is split over gridDim.y chunks for occupancy; the march-axis table entries of
    // the chunk are staged in LDS once (wave-uniform look-ups inside the loop become LDS broadcasts)
    const int mcount = NDIM == 1 ? 1 : nlast;
    const int chunk = (mcount + (int)gridDim.y - 1) / (int)gridDim.y;
    const int mb = (int)blockIdx.y * chunk, me = mb + chunk < mcount ? mb + chunk : mcount;
    __shared__ double tab[3 * CFL_MAX_CHUNK];
    const CoeffArgs& c = a.coeff;
    if constexpr (CKIND == LSM_COEFF_SEPARABLE && NDIM > 1) {
        const int toff = (NDIM == 3 ? a.gn[0] + a.gn[1] : a.gn[0]) + a.goff[NDIM - 1];
#pragma unroll
        for (int k = 0; k < NCOMP; ++k)
            for (int j = threadIdx.x; j < me - mb; j += blockDim.x) tab[k * CFL_MAX_CHUNK + j] = c.sep[k][toff + mb + j];
        __syncthreads();
    }
    for (long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x; col < ncol; col += (long long)gridDim.x * blockDim.x) {
        const int i0 = NDIM == 3 ? (int)(col % a.n[0]) : (int)col;
        const int i1 = NDIM == 3 ? (int)(col / a.n[0]) : 0;
        const int g0 = i0 + a.goff[0], g1 = i1 + a.goff[1];
        const long long cbase = a.origin + i0 + (NDIM == 3 ? i1 * a.s1 : 0);
        double pre[3] = {0, 0, 0};
        if constexpr (CKIND == LSM_COEFF_SEPARABLE) {
#pragma unroll
            for (int k = 0; k < NCOMP; ++k) {
                double p = c.sep[k][g0];
                if (NDIM == 3) p = p * c.sep[k][a.gn[0] + g1];
                pre[k] = p;
            }
        } else if constexpr (CKIND == LSM_COEFF_ROTATION) {
            const double x1 = a.lc[0] + (double)g0 * a.h[0];
            pre[1] = c.v[0] * (x1 - c.v[1]);
            if (NDIM == 3) pre[0] = -(c.v[0] * ((a.lc[1] + (double)g1 * a.h[1]) - c.v[2]));
        }
        // narrow band: the column crosses tiles of tm planes; tiles without band nodes are stepped over
        const unsigned tcol = a.tile_active ? (unsigned)(i0 / a.tx) + a.nbx * (NDIM == 3 ? (unsigned)(i1 / a.ty) : 0u) : 0u;
        for (int m = mb; m < me; ++m) {
            if (a.tile_active && NDIM > 1 && !a.tile_active[tcol + a.nbx * a.nby * (unsigned)(m / a.tm)]) {
                m = (m / a.tm + 1) * a.tm - 1;
                continue;
            }
            if (a.mask && !a.mask[cbase + (NDIM == 1 ? 0 : m * slast)]) continue;   // narrow band: active nodes only
            double u[3] = {0, 0, 0};
            if constexpr (CKIND == LSM_COEFF_CONST) {
#pragma unroll
                for (int k = 0; k < NCOMP; ++k) u[k] = c.v[k];
            } else if constexpr (CKIND == LSM_COEFF_SEPARABLE) {
#pragma unroll
                for (int k = 0; k < NCOMP; ++k) {
                    double p = pre[k];
                    if (NDIM > 1) p = p * tab[k * CFL_MAX_CHUNK + (m - mb)];
                    u[k] = p * c.tfac;
                }
            } else if constexpr (CKIND == LSM_COEFF_ROTATION) {
                const int gm = m + a.goff[NDIM - 1];
                u[0] = NDIM == 2 ? -(c.v[0] * ((a.lc[1] + (double)gm * a.h[1]) - c.v[2])) : pre[0];
                u[1] = pre[1];
            } else {
#pragma unroll
                for (int k = 0; k < NCOMP; ++k) u[k] = c.f[k][cbase + (NDIM == 1 ? 0 : m * slast)];
            }
            if (u[0] != u[0] || u[1] != u[1] || u[2] != u[2]) sawnan = 1;
            double sv;   // approximate first
            if constexpr (TKIND == LSM_TERM_ADVECTION) {
                sv = __builtin_fabs(u[0]) * ih0;
                if (NDIM > 1) sv = sv + __builtin_fabs(u[1]) * ih1;
                if (NDIM > 2) sv = sv + __builtin_fabs(u[2]) * ih2;
            } else if constexpr (TKIND == LSM_TERM_NORMAL_MOTION) {
                sv = __builtin_fabs(u[0]) * ih0;
                if (NDIM > 1) sv = sv + __builtin_fabs(u[0]) * ih1;
                if (NDIM > 2) sv = sv + __builtin_fabs(u[0]) * ih2;
            } else {
                sv = __builtin_fabs(u[0]);
            }
            if (pass >= 1 && sv >= thresh) {   // candidate: exact value with the reference's divisions
                if (pass == 2) {
                    const unsigned slot = atomicAdd(a.cand_count, 1u);
                    if (slot < a.cand_cap) a.cand[slot] = (NDIM == 3 ? (long long)i0 + (long long)a.n[0] * i1 : (long long)i0) + ncol * m;
                }
                if constexpr (TKIND == LSM_TERM_ADVECTION) {
                    sv = __builtin_fabs(u[0]) / a.h[0];
                    if (NDIM > 1) sv = sv + __builtin_fabs(u[1]) / a.h[1];
                    if (NDIM > 2) sv = sv + __builtin_fabs(u[2]) / a.h[2];
                } else if constexpr (TKIND == LSM_TERM_NORMAL_MOTION) {
                    sv = __builtin_fabs(u[0]) / a.h[0];
                    if (NDIM > 1) sv = sv + __builtin_fabs(u[0]) / a.h[1];
                    if (NDIM > 2) sv = sv + __builtin_fabs(u[0]) / a.h[2];
                }
            } else if (pass >= 1) {
                sv = 0.0;   // not a candidate
            }
            if (sv == sv) best = sv > best ? sv : best;
        }
    }
    best = wave_max(best);
    sawnan = __any(sawnan) ? 1 : 0;
    __shared__ double smax[4];
    __shared__ int snan[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { smax[wave] = best; snan[wave] = sawnan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = smax[0];
        int f = snan[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { m = smax[w] > m ? smax[w] : m; f |= snan[w]; }
        a.partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
        if (f) atomicOr(a.nanflag, 1);
    }
}

__global__ void __launch_bounds__(256) cfl_final_kernel(const double* partial, int nblocks, const int* nanflag, double* out,
                                                        int term_kind, double dxmin, int pass) {
    double best = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) best = partial[i] > best ? partial[i] : best;
    best = wave_max(best);
    __shared__ double smax[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) smax[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = smax[0];
        for (int w = 1; w < 4; ++w) m = smax[w] > m ? smax[w] : m;
        if (pass == 0) {
            out[1] = m * (1.0 - 1.0e-13);       // candidate threshold for pass 1
        } else {
            const double cfl = term_kind == LSM_TERM_CURVATURE ? (dxmin * dxmin) / (2 * m) : 1 / m;
            out[0] = *nanflag ? __builtin_nan("") : cfl;
        }
    }
}

// The candidates of a SEPARABLE coefficient u(x)·g(t) do not depend on t (the computed s is |g|·S(x)
// up to a few ulp), so they are recorded once (pass 2 above, with g = 1) and later steps evaluate the
// exact s at those nodes only: one small block instead of two sweeps of the grid.  Same arithmetic as
// the candidate branch of cfl_kernel.
template <int NDIM, int TKIND>
__global__ void __launch_bounds__(256) cfl_cand_kernel(const CflArgs a, unsigned count) {
    constexpr int NCOMP = TKIND == LSM_TERM_ADVECTION ? NDIM : 1;
    const CoeffArgs& c = a.coeff;
    const long long ncol = NDIM == 1 ? a.n[0] : (NDIM == 2 ? a.n[0] : (long long)a.n[0] * a.n[1]);
    const int toff = (NDIM == 3 ? a.gn[0] + a.gn[1] : a.gn[0]) + a.goff[NDIM - 1];
    double best = 0.0;
    int sawnan = 0;
    for (unsigned i = threadIdx.x; i < count; i += blockDim.x) {
        const long long e = a.cand[i];
        const int m = (int)(e / ncol);
        const long long col = e - (long long)m * ncol;
        const int i0 = NDIM == 3 ? (int)(col % a.n[0]) : (int)col, i1 = NDIM == 3 ? (int)(col / a.n[0]) : 0;
        const int g0 = i0 + a.goff[0], g1 = i1 + a.goff[1];
        double u[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) {
            double p = c.sep[k][g0];
            if (NDIM == 3) p = p * c.sep[k][a.gn[0] + g1];
            if (NDIM > 1) p = p * c.sep[k][toff + m];
            u[k] = p * c.tfac;
        }
        if (u[0] != u[0] || u[1] != u[1] || u[2] != u[2]) sawnan = 1;
        double sv;
        if constexpr (TKIND == LSM_TERM_ADVECTION) {
            sv = __builtin_fabs(u[0]) / a.h[0];
            if (NDIM > 1) sv = sv + __builtin_fabs(u[1]) / a.h[1];
            if (NDIM > 2) sv = sv + __builtin_fabs(u[2]) / a.h[2];
        } else {
            sv = __builtin_fabs(u[0]) / a.h[0];
            if (NDIM > 1) sv = sv + __builtin_fabs(u[0]) / a.h[1];
            if (NDIM > 2) sv = sv + __builtin_fabs(u[0]) / a.h[2];
        }
        if (sv == sv) best = sv > best ? sv : best;
    }
    best = wave_max(best);
    sawnan = __any(sawnan) ? 1 : 0;
    __shared__ double smax[4];
    __shared__ int snan[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { smax[wave] = best; snan[wave] = sawnan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = smax[0];
        int f = snan[0];
        for (int w = 1; w < 4; ++w) { m = smax[w] > m ? smax[w] : m; f |= snan[w]; }
        a.partial[0] = m;
        if (f) atomicOr(a.nanflag, 1);
    }
}
void launch_cfl_candidates(int ndim, const CflArgs& a, unsigned count, hipStream_t s) {
    const bool adv = a.term_kind == LSM_TERM_ADVECTION;
    if (ndim == 1) { if (adv) hipLaunchKernelGGL((cfl_cand_kernel<1, LSM_TERM_ADVECTION>), dim3(1), dim3(256), 0, s, a, count); else hipLaunchKernelGGL((cfl_cand_kernel<1, LSM_TERM_NORMAL_MOTION>), dim3(1), dim3(256), 0, s, a, count); }
    else if (ndim == 2) { if (adv) hipLaunchKernelGGL((cfl_cand_kernel<2, LSM_TERM_ADVECTION>), dim3(1), dim3(256), 0, s, a, count); else hipLaunchKernelGGL((cfl_cand_kernel<2, LSM_TERM_NORMAL_MOTION>), dim3(1), dim3(256), 0, s, a, count); }
    else { if (adv) hipLaunchKernelGGL((cfl_cand_kernel<3, LSM_TERM_ADVECTION>), dim3(1), dim3(256), 0, s, a, count); else hipLaunchKernelGGL((cfl_cand_kernel<3, LSM_TERM_NORMAL_MOTION>), dim3(1), dim3(256), 0, s, a, count); }
}

// Narrow band in 3-D with the compact list of active tiles at hand (lsm_band_status): one workgroup per listed 32×8×tm
// tile instead of a march over every column of the grid (at 768³ the column kernel spends 0.15 ms stepping over 221 k tile
// flags to visit 3.7 M band nodes).  Exact divisions throughout: the band is small.  Same node formulas and the same
// max-then-divide reduction as cfl_kernel, so Δt is bit-identical.
template <int TKIND, int CKIND>
__global__ void __launch_bounds__(256) cfl_band_list_kernel(const CflArgs a, const int* list, unsigned nlist_host, const unsigned* nlist_dev) {
    const unsigned nlist = nlist_dev ? *nlist_dev : nlist_host;      // the list's length may still be on the device only (prefetch after update_band!)
    constexpr int NCOMP = TKIND == LSM_TERM_ADVECTION ? 3 : 1;
    const CoeffArgs& c = a.coeff;
    double best = 0.0;
    int sawnan = 0;
    const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;     // the launcher checks tx·ty == 256
    for (unsigned e = blockIdx.x; e < nlist; e += gridDim.x) {
        const unsigned tile = (unsigned)list[e];
        const int i0 = (int)(tile % a.nbx) * a.tx + lx, i1 = (int)((tile / a.nbx) % a.nby) * a.ty + ly;
        const int mlo = (int)(tile / (a.nbx * a.nby)) * a.tm;
        if (i0 >= a.n[0] || i1 >= a.n[1]) continue;
        const int g0 = i0 + a.goff[0], g1 = i1 + a.goff[1];
        const long long cbase = a.origin + i0 + i1 * a.s1;
        if constexpr (CKIND == LSM_COEFF_ROTATION && TKIND == LSM_TERM_ADVECTION) {
            // the in-plane rotation's speed does not depend on the march index: one evaluation per column that holds a band node
            // (same operations on the same operands as the per-node form below: the maximum is bit-identical)
            unsigned anyb = 0;
            if (a.rowbits) {
                const unsigned* wr = a.rowbits + (size_t)tile * (size_t)(a.ty * a.tm) + ly;
                for (int i = 0; i < a.tm && mlo + i < a.n[2]; ++i) anyb |= wr[a.ty * i];
                anyb = (anyb >> lx) & 1u;
            } else {
                for (int m = mlo; m < mlo + a.tm && m < a.n[2]; ++m) anyb |= a.mask[cbase + m * a.s2];
            }
            if (!anyb) continue;
            const double u0 = -(c.v[0] * ((a.lc[1] + (double)g1 * a.h[1]) - c.v[2]));
            const double u1 = c.v[0] * ((a.lc[0] + (double)g0 * a.h[0]) - c.v[1]);
            if (u0 != u0 || u1 != u1) sawnan = 1;
            const double sv = (__builtin_fabs(u0) / a.h[0] + __builtin_fabs(u1) / a.h[1]) + __builtin_fabs(0.0) / a.h[2];
            if (sv == sv) best = sv > best ? sv : best;
            continue;
        }
        for (int m = mlo; m < mlo + a.tm && m < a.n[2]; ++m) {
            const long long q = cbase + m * a.s2;
            if (a.rowbits ? !((a.rowbits[(size_t)tile * (size_t)(a.ty * a.tm) + ly + a.ty * (m - mlo)] >> lx) & 1u) : !a.mask[q]) continue;
            double u[3] = {0, 0, 0};
            if constexpr (CKIND == LSM_COEFF_CONST) {
#pragma unroll
                for (int k = 0; k < NCOMP; ++k) u[k] = c.v[k];
            } else if constexpr (CKIND == LSM_COEFF_SEPARABLE) {
                const int gm = a.gn[0] + a.gn[1] + a.goff[2] + m;
#pragma unroll
                for (int k = 0; k < NCOMP; ++k) u[k] = ((c.sep[k][g0] * c.sep[k][a.gn[0] + g1]) * c.sep[k][gm]) * c.tfac;
            } else if constexpr (CKIND == LSM_COEFF_ROTATION) {
                u[0] = -(c.v[0] * ((a.lc[1] + (double)g1 * a.h[1]) - c.v[2]));
                u[1] = c.v[0] * ((a.lc[0] + (double)g0 * a.h[0]) - c.v[1]);
            } else {
#pragma unroll
                for (int k = 0; k < NCOMP; ++k) u[k] = c.f[k][q];
            }
            if (u[0] != u[0] || u[1] != u[1] || u[2] != u[2]) sawnan = 1;
            double sv;
            if constexpr (TKIND == LSM_TERM_ADVECTION) sv = (__builtin_fabs(u[0]) / a.h[0] + __builtin_fabs(u[1]) / a.h[1]) + __builtin_fabs(u[2]) / a.h[2];
            else if constexpr (TKIND == LSM_TERM_NORMAL_MOTION) sv = (__builtin_fabs(u[0]) / a.h[0] + __builtin_fabs(u[0]) / a.h[1]) + __builtin_fabs(u[0]) / a.h[2];
            else sv = __builtin_fabs(u[0]);
            if (sv == sv) best = sv > best ? sv : best;
        }
    }
    best = wave_max(best);
    sawnan = __any(sawnan) ? 1 : 0;
    __shared__ double smax[4];
    __shared__ int snan[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { smax[wave] = best; snan[wave] = sawnan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = smax[0];
        int f = snan[0];
        for (int w = 1; w < 4; ++w) { m = smax[w] > m ? smax[w] : m; f |= snan[w]; }
        a.partial[blockIdx.x] = m;
        if (f) atomicOr(a.nanflag, 1);
    }
}
template <int TKIND>
static void launch_cfl_band_list_ck(const CflArgs& a, const int* list, unsigned nlist, const unsigned* nlist_dev, unsigned grid, hipStream_t s) {
    switch (a.coeff.kind) {
    case LSM_COEFF_CONST: hipLaunchKernelGGL((cfl_band_list_kernel<TKIND, LSM_COEFF_CONST>), dim3(grid), dim3(256), 0, s, a, list, nlist, nlist_dev); break;
    case LSM_COEFF_ROTATION: hipLaunchKernelGGL((cfl_band_list_kernel<TKIND, LSM_COEFF_ROTATION>), dim3(grid), dim3(256), 0, s, a, list, nlist, nlist_dev); break;
    case LSM_COEFF_SEPARABLE: hipLaunchKernelGGL((cfl_band_list_kernel<TKIND, LSM_COEFF_SEPARABLE>), dim3(grid), dim3(256), 0, s, a, list, nlist, nlist_dev); break;
    default: hipLaunchKernelGGL((cfl_band_list_kernel<TKIND, LSM_COEFF_FIELD>), dim3(grid), dim3(256), 0, s, a, list, nlist, nlist_dev);
    }
}
// returns the number of partials written (0: not applicable — use launch_cfl)
// nlist_dev != NULL: the length is read on the device and max_partials workgroups are launched (those without a tile write the
// neutral partial 0)
int launch_cfl_band_list(const CflArgs& a, const int* list, unsigned nlist, const unsigned* nlist_dev, int max_partials, hipStream_t s) {
    if (!list || (!nlist_dev && nlist == 0) || !a.mask || a.tx * a.ty != 256 || a.tm < 1) return 0;
    const unsigned grid = (nlist_dev || nlist >= (unsigned)max_partials) ? (unsigned)max_partials : nlist;
    if (a.term_kind == LSM_TERM_ADVECTION) launch_cfl_band_list_ck<LSM_TERM_ADVECTION>(a, list, nlist, nlist_dev, grid, s);
    else if (a.term_kind == LSM_TERM_NORMAL_MOTION) launch_cfl_band_list_ck<LSM_TERM_NORMAL_MOTION>(a, list, nlist, nlist_dev, grid, s);
    else launch_cfl_band_list_ck<LSM_TERM_CURVATURE>(a, list, nlist, nlist_dev, grid, s);
    return (int)grid;
}

static int cfl_chunks(int ndim, const int n[3]) {
    if (ndim == 1) return 1;
    int c = 4;
    while ((n[ndim - 1] + c - 1) / c > CFL_MAX_CHUNK) c *= 2;
    return c;
}
int cfl_blocks(int ndim, const int n[3]) {
    const long long cols = ndim == 1 ? n[0] : (ndim == 2 ? n[0] : (long long)n[0] * n[1]);
    const long long b = (cols + 255) / 256;
    const int bx = (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
    return bx * cfl_chunks(ndim, n);   // number of partials
}
template <int NDIM, int TKIND>
static void launch_cfl_ck(const CflArgs& a, dim3 grid, int pass, const double* thresh, hipStream_t s) {
    switch (a.coeff.kind) {
    case LSM_COEFF_CONST: hipLaunchKernelGGL((cfl_kernel<NDIM, TKIND, LSM_COEFF_CONST>), grid, dim3(256), 0, s, a, pass, thresh); break;
    case LSM_COEFF_ROTATION: hipLaunchKernelGGL((cfl_kernel<NDIM, TKIND, LSM_COEFF_ROTATION>), grid, dim3(256), 0, s, a, pass, thresh); break;
    case LSM_COEFF_SEPARABLE: hipLaunchKernelGGL((cfl_kernel<NDIM, TKIND, LSM_COEFF_SEPARABLE>), grid, dim3(256), 0, s, a, pass, thresh); break;
    default: hipLaunchKernelGGL((cfl_kernel<NDIM, TKIND, LSM_COEFF_FIELD>), grid, dim3(256), 0, s, a, pass, thresh);
    }
}
template <int NDIM>
static void launch_cfl_nd(const CflArgs& a, dim3 grid, int pass, const double* thresh, hipStream_t s) {
    if (a.term_kind == LSM_TERM_ADVECTION) launch_cfl_ck<NDIM, LSM_TERM_ADVECTION>(a, grid, pass, thresh, s);
    else if (a.term_kind == LSM_TERM_NORMAL_MOTION) launch_cfl_ck<NDIM, LSM_TERM_NORMAL_MOTION>(a, grid, pass, thresh, s);
    else launch_cfl_ck<NDIM, LSM_TERM_CURVATURE>(a, grid, pass, thresh, s);
}
void launch_cfl(int ndim, const CflArgs& a, int nblocks, int pass, const double* thresh, hipStream_t s) {
    const int chunks = cfl_chunks(ndim, a.n);
    const dim3 grid(nblocks / chunks, chunks);
    if (ndim == 1) launch_cfl_nd<1>(a, grid, pass, thresh, s);
    else if (ndim == 2) launch_cfl_nd<2>(a, grid, pass, thresh, s);
    else launch_cfl_nd<3>(a, grid, pass, thresh, s);
}
void launch_cfl_final(const double* partial, int nblocks, const int* nanflag, double* out, int term_kind, double dxmin,
                      int pass, hipStream_t s) {
    hipLaunchKernelGGL(cfl_final_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, nanflag, out, term_kind, dxmin, pass);
}

// ---------------------------------------------------------------------------------------------
// extrema of the interior (show, src/meshfield.jl:300-303)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) extrema_kernel(int n0, int n1, int n2, long long s1, long long s2, long long origin,
                                                      const void* v, int f32, double* pmin, double* pmax) {
    const long long total = (long long)n0 * n1 * n2;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % n0), i1 = (int)((t / n0) % n1), i2 = (int)(t / ((long long)n0 * n1));
        const double x = ld_val(v, origin + i0 + i1 * s1 + i2 * s2, f32);
        lo = x < lo ? x : lo;
        hi = x > hi ? x : hi;
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    __shared__ double sl[4], sh[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sl[wave] = lo; sh[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = sl[w] < lo ? sl[w] : lo; hi = sh[w] > hi ? sh[w] : hi; }
        lo = sl[0] < lo ? sl[0] : lo;
        hi = sh[0] > hi ? sh[0] : hi;
        pmin[blockIdx.x] = lo;
        pmax[blockIdx.x] = hi;
    }
}
__global__ void __launch_bounds__(256) extrema_final_kernel(const double* pmin, const double* pmax, int nblocks, double* out2) {
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        lo = pmin[i] < lo ? pmin[i] : lo;
        hi = pmax[i] > hi ? pmax[i] : hi;
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    __shared__ double sl[4], sh[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sl[wave] = lo; sh[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 4; ++w) { lo = sl[w] < lo ? sl[w] : lo; hi = sh[w] > hi ? sh[w] : hi; }
        out2[0] = lo;
        out2[1] = hi;
    }
}
void launch_extrema(int /*ndim*/, const int n[3], long long s1, long long s2, long long origin, const void* v, int f32,
                    double* partial_min, double* partial_max, int nblocks, double* out2, hipStream_t s) {
    hipLaunchKernelGGL(extrema_kernel, dim3(nblocks), dim3(256), 0, s, n[0], n[1], n[2], s1, s2, origin, v, f32, partial_min,
                       partial_max);
    hipLaunchKernelGGL(extrema_final_kernel, dim3(1), dim3(256), 0, s, partial_min, partial_max, nblocks, out2);
}

// ---------------------------------------------------------------------------------------------
// volume / perimeter (src/levelsetops.jl:27-33,139-149,171-183): grid sums of the smoothed
// Heaviside H(-ϕ) resp. of δ(ϕ)·‖∇ϕ‖ (centred differences, ghosts through the field's BCs),
// times prod(h).  One partial sum per workgroup (wave shuffles + LDS, fixed order), summed by a
// second tiny kernel: deterministic, and equal to the reference's pairwise sum up to rounding.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double smooth_heaviside(double x, double alpha) {   // src/levelsetops.jl:171-179
    if (x > alpha) return 1.0;
    if (x < -alpha) return 0.0;
    return 0.5 * (1.0 + x / alpha + 1.0 / M_PI * sin(M_PI * x / alpha));
}
__device__ __forceinline__ double smooth_delta(double x, double alpha) {       // src/levelsetops.jl:181-183
    return __builtin_fabs(x) > alpha ? 0.0 : 0.5 / alpha * (1.0 + cos(M_PI * x / alpha));
}
// mode 0: volume, 1: perimeter
__global__ void __launch_bounds__(256) measure_kernel(int mode, int ndim, int n0, int n1, int n2, long long s1, long long s2,
                                                      long long origin, double h0, double h1, double h2, double dmin,
                                                      const void* vp, int f32, double* partial, const unsigned char* mask) {
    auto v = [&](long long i) { return ld_val(vp, i, f32); };
    const long long total = (long long)n0 * n1 * n2;
    double acc = 0.0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % n0), i1 = (int)((t / n0) % n1), i2 = (int)(t / ((long long)n0 * n1));
        const long long q = origin + i0 + i1 * s1 + i2 * s2;
        if (mask && !mask[q]) continue;      // NarrowBandMeshField: the sum runs over the active nodes (src/levelsetops.jl:157-166)
        const double c = v(q);
        if (mode == 0) {
            acc += smooth_heaviside(-c, dmin);
        } else {
            const double d = smooth_delta(c, dmin);
            if (d != 0.0) {   // ‖∇ϕ‖ from D⁰ (src/levelsetops.jl:212-215), only inside the delta's support
                double g0 = (v(q + 1) - v(q - 1)) / (2 * h0);
                double nrm2 = g0 * g0;
                if (ndim > 1) { double g1 = (v(q + s1) - v(q - s1)) / (2 * h1); nrm2 = nrm2 + g1 * g1; }
                if (ndim > 2) { double g2 = (v(q + s2) - v(q - s2)) / (2 * h2); nrm2 = nrm2 + g2 * g2; }
                acc += d * __builtin_sqrt(nrm2);
            }
        }
    }
    acc = wave_sum(acc);
    __shared__ double ssum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) ssum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
}
__global__ void __launch_bounds__(256) measure_final_kernel(const double* partial, int nblocks, double scale, double* out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) acc += partial[i];
    acc = wave_sum(acc);
    __shared__ double ssum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) ssum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = scale * ((ssum[0] + ssum[1]) + (ssum[2] + ssum[3]));
}
void launch_measure(int mode, int ndim, const int n[3], long long s1, long long s2, long long origin, const double h[3],
                    double dmin, double scale, const void* v, int f32, double* partial, int nblocks, double* out, hipStream_t s,
                    const unsigned char* mask) {
    hipLaunchKernelGGL(measure_kernel, dim3(nblocks), dim3(256), 0, s, mode, ndim, n[0], n[1], n[2], s1, s2, origin, h[0], h[1], h[2],
                       dmin, v, f32, partial, mask);
    hipLaunchKernelGGL(measure_final_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, scale, out);
}

// ---------------------------------------------------------------------------------------------
// volume(nb::NarrowBandMeshField) (src/levelsetops.jl:34-116): Σ_band H(-ϕ) + the number of off-band nodes inside.
// (1) one wave per grid line along dimension 1: 64 mask bytes per step, band / negative-band bit masks by ballot; the
//     off-band interior of the line = the tail before the first band node if that node is negative, the tail after the
//     last one likewise, and every gap whose two end nodes are both negative (_count_offband_interior, :67-91); also
//     the line's Σ H(-ϕ), and the band node nearest to x0 = n1÷2 (the reference's query point) with its sign.
// (2) a line without a band node carries no crossing and takes the sign of the nearest band node of the whole band
//     (_count_bandfree_interior, :96-113, a KD-tree there).  Its squared index distance to a band line t' is
//     |t - t'|² + (nearest band node of t' to x0)², a separable min-plus problem: one pass along the second dimension,
//     one along the third, exact in integers.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) band_lines_kernel(int ndim, int n0, int n1, int n2, long long s1, long long s2, long long origin, double dmin,
                                                         const void* vp, int f32, const unsigned char* mask, int x0, long long* line_count,
                                                         double* line_hsum, int* line_near, int* line_neg) {
    const int lane = threadIdx.x & 63;
    const long long nlines = (long long)n1 * n2;
    for (long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < nlines; t += ((long long)gridDim.x * blockDim.x) >> 6) {
        const int i1 = (int)(t % n1), i2 = (int)(t / n1);
        const long long row = origin + i1 * s1 + i2 * s2;
        long long count = 0;
        double hs = 0.0;
        int prev = -1, prev_neg = 0;          // last band node seen so far on the line and whether its value is negative
        int best = 0x7fffffff, best_neg = 0;  // distance of the band node nearest to x0, and its sign
        for (int b = 0; b < n0; b += 64) {
            const int i0 = b + lane;
            const bool on = i0 < n0 && mask[row + i0] != 0;
            const double v = on ? ld_val(vp, row + i0, f32) : 0.0;
            if (on) hs += smooth_heaviside(-v, dmin);
            unsigned long long bm = __ballot(on), ng = __ballot(on && v < 0.0);
            while (bm) {                       // wave-uniform walk over the band nodes of this chunk
                const int k = __ffsll((long long)bm) - 1;
                bm &= bm - 1;
                const int x = b + k, neg = (int)((ng >> k) & 1ull);
                if (prev < 0) { if (neg) count += x; }                       // tail before the first band node (x nodes: 0..x-1)
                else if (x - prev - 1 > 0 && neg && prev_neg) count += x - prev - 1;
                const int d = x > x0 ? x - x0 : x0 - x;
                if (d < best) { best = d; best_neg = neg; }
                prev = x; prev_neg = neg;
            }
        }
        if (prev >= 0 && prev_neg) count += n0 - 1 - prev;                   // tail after the last band node
        hs = wave_sum(hs);
        if (lane == 0) {
            line_count[t] = count;
            line_hsum[t] = hs;
            line_near[t] = prev >= 0 ? best : -1;
            line_neg[t] = best_neg;
        }
    }
}
// pass along dimension `dim` (1 or 2) of the line lattice: g'(t) = min_{t'} (Δ_dim² + g(t')) with the sign carried along
__global__ void __launch_bounds__(256) band_lines_minplus_kernel(int dim, int n1, int n2, const long long* gin, const int* negin, long long* gout,
                                                                 int* negout) {
    const long long nlines = (long long)n1 * n2;
    const long long INF = 0x3fffffffffffffffll;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nlines; t += (long long)gridDim.x * blockDim.x) {
        const int i1 = (int)(t % n1), i2 = (int)(t / n1);
        const int len = dim == 1 ? n1 : n2, me = dim == 1 ? i1 : i2;
        const long long base = dim == 1 ? (long long)i2 * n1 : i1, step = dim == 1 ? 1 : n1;
        long long best = INF;
        int bneg = 0;
        for (int k = 0; k < len; ++k) {
            const long long g = gin[base + k * step];
            if (g >= INF) continue;
            const long long d = (long long)(k - me) * (k - me) + g;
            if (d < best) { best = d; bneg = negin[base + k * step]; }
        }
        gout[t] = best;
        negout[t] = bneg;
    }
}
__global__ void __launch_bounds__(256) band_lines_init_kernel(long long nlines, const int* line_near, long long* g) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nlines; t += (long long)gridDim.x * blockDim.x)
        g[t] = line_near[t] < 0 ? 0x3fffffffffffffffll : (long long)line_near[t] * line_near[t];
}
// Σ over the lines: H sums + counted interior nodes + n0 for every band-free line whose nearest band node is negative
__global__ void __launch_bounds__(256) band_lines_final_kernel(long long nlines, int n0, const long long* line_count, const double* line_hsum,
                                                               const int* line_near, const long long* g, const int* gneg, double scale, double* out) {
    double acc = 0.0;
    bool any = false;
    for (long long t = threadIdx.x; t < nlines; t += blockDim.x) {
        if (line_near[t] >= 0) { acc += line_hsum[t] + (double)line_count[t]; any = true; }
        else if (g[t] < 0x3fffffffffffffffll && gneg[t]) acc += (double)n0;
    }
    acc = wave_sum(acc);
    __shared__ double ssum[4];
    __shared__ int sany;
    if (threadIdx.x == 0) sany = 0;
    __syncthreads();
    if (any) sany = 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) ssum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sany ? scale * ((ssum[0] + ssum[1]) + (ssum[2] + ssum[3])) : 0.0;   // empty band: 0 (:54)
}
int launch_band_volume(int ndim, const int n[3], long long s1, long long s2, long long origin, double dmin, double scale, const void* v, int f32,
                       const unsigned char* mask, double* out, hipStream_t s) {
    const int n0 = n[0], n1 = ndim > 1 ? n[1] : 1, n2 = ndim > 2 ? n[2] : 1;
    const long long nlines = (long long)n1 * n2;
    long long *cnt = nullptr, *g0 = nullptr, *g1 = nullptr;
    double* hsum = nullptr;
    int *near = nullptr, *neg0 = nullptr, *neg1 = nullptr;
    auto cleanup = [&]() { (void)hipFree(cnt); (void)hipFree(g0); (void)hipFree(g1); (void)hipFree(hsum); (void)hipFree(near); (void)hipFree(neg0); (void)hipFree(neg1); };
    if (hipMalloc((void**)&cnt, 8 * nlines) != hipSuccess || hipMalloc((void**)&g0, 8 * nlines) != hipSuccess ||
        hipMalloc((void**)&g1, 8 * nlines) != hipSuccess || hipMalloc((void**)&hsum, 8 * nlines) != hipSuccess ||
        hipMalloc((void**)&near, 4 * nlines) != hipSuccess || hipMalloc((void**)&neg0, 4 * nlines) != hipSuccess ||
        hipMalloc((void**)&neg1, 4 * nlines) != hipSuccess) { cleanup(); return 1; }
    const int x0 = n0 / 2 - 1;                                   // the reference's 1-based n1 ÷ 2
    long long wb = (nlines * 64 + 255) / 256;
    hipLaunchKernelGGL(band_lines_kernel, dim3((unsigned)(wb > 65535 ? 65535 : wb)), dim3(256), 0, s, ndim, n0, n1, n2, s1, s2, origin, dmin, v, f32,
                       mask, x0, cnt, hsum, near, neg0);
    const unsigned lb = (unsigned)((nlines + 255) / 256 > 4096 ? 4096 : (nlines + 255) / 256);
    hipLaunchKernelGGL(band_lines_init_kernel, dim3(lb), dim3(256), 0, s, nlines, near, g0);
    hipLaunchKernelGGL(band_lines_minplus_kernel, dim3(lb), dim3(256), 0, s, 1, n1, n2, g0, neg0, g1, neg1);
    hipLaunchKernelGGL(band_lines_minplus_kernel, dim3(lb), dim3(256), 0, s, 2, n1, n2, g1, neg1, g0, neg0);
    hipLaunchKernelGGL(band_lines_final_kernel, dim3(1), dim3(256), 0, s, nlines, n0, cnt, hsum, near, g0, neg0, scale, out);
    const bool ok = hipStreamSynchronize(s) == hipSuccess;
    cleanup();
    return ok ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// _signed_normal_components (src/velocityextension.jl:96-116) with the frozen mask
// (_normalize_frozen_mask, :78-94) folded in: a_d = S·∂_dϕ/|∇ϕ| (centred differences, S = ϕ/√(ϕ²+Δ²)),
// 0 where |∇ϕ|² <= min_norm² — and 0 on frozen nodes, which makes the upwind update of
// extend_along_normals! leave them untouched exactly (F - τ·Σ 0·dF = F).
// frozen: optional padded array (non-zero = frozen); NULL -> band rule |ϕ| <= band_width.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) signed_normals_kernel(int ndim, int n0, int n1, int n2, long long s1, long long s2,
                                                             long long origin, double h0, double h1, double h2, double delta,
                                                             double band_width, double min_norm2, const void* phip, int f32,
                                                             const double* frozen, double* c0, double* c1, double* c2) {
    auto phi = [&](long long i) { return ld_val(phip, i, f32); };
    const long long total = (long long)n0 * n1 * n2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % n0), i1 = (int)((t / n0) % n1), i2 = (int)(t / ((long long)n0 * n1));
        const long long q = origin + i0 + i1 * s1 + i2 * s2;
        const double c = phi(q);
        double g[3] = {0, 0, 0};
        g[0] = (phi(q + 1) - phi(q - 1)) / (2 * h0);
        double nrm2 = g[0] * g[0];
        if (ndim > 1) { g[1] = (phi(q + s1) - phi(q - s1)) / (2 * h1); nrm2 = nrm2 + g[1] * g[1]; }
        if (ndim > 2) { g[2] = (phi(q + s2) - phi(q - s2)) / (2 * h2); nrm2 = nrm2 + g[2] * g[2]; }
        const bool fz = frozen ? frozen[q] != 0.0 : __builtin_fabs(c) <= band_width;
        double a[3] = {0, 0, 0};
        if (!fz && !(nrm2 <= min_norm2)) {
            const double invnorm = 1.0 / __builtin_sqrt(nrm2);
            const double S = c / __builtin_sqrt(c * c + delta * delta);
            a[0] = S * g[0] * invnorm;
            a[1] = S * g[1] * invnorm;
            a[2] = S * g[2] * invnorm;
        }
        c0[q] = a[0];
        if (ndim > 1) c1[q] = a[1];
        if (ndim > 2) c2[q] = a[2];
    }
}
void launch_signed_normals(int ndim, const int n[3], long long s1, long long s2, long long origin, const double h[3], double delta,
                           double band_width, double min_norm2, const void* phi, int f32, const double* frozen, double* c0, double* c1,
                           double* c2, hipStream_t s) {
    const long long total = (long long)n[0] * n[1] * n[2];
    long long b = (total + 255) / 256;
    const int nb = (int)(b > 4096 ? 4096 : b);
    hipLaunchKernelGGL(signed_normals_kernel, dim3(nb), dim3(256), 0, s, ndim, n[0], n[1], n[2], s1, s2, origin, h[0], h[1], h[2],
                       delta, band_width, min_norm2, phi, f32, frozen, c0, c1, c2);
}

// ---------------------------------------------------------------------------------------------
// curvature / gradient / normal at every node (src/levelsetops.jl:197-226), the reference's operation order
// (this file is built with -ffp-contract=off): D⁰, D2⁰, D2 of src/derivatives.jl:129-149; gᵀHg in the order of
// LinearAlgebra.dot(x, ::Symmetric, y) as in the CurvatureTerm of the stage kernel.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) geometry_kernel(int what, int ndim, int n0, int n1, int n2, long long s1, long long s2, long long origin,
                                                       double h0, double h1, double h2, double scale, double band_width, double fill,
                                                       const void* phip, int f32, double* o0, double* o1, double* o2, double* frozen,
                                                       const unsigned char* mask) {
    auto phi = [&](long long i) { return ld_val(phip, i, f32); };
    const long long total = (long long)n0 * n1 * n2;
    const double hh[3] = {h0, h1, h2};
    const long long st[3] = {1, s1, s2};
    double* out[3] = {o0, o1, o2};
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % n0), i1 = (int)((t / n0) % n1), i2 = (int)(t / ((long long)n0 * n1));
        const long long q = origin + i0 + i1 * s1 + i2 * s2;
        const double c = phi(q);
        const bool on = (band_width < 0.0 || __builtin_fabs(c) <= band_width) && (!mask || mask[q]);   // band fields: active nodes only
        if (frozen) frozen[q] = on ? 1.0 : 0.0;
        if (!on) {
            if (what == LSM_GEOM_CURVATURE) o0[q] = fill;
            else for (int d = 0; d < ndim; ++d) out[d][q] = fill;
            continue;
        }
        double g[3] = {0, 0, 0};
        for (int d = 0; d < ndim; ++d) g[d] = (phi(q + st[d]) - phi(q - st[d])) / (2 * hh[d]);
        double nrmsq = g[0] * g[0];
        for (int d = 1; d < ndim; ++d) nrmsq = nrmsq + g[d] * g[d];
        if (what != LSM_GEOM_CURVATURE) {
            const double nrm = __builtin_sqrt(nrmsq);
            for (int d = 0; d < ndim; ++d) out[d][q] = scale * (what == LSM_GEOM_GRADIENT ? g[d] : g[d] / nrm);
            continue;
        }
        double kappa = 0.0;
        if (!(nrmsq < 2.220446049250313e-16)) {
            double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int a = 0; a < ndim; ++a) {
                H[a][a] = (phi(q + st[a]) - 2 * c + phi(q - st[a])) / (hh[a] * hh[a]);
                for (int b = a + 1; b < ndim; ++b)
                    H[a][b] = ((phi(q + st[a] + st[b]) - phi(q + st[a] - st[b])) / (2 * hh[b]) -
                               (phi(q - st[a] + st[b]) - phi(q - st[a] - st[b])) / (2 * hh[b])) / (2 * hh[a]);
            }
            double lap = H[0][0];
            for (int d = 1; d < ndim; ++d) lap = lap + H[d][d];
            double r = 0.0;
            for (int j = 0; j < ndim; ++j) {
                r += g[j] * (H[j][j] * g[j]);
                for (int i = 0; i < j; ++i) r += g[i] * (H[i][j] * g[j]) + g[j] * (H[i][j] * g[i]);
            }
            kappa = (lap * nrmsq - r) / pow(nrmsq, 1.5);
        }
        o0[q] = scale * kappa;
    }
}
void launch_geometry(int what, int ndim, const int n[3], long long s1, long long s2, long long origin, const double h[3], double scale,
                     double band_width, double fill, const void* phi, int f32, double* o0, double* o1, double* o2, double* frozen,
                     hipStream_t s, const unsigned char* mask) {
    const long long total = (long long)n[0] * n[1] * n[2];
    long long b = (total + 255) / 256;
    const int nb = (int)(b > 8192 ? 8192 : b);
    hipLaunchKernelGGL(geometry_kernel, dim3(nb), dim3(256), 0, s, what, ndim, n[0], n[1], n[2], s1, s2, origin, h[0], h[1], h[2], scale,
                       band_width, fill, phi, f32, o0, o1, o2, frozen, mask);
}

// ---------------------------------------------------------------------------------------------
// EikonalReinitializationTerm(ϕ₀): S₀ = v / sqrt(v² + Δx²) on the interior (src/levelsetterms.jl:217-221)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) eikonal_sign_kernel(int n0, int n1, int n2, long long s1, long long s2,
                                                           long long origin, double dx, const void* phi0, int f32, double* s0) {
    const long long total = (long long)n0 * n1 * n2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i0 = (int)(t % n0), i1 = (int)((t / n0) % n1), i2 = (int)(t / ((long long)n0 * n1));
        const long long q = origin + i0 + i1 * s1 + i2 * s2;
        const double v = ld_val(phi0, q, f32);
        s0[q] = v / __builtin_sqrt(v * v + dx * dx);
    }
}
void launch_eikonal_sign(int /*ndim*/, const int n[3], long long s1, long long s2, long long origin, double dxmin,
                         const void* phi0, int f32, double* s0, hipStream_t s) {
    const long long total = (long long)n[0] * n[1] * n[2];
    long long b = (total + 255) / 256;
    const int nb = (int)(b > 4096 ? 4096 : b);
    hipLaunchKernelGGL(eikonal_sign_kernel, dim3(nb), dim3(256), 0, s, n[0], n[1], n[2], s1, s2, origin, dxmin, phi0, f32, s0);
}

}  // namespace lsm
