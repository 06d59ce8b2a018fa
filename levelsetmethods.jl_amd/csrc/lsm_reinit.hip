// lsm_reinit.hip — reinitialize!(ϕ; order, upsample, maxiters, xtol, ftol) on the device: the Newton
// closest-point signed distance of the reference (src/reinitializer.jl:12-42, src/sdf.jl,
// src/interpolation.jl, src/bernstein.jl).
//
//  * the continuous field is the reference's piecewise interpolant: on the cell with lower corner I, the
//    tensor-product polynomial whose Bernstein coefficients are kron(M,…,M)·(stencil values), M the
//    pseudo-inverse of the Bernstein collocation matrix (src/interpolation.jl:52-63,100-151).  It is evaluated
//    here in the equivalent cardinal form p(x) = Σ_j v_j Π_d L_{j_d}(t_d), L_j(t) = Σ_i M_ij B_i(t): no
//    per-cell coefficient storage, value / gradient / Hessian are the exact derivatives of the same polynomial
//    (the reference differentiates it with ForwardDiff).
//  * sampling (src/sdf.jl:186-221): uniformly spaced start points of every candidate cell are projected onto
//    the zero set by Newton steps that follow the iterate across cells; converged points that land back in
//    their own cell are kept.  The reference skips cells it can prove empty from the Bernstein coefficients;
//    skipping is only an optimisation (a start point of an empty cell never converges back into it).  Here a
//    cheap conservative bound (|p - c| <= Λ·spread of the stencil values) runs over all cells and the
//    reference's Bernstein test over the survivors.
//  * closest point (src/sdf.jl:85-131,223-249): the nearest sample — exact, like the reference's KD-tree query:
//    a cheap estimate of the closest point (one lane per node) leads to the cells that hold it, and every cell meeting the
//    ball of that radius around the node is then scanned (16 lanes per node; occupancy bits per cell); nodes without a
//    usable estimate fall back to expanding shells of cells, nodes far from the interface to the occupied 8^N-cell blocks of the
//    occupied super-blocks (8^N blocks) that meet the ball of the best distance so far — seeds a damped
//    Newton–Lagrange solve on the seed cell's patch.  Where that solve fails, the NSEED nearest samples are collected by the
//    shell search and tried in order in a second pass (the reference: nn first, then knn with up to 10).
// fp64 throughout; -ffp-contract=off except inside the patch evaluation (cardinal functions and the tensor contraction), which
// contracts to fma.
#include <algorithm>
#include <cmath>
#include <vector>

#include "lsm_internal.h"

namespace lsm {

struct ReinitArgs {
    int ndim;
    int n[3];              // nodes of the local slab
    int goff[3];           // global index of local node 0 (coordinates are lc + (I + goff)·h, as on a single device)
    long long s1, s2, origin;
    double lc[3], h[3];
    int order, nv, off;    // polynomial degree, stencil size, stencil offset of the cell's lower corner
    double M[36];          // [order+1][nv]
    double lambda;         // Lebesgue-type constant of the patch: |p - c| <= lambda * max_j |v_j - c|
    int upsample, maxiters;
    double xtol, ftol;
    const void* phi;       // padded field, ghosts / band halo filled
    int f32;
    const unsigned char* mask;   // NULL = dense
};

// A length that may still be on the device when the kernel is launched: the host sizes grids and buffers from what the previous call
// needed, the kernel reads the count and never goes beyond the buffer's capacity; whether it fitted is checked once, at the end of
// the call (reinit_run).  p == NULL: the host knows the length.
struct DevCount {
    const unsigned* p;
    unsigned cap;
    long long host;
};
__device__ __forceinline__ long long count_of(const DevCount& c) {
    if (!c.p) return c.host;
    const unsigned v = *c.p;
    return v < c.cap ? v : c.cap;
}

__device__ __constant__ double kBinom[6][6] = {{1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0},
                                               {1, 3, 3, 1, 0, 0}, {1, 4, 6, 4, 1, 0}, {1, 5, 10, 10, 5, 1}};

// NV = stencil size (2, 4 or 6), a compile-time constant so that the cardinal functions and the contraction live in
// registers and unroll (the generic-size version needed 256 VGPRs and spilled)
template <int NV>
struct Card { double L[NV], dL[NV], d2L[NV]; };

// degree-m Bernstein basis value with out-of-range indices = 0
__device__ __forceinline__ double bern(int m, int i, const double* tp, const double* sp) {
    return (i < 0 || i > m || m < 0) ? 0.0 : kBinom[m][i] * tp[i] * sp[m - i];
}
// cardinal functions of one dimension at local coordinate t, derivatives w.r.t. x (1/h folded in)
template <int NV>
__device__ __forceinline__ void cardinal(const ReinitArgs& a, double t, double invh, bool second, Card<NV>& o) {
#pragma clang fp contract(fast)
    const int n = a.order;       // NV - 1 (odd orders) or NV - 2 (even orders: least-squares fit of a larger stencil)
    double tp[NV], sp[NV];
    tp[0] = sp[0] = 1.0;
#pragma unroll
    for (int i = 1; i < NV; ++i) { tp[i] = tp[i - 1] * t; sp[i] = sp[i - 1] * (1.0 - t); }
    double B[NV], dB[NV], d2B[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool in = i <= n;
        B[i] = in ? bern(n, i, tp, sp) : 0.0;
        dB[i] = in ? n * (bern(n - 1, i - 1, tp, sp) - bern(n - 1, i, tp, sp)) : 0.0;
        d2B[i] = in && second ? n * (n - 1) * (bern(n - 2, i - 2, tp, sp) - 2.0 * bern(n - 2, i - 1, tp, sp) + bern(n - 2, i, tp, sp)) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        double l = 0.0, dl = 0.0, d2l = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double m = i <= n ? a.M[i * NV + j] : 0.0;
            l += m * B[i]; dl += m * dB[i]; d2l += m * d2B[i];
        }
        o.L[j] = l; o.dL[j] = dl * invh; o.d2L[j] = d2l * invh * invh;
    }
}

// The stencil values of the patch of cell I.  NV <= 4: held in registers by the caller across the iterations of a solve on one
// cell (64 doubles in 3-D) — re-loading them per evaluation was 64 gathers in front of every Newton step; NV = 6 (216 values) reads
// them from memory at every evaluation.
// ND > 0: the dimension as a compile-time constant (every loop over dimensions and stencil rows unrolls without selects); ND = 0: a.ndim
template <int ND>
__device__ __forceinline__ int ndim_of(const ReinitArgs& a) {
    if constexpr (ND > 0) return ND;
    else return a.ndim;
}
template <int NV, int ND = 0, bool REGS = (NV <= 4)>
struct PatchValues {
    static constexpr bool IN_REGS = REGS;
    double v[IN_REGS ? NV * NV * NV : 1];
    long long q0;
    __device__ __forceinline__ void load(const ReinitArgs& a, const int I[3]) {
        const int nd = ndim_of<ND>(a);
        q0 = a.origin + (I[0] + a.off) + (nd > 1 ? (I[1] + a.off) * a.s1 : 0) + (nd > 2 ? (I[2] + a.off) * a.s2 : 0);
        if constexpr (IN_REGS) {
            const int nv1 = nd > 1 ? NV : 1, nv2 = nd > 2 ? NV : 1;
#pragma unroll
            for (int j2 = 0; j2 < NV; ++j2) {
                if (j2 >= nv2) break;
#pragma unroll
                for (int j1 = 0; j1 < NV; ++j1) {
                    if (j1 >= nv1) break;
#pragma unroll
                    for (int j0 = 0; j0 < NV; ++j0) v[j0 + NV * (j1 + NV * j2)] = ld_val(a.phi, q0 + j0 + j1 * a.s1 + j2 * a.s2, a.f32);
                }
            }
        }
    }
    __device__ __forceinline__ double at(const ReinitArgs& a, int j0, int j1, int j2) const {
        if constexpr (IN_REGS) return v[j0 + NV * (j1 + NV * j2)];
        else return ld_val(a.phi, q0 + j0 + j1 * a.s1 + j2 * a.s2, a.f32);
    }
};

// value, gradient and (optionally) Hessian {00,11,22,01,02,12} of the patch of cell I at x
template <int NV, int ND, bool REGS>
__device__ __forceinline__ void patch_eval(const ReinitArgs& a, const PatchValues<NV, ND, REGS>& pv, const int I[3], const double x[3], bool second, double& val,
                                           double g[3], double H[6]) {
#pragma clang fp contract(fast)
    const int nd = ndim_of<ND>(a);
    Card<NV> c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d < nd) {
            const double t = (x[d] - (a.lc[d] + (double)(I[d] + a.goff[d]) * a.h[d])) / a.h[d];
            cardinal<NV>(a, t, 1.0 / a.h[d], second, c[d]);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) { c[d].L[j] = j == 0 ? 1.0 : 0.0; c[d].dL[j] = 0.0; c[d].d2L[j] = 0.0; }
        }
    }
    const int nv1 = nd > 1 ? NV : 1, nv2 = nd > 2 ? NV : 1;
    val = 0.0; g[0] = g[1] = g[2] = 0.0;
    for (int k = 0; k < 6; ++k) H[k] = 0.0;
#pragma unroll
    for (int j2 = 0; j2 < NV; ++j2) {
        if (j2 >= nv2) break;
#pragma unroll
        for (int j1 = 0; j1 < NV; ++j1) {
            if (j1 >= nv1) break;
            double s = 0.0, ds = 0.0, d2s = 0.0;
#pragma unroll
            for (int j0 = 0; j0 < NV; ++j0) {
                const double v = pv.at(a, j0, j1, j2);
                s += v * c[0].L[j0]; ds += v * c[0].dL[j0]; d2s += v * c[0].d2L[j0];
            }
            const double L1 = c[1].L[j1], D1 = c[1].dL[j1], E1 = c[1].d2L[j1];
            const double L2 = c[2].L[j2], D2 = c[2].dL[j2], E2 = c[2].d2L[j2];
            val += s * L1 * L2;
            g[0] += ds * L1 * L2; g[1] += s * D1 * L2; g[2] += s * L1 * D2;
            if (second) {
                H[0] += d2s * L1 * L2; H[1] += s * E1 * L2; H[2] += s * L1 * E2;
                H[3] += ds * D1 * L2; H[4] += ds * L1 * D2; H[5] += s * D1 * D2;
            }
        }
    }
}
// one evaluation on the patch of cell I (values loaded for it)
template <int NV>
__device__ void patch_eval(const ReinitArgs& a, const int I[3], const double x[3], bool second, double& val, double g[3], double H[6]) {
    PatchValues<NV, 0> pv;
    pv.load(a, I);
    patch_eval(a, pv, I, x, second, val, g, H);
}

// compute_index (src/meshes.jl:155-167): cell containing x, clamped to the grid
template <int ND = 0>
__device__ __forceinline__ void cell_of(const ReinitArgs& a, const double x[3], int I[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d >= ndim_of<ND>(a)) { I[d] = 0; continue; }
        int i = (int)floor((x[d] - a.lc[d]) / a.h[d]) - a.goff[d];
        I[d] = i < 0 ? 0 : (i > a.n[d] - 2 ? a.n[d] - 2 : i);
    }
}
__device__ __forceinline__ long long cell_lin(const ReinitArgs& a, const int I[3]) {   // over the (n-1)^N cells
    return I[0] + (long long)(a.n[0] - 1) * (I[1] + (long long)(a.ndim > 1 ? a.n[1] - 1 : 1) * I[2]);
}
__device__ __forceinline__ long long ncells(const ReinitArgs& a) {
    return (long long)(a.n[0] - 1) * (a.ndim > 1 ? a.n[1] - 1 : 1) * (a.ndim > 2 ? a.n[2] - 1 : 1);
}
__device__ __forceinline__ void cell_unlin(const ReinitArgs& a, long long c, int I[3]) {
    const long long c0 = a.n[0] - 1, c1 = a.ndim > 1 ? a.n[1] - 1 : 1;
    I[0] = (int)(c % c0); I[1] = (int)((c / c0) % c1); I[2] = (int)(c / (c0 * c1));
}

// extrema of the Bernstein coefficients of the patch whose stencil starts at padded index q0 (cell_extrema,
// src/interpolation.jl:262-265): M applied along every dimension in turn
__device__ void bernstein_extrema(const ReinitArgs& a, long long q0, double& lo, double& hi) {
    const int nv = a.nv, nc = a.order + 1;
    const int e1 = a.ndim > 1 ? 1 : 0, e2 = a.ndim > 2 ? 1 : 0;
    const int v1 = e1 ? nv : 1, v2 = e2 ? nv : 1, c1n = e1 ? nc : 1, c2n = e2 ? nc : 1;
    double buf0[216], buf1[216];
    for (int j2 = 0; j2 < v2; ++j2)
        for (int j1 = 0; j1 < v1; ++j1)
            for (int j0 = 0; j0 < nv; ++j0) buf0[j0 + nv * (j1 + v1 * j2)] = ld_val(a.phi, q0 + j0 + j1 * a.s1 + j2 * a.s2, a.f32);
    // dimension 0: (nv, v1, v2) -> (nc, v1, v2)
    for (int j2 = 0; j2 < v2; ++j2)
        for (int j1 = 0; j1 < v1; ++j1)
            for (int i = 0; i < nc; ++i) {
                double sacc = 0.0;
                for (int j = 0; j < nv; ++j) sacc += a.M[i * nv + j] * buf0[j + nv * (j1 + v1 * j2)];
                buf1[i + nc * (j1 + v1 * j2)] = sacc;
            }
    // dimension 1: (nc, v1, v2) -> (nc, c1n, v2)
    if (e1) {
        for (int j2 = 0; j2 < v2; ++j2)
            for (int i1 = 0; i1 < nc; ++i1)
                for (int i0 = 0; i0 < nc; ++i0) {
                    double sacc = 0.0;
                    for (int j = 0; j < nv; ++j) sacc += a.M[i1 * nv + j] * buf1[i0 + nc * (j + v1 * j2)];
                    buf0[i0 + nc * (i1 + c1n * j2)] = sacc;
                }
    } else {
        for (int i = 0; i < nc; ++i) buf0[i] = buf1[i];
    }
    // dimension 2: (nc, c1n, v2) -> (nc, c1n, c2n), extrema on the fly
    lo = __builtin_inf(); hi = -__builtin_inf();
    for (int i2 = 0; i2 < c2n; ++i2)
        for (int i1 = 0; i1 < c1n; ++i1)
            for (int i0 = 0; i0 < nc; ++i0) {
                double sacc = 0.0;
                if (e2) for (int j = 0; j < nv; ++j) sacc += a.M[i2 * nv + j] * buf0[i0 + nc * (i1 + c1n * j)];
                else sacc = buf0[i0 + nc * i1];
                lo = sacc < lo ? sacc : lo; hi = sacc > hi ? sacc : hi;
            }
}

// The same with the stencil size, the coefficient count and the dimension fixed at compile time (orders 1–3): the NV^N
// stencil values stay in registers (128 VGPRs for 4³) and every loop unrolls — the general version above keeps two
// 216-value arrays per lane in scratch memory.  For each index i_last of the last dimension the stencil is contracted
// along that dimension first (one NV^(N-1) slab), then along the others, and the extrema are taken on the fly.
template <int NV, int NC, int NDIM>
__device__ void bernstein_extrema_reg(const ReinitArgs& a, long long q0, double& lo, double& hi) {
    constexpr int V1 = NDIM > 1 ? NV : 1, V2 = NDIM > 2 ? NV : 1;
    constexpr int C0 = NC, C1 = NDIM > 1 ? NC : 1, CL = NDIM > 2 ? NC : 1;
    double v[NV * V1 * V2];
#pragma unroll
    for (int j2 = 0; j2 < V2; ++j2)
#pragma unroll
        for (int j1 = 0; j1 < V1; ++j1)
#pragma unroll
            for (int j0 = 0; j0 < NV; ++j0) v[j0 + NV * (j1 + V1 * j2)] = ld_val(a.phi, q0 + j0 + j1 * a.s1 + j2 * a.s2, a.f32);
    double m[NC * NV];
#pragma unroll
    for (int k = 0; k < NC * NV; ++k) m[k] = a.M[k];
    lo = __builtin_inf(); hi = -__builtin_inf();
#pragma unroll
    for (int i2 = 0; i2 < CL; ++i2) {
        double t[NV * V1];                       // contracted along dimension 2 (3-D) — or the stencil itself
#pragma unroll
        for (int k = 0; k < NV * V1; ++k) {
            if constexpr (NDIM > 2) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) acc += m[i2 * NV + j] * v[k + NV * V1 * j];
                t[k] = acc;
            } else {
                t[k] = v[k];
            }
        }
        double u[C0 * V1];                       // along dimension 0
#pragma unroll
        for (int j1 = 0; j1 < V1; ++j1)
#pragma unroll
            for (int i0 = 0; i0 < C0; ++i0) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) acc += m[i0 * NV + j] * t[j + NV * j1];
                u[i0 + C0 * j1] = acc;
            }
#pragma unroll
        for (int i1 = 0; i1 < C1; ++i1)          // along dimension 1, extrema on the fly
#pragma unroll
            for (int i0 = 0; i0 < C0; ++i0) {
                double acc;
                if constexpr (NDIM > 1) {
                    acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) acc += m[i1 * NV + j] * u[i0 + C0 * j];
                } else {
                    acc = u[i0];
                }
                lo = acc < lo ? acc : lo; hi = acc > hi ? acc : hi;
            }
    }
}

// ---- 1. candidate cells: active (all corners in the band, src/meshfield.jl:364-369) and not provably empty.
// (a) every cell: the cheap bound |p - mid| <= Λ·spread over the stencil values -> list of "maybe" cells (wave-
//     aggregated append); (b) the maybe cells only (≈6x the final count): the reference's own test on the extrema of
//     the Bernstein coefficients (proven_empty, src/interpolation.jl:271-274) -> candidate ids.
// node_list != NULL (band fields): only the cells whose lowest corner is a listed (active) node are looked at, and
// cand_id has been set to -1 by the caller.
__global__ void __launch_bounds__(256) reinit_cells_kernel(ReinitArgs a, int* cand_id, long long* maybe, unsigned* maybe_count,
                                                           const long long* node_list, DevCount nlist) {
    __shared__ unsigned blk_n, blk_base;
    const long long nc = node_list ? count_of(nlist) : ncells(a);
    const long long span = (nc + 255) / 256 * 256;      // whole workgroups reach the barriers
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < span; w += (long long)gridDim.x * blockDim.x) {
        bool keep = false;
        long long c = w;
        if (w < nc) {
            int I[3];
            bool has_cell = true;
            if (node_list) {
                const long long t = node_list[w];
                I[0] = (int)(t % a.n[0]); I[1] = (int)((t / a.n[0]) % a.n[1]); I[2] = (int)(t / ((long long)a.n[0] * a.n[1]));
                for (int d = 0; d < a.ndim; ++d) has_cell = has_cell && I[d] < a.n[d] - 1;
                c = has_cell ? cell_lin(a, I) : 0;
            } else {
                cell_unlin(a, c, I);
            }
            const long long qc = a.origin + I[0] + I[1] * a.s1 + I[2] * a.s2;
            bool act = has_cell;
            if (a.mask && act)
                for (int k = 0; k < (1 << a.ndim); ++k)
                    act = act && a.mask[qc + (k & 1) + ((k >> 1) & 1) * a.s1 + ((k >> 2) & 1) * a.s2];
            if (act) {
                const int nv1 = a.ndim > 1 ? a.nv : 1, nv2 = a.ndim > 2 ? a.nv : 1;
                const long long q0 = qc + a.off + (a.ndim > 1 ? a.off * a.s1 : 0) + (a.ndim > 2 ? a.off * a.s2 : 0);
                double lo = __builtin_inf(), hi = -__builtin_inf();
                for (int j2 = 0; j2 < nv2; ++j2)
                    for (int j1 = 0; j1 < nv1; ++j1)
                        for (int j0 = 0; j0 < a.nv; ++j0) {
                            const double v = ld_val(a.phi, q0 + j0 + j1 * a.s1 + j2 * a.s2, a.f32);
                            lo = v < lo ? v : lo; hi = v > hi ? v : hi;
                        }
                const double mid = 0.5 * (lo + hi), spread = 0.5 * (hi - lo);
                const bool empty = mid - a.lambda * spread > 0.0 || mid + a.lambda * spread < 0.0;   // p cannot vanish
                keep = !empty && lo == lo && hi == hi;
            }
            if (!node_list) cand_id[c] = -1;
        }
        // one append per workgroup (every thread of it reaches this point: `span` is a multiple of 256)
        const unsigned long long bal = __ballot(keep);
        const int lane = threadIdx.x & 63;
        if (threadIdx.x == 0) blk_n = 0;
        __syncthreads();
        unsigned wbase = 0;
        if (bal && lane == 0) wbase = atomicAdd(&blk_n, (unsigned)__popcll(bal));
        wbase = __shfl(wbase, 0, 64);
        __syncthreads();
        if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(maybe_count, blk_n);
        __syncthreads();
        if (keep) maybe[blk_base + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = c;
    }
}
template <int NV, int NC, int NDIM>   // NV = 0: the general version
__global__ void __launch_bounds__(256) reinit_cells2_kernel(ReinitArgs a, const long long* maybe, const unsigned* maybe_count, int* cand_id,
                                                            long long* cand_cell, unsigned* cand_count, unsigned cand_cap) {
    const unsigned nmaybe = *maybe_count;          // stays on the device: the launch does not wait for it
    __shared__ unsigned blk_n, blk_base;
    const int lane = threadIdx.x & 63;
    for (unsigned base = blockIdx.x * blockDim.x; base < nmaybe; base += gridDim.x * blockDim.x) {     // uniform per workgroup
        const unsigned i = base + threadIdx.x;
        bool keep = false;
        long long c = 0;
        if (i < nmaybe) {
            c = maybe[i];
            int I[3];
            cell_unlin(a, c, I);
            const long long q0 = a.origin + (I[0] + a.off) + (a.ndim > 1 ? (I[1] + a.off) * a.s1 : 0) + (a.ndim > 2 ? (I[2] + a.off) * a.s2 : 0);
            double clo, chi;
            if constexpr (NV == 0) bernstein_extrema(a, q0, clo, chi);
            else bernstein_extrema_reg<NV, NC, NDIM>(a, q0, clo, chi);
            keep = !(clo * chi > 0.0);
        }
        // candidate ids: one append to the counter per workgroup
        const unsigned long long bal = __ballot(keep);
        if (threadIdx.x == 0) blk_n = 0;
        __syncthreads();
        unsigned wbase = 0;
        if (bal && lane == 0) wbase = atomicAdd(&blk_n, (unsigned)__popcll(bal));
        wbase = __shfl(wbase, 0, 64);
        __syncthreads();
        if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(cand_count, blk_n);
        __syncthreads();
        if (keep) {
            // a candidate beyond what the per-candidate buffers (samples, counts) hold is counted — the host repeats the round with larger
            // buffers — and otherwise left alone: no id may point outside them
            const unsigned id = blk_base + wbase + __popcll(bal & ((1ull << lane) - 1ull));
            if (id < cand_cap) {
                cand_id[c] = (int)id;
                cand_cell[id] = c;
            }
        }
    }
}

// ---- 2. interface samples.  Every candidate cell has (upsample + 1)^N start points on its closure, and a start point on a
// face, edge or corner belongs to up to 2^N cells: its projection follows the iterate across cells whichever cell it was started
// for, so the copies run the same trajectory and only the copy of the cell the point lands in is kept (src/sdf.jl:186-236).  Here
// each start point is projected once: (a) the copy of the lowest candidate cell sharing a point registers it; (b) one thread per
// registered point projects it and, where it converges into a candidate cell whose closure holds the start point, stores it in
// that cell's slot of that start point — the reference's sample set, with 8 instead of 27 projections per cell in 3-D.  A cell
// computes a start point from its own corner, the copies of two sharing cells may differ in the last bit, and a point ON a cell
// boundary picks the patch of its first step by that bit (projections from the two patches end 1e-5 apart along the interface): two
// cells are served by one projection only where their copies pick the same first cell (≈85 % of the shared coordinates); the
// shared projection then differs from a cell's own by the last bit of its start.
// coordinate d of start point xi (0..upsample) of the cell with lower corner index c, as the reference's cell computes it:
// cell.lc .+ (cell.hc .- cell.lc) .* ξ ./ upsample with cell.lc = g.lc + (I - 1)·h, cell.hc = cell.lc + h (src/sdf.jl:167,
// src/meshes.jl:114-117,183-184) — the cell's width is the difference of its corners, not h
__device__ __forceinline__ double start_coord(const ReinitArgs& a, int d, int c, int xi) {
    const double lo = a.lc[d] + (double)(c + a.goff[d]) * a.h[d];
    const double hi = lo + a.h[d];
    return lo + (hi - lo) * (double)xi / (double)a.upsample;
}
// index along d of the cell whose patch the first projection step from coordinate x uses (cell_of, one dimension)
__device__ __forceinline__ int first_cell(const ReinitArgs& a, int d, double x) {
    const int i = (int)floor((x - a.lc[d]) / a.h[d]) - a.goff[d];
    return i < 0 ? 0 : (i > a.n[d] - 2 ? a.n[d] - 2 : i);
}
__global__ void __launch_bounds__(256) reinit_starts_kernel(ReinitArgs a, const long long* cand_cell, const int* cand_id, DevCount ncand_, int S,
                                                            unsigned long long* starts, unsigned* nstarts) {
    const long long ncand = count_of(ncand_);
    // a workgroup collects the registered points of 1024 (cell, start point) pairs in LDS and appends them with ONE atomic: a wave-level
    // append to the single counter took 375 µs for 16 k waves
    constexpr int CHUNK = 4;
    __shared__ unsigned long long buf[256 * CHUNK];
    __shared__ unsigned nbuf, gbase;
    const long long total = (long long)ncand * S;
    const int up = a.upsample, up1 = up + 1;
    const int nc_[3] = {a.n[0] - 1, a.ndim > 1 ? a.n[1] - 1 : 1, a.ndim > 2 ? a.n[2] - 1 : 1};
    const int lane = threadIdx.x & 63;
    for (long long sc = blockIdx.x; sc * (256 * CHUNK) < total; sc += gridDim.x) {
        if (threadIdx.x == 0) nbuf = 0;
        __syncthreads();
        for (int c = 0; c < CHUNK; ++c) {
            const long long w = (sc * CHUNK + c) * 256 + threadIdx.x;
            bool own = false;
            unsigned long long rec = 0;
            if (w < total) {
                const long long id = w / S;
                const int s = (int)(w - id * S);
                int I[3];
                cell_unlin(a, cand_cell[id], I);
                const int xi[3] = {s % up1, (s / up1) % up1, s / (up1 * up1)};
                // the cells sharing the point: I + δ, δ_d ∈ {-1, 0} where the point lies on the cell's lower face of dimension d,
                // {0, +1} on its upper face; a copy is dropped when a candidate cell that precedes I (last dimension first) shares it
                // (a neighbour computes the shared point from its own corner; the two may differ in the last bit, and a point ON a cell
                // boundary picks the patch of its first step by that bit: the copies are one projection only where they pick the same)
                int lo[3], hi[3];
                for (int d = 0; d < 3; ++d) {
                    lo[d] = (d < a.ndim && xi[d] == 0 && I[d] > 0 &&
                             first_cell(a, d, start_coord(a, d, I[d] - 1, up)) == first_cell(a, d, start_coord(a, d, I[d], 0))) ? -1 : 0;
                    hi[d] = (d < a.ndim && xi[d] == up && I[d] < nc_[d] - 1 &&
                             first_cell(a, d, start_coord(a, d, I[d] + 1, 0)) == first_cell(a, d, start_coord(a, d, I[d], up))) ? 1 : 0;
                }
                own = true;
                for (int d2 = lo[2]; d2 <= hi[2] && own; ++d2)
                    for (int d1 = lo[1]; d1 <= hi[1] && own; ++d1)
                        for (int d0 = lo[0]; d0 <= hi[0] && own; ++d0) {
                            const bool lower = d2 < 0 || (d2 == 0 && (d1 < 0 || (d1 == 0 && d0 < 0)));
                            if (!lower) continue;
                            const int C[3] = {I[0] + d0, I[1] + d1, I[2] + d2};
                            if (cand_id[cell_lin(a, C)] >= 0) own = false;
                        }
                rec = (unsigned long long)id | ((unsigned long long)s << 32);
            }
            const unsigned long long bal = __ballot(own);
            if (bal) {
                const int leader = __ffsll((long long)bal) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&nbuf, (unsigned)__popcll(bal));
                base = __shfl(base, leader, 64);
                if (own) buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = rec;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) gbase = atomicAdd(nstarts, nbuf);
        __syncthreads();
        for (unsigned i = threadIdx.x; i < nbuf; i += 256) starts[gbase + i] = buf[i];
        __syncthreads();
    }
}
template <int NV, int ND>      // (two workgroups per CU: at most 256 registers)
__global__ void __launch_bounds__(256, 2) reinit_sample_kernel(ReinitArgs a, const long long* cand_cell, const int* cand_id, int S,
                                                                                  const unsigned long long* starts, const unsigned* nstarts,
                                                                                  double* pts, unsigned char* valid) {
    const int nd = ndim_of<ND>(a);
    const long long total = (long long)*nstarts;
    double hmax = a.h[0];
    for (int d = 1; d < nd; ++d) hmax = a.h[d] > hmax ? a.h[d] : hmax;
    const int up = a.upsample, up1 = up + 1;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        const unsigned long long rec = starts[w];
        const long long id = (long long)(rec & 0xffffffffull);
        const int s = (int)(rec >> 32);
        int I[3];
        cell_unlin(a, cand_cell[id], I);
        const int xi[3] = {s % up1, (s / up1) % up1, s / (up1 * up1)};
        double x0[3] = {0, 0, 0}, x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < nd) x0[d] = start_coord(a, d, I[d], xi[d]);
        x[0] = x0[0]; x[1] = x0[1]; x[2] = x0[2];
        bool conv = false;
        PatchValues<NV, ND> pv;
        int Jp[3] = {-1, -1, -1};
        for (int it = 0; it < a.maxiters; ++it) {      // _project_to_interface (src/sdf.jl:223-236)
            int J[3];
            cell_of<ND>(a, x, J);
            if (J[0] != Jp[0] || J[1] != Jp[1] || J[2] != Jp[2]) { pv.load(a, J); Jp[0] = J[0]; Jp[1] = J[1]; Jp[2] = J[2]; }
            double val, g[3], H[6];
            patch_eval(a, pv, J, x, false, val, g, H);
            if (fabs(val) < a.ftol) { conv = true; break; }
            const double g2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (g2 == 0.0 || !(g2 == g2)) break;
            double dist2 = 0.0;
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if (d < nd) { x[d] = x[d] - val * g[d] / g2; dist2 += (x[d] - x0[d]) * (x[d] - x0[d]); }
            if (sqrt(dist2) > hmax) break;
        }
        if (!conv) continue;
        // the cell the point landed in keeps it — if it is a candidate and the start point is one of its own
        int J[3];
        cell_of<ND>(a, x, J);
        const int jid = cand_id[cell_lin(a, J)];
        if (jid < 0) continue;
        int sj = 0, mul = 1;
        bool mine = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xj = d < nd ? up * (I[d] - J[d]) + xi[d] : 0;     // the start point's index on cell J
            mine = mine && xj >= 0 && xj <= up && (d >= nd || J[d] == I[d] || first_cell(a, d, start_coord(a, d, J[d], xj)) == first_cell(a, d, x0[d]));
            sj += xj * mul;
            mul *= up1;
        }
        if (!mine) continue;
        const long long slot = (long long)jid * S + sj;
        valid[slot] = 1;
        pts[3 * slot] = x[0]; pts[3 * slot + 1] = x[1]; pts[3 * slot + 2] = x[2];
    }
}

// per candidate cell: move its kept samples to the front of its S slots, count them, and flag the coarse block
// (RB^N cells) of every cell that holds a sample — the far-field search walks blocks, not cells
constexpr int RB = 8;
__device__ __forceinline__ long long blk_lin(const ReinitArgs& a, const int B[3]) {
    const long long b0 = (a.n[0] - 1 + RB - 1) / RB, b1 = a.ndim > 1 ? (a.n[1] - 1 + RB - 1) / RB : 1;
    return B[0] + b0 * (B[1] + b1 * B[2]);
}
__device__ __forceinline__ long long bits_row(const ReinitArgs& a, int c1, int c2) {   // first word of the cell row (c1, c2)
    const long long wpr = (a.n[0] - 1 + 63) / 64, c1n = a.ndim > 1 ? a.n[1] - 1 : 1;
    return wpr * (c1 + c1n * (long long)c2);
}
__global__ void __launch_bounds__(256) reinit_compact_kernel(ReinitArgs a, const long long* cand_cell, DevCount ncand_, int S, double* pts,
                                                             const unsigned char* valid, unsigned char* cnt, unsigned char* blk,
                                                             unsigned long long* bits) {
    const unsigned ncand = (unsigned)count_of(ncand_);
    auto mark = [&](unsigned id) {
        int I[3];
        cell_unlin(a, cand_cell[id], I);
        const int B[3] = {I[0] / RB, I[1] / RB, I[2] / RB};
        blk[blk_lin(a, B)] = 1;                                                        // one byte per block of RB^N cells with samples
        atomicOr(bits + bits_row(a, I[1], I[2]) + I[0] / 64, 1ull << (I[0] & 63));     // one bit per cell with samples
    };
    if (S <= 32) {
        // half a wave per cell, a lane per slot: the kept samples move to the front in slot order (one thread per cell walked its
        // S slots through a chain of dependent loads — 53 µs for 38 k cells)
        const int lane = threadIdx.x & 63, sub = lane & 31, half = lane >> 5;
        const unsigned per_block = blockDim.x / 32;
        for (unsigned base = blockIdx.x * per_block; base < ncand; base += gridDim.x * per_block) {      // uniform trip count per block
            const unsigned id = base + (threadIdx.x >> 5);
            const bool in = id < ncand && sub < S;
            const long long slot = (long long)id * S + sub;
            const bool keep = in && valid[slot];
            double x0 = 0.0, x1 = 0.0, x2 = 0.0;
            if (keep) { x0 = pts[3 * slot]; x1 = pts[3 * slot + 1]; x2 = pts[3 * slot + 2]; }
            const unsigned kept = (unsigned)((__ballot(keep) >> (32 * half)) & 0xffffffffull);
            const int m = __popc(kept), rank = __popc(kept & ((1u << sub) - 1u));
            // every lane of the half-wave has read its sample before any writes (the loads above are complete: the ballot depends on them)
            __builtin_amdgcn_wave_barrier();
            if (keep && rank != sub) {
                const long long dst = (long long)id * S + rank;
                pts[3 * dst] = x0; pts[3 * dst + 1] = x1; pts[3 * dst + 2] = x2;
            }
            if (id < ncand && sub == 0) {
                cnt[id] = (unsigned char)m;
                if (m) mark(id);
            }
        }
        return;
    }
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < ncand; id += gridDim.x * blockDim.x) {
        int m = 0;
        for (int k = 0; k < S; ++k) {
            const long long slot = (long long)id * S + k;
            if (!valid[slot]) continue;
            if (k != m) {
                const long long dst = (long long)id * S + m;
                pts[3 * dst] = pts[3 * slot]; pts[3 * dst + 1] = pts[3 * slot + 1]; pts[3 * dst + 2] = pts[3 * slot + 2];
            }
            ++m;
        }
        cnt[id] = (unsigned char)(m > 255 ? 255 : m);
        if (m) mark(id);
    }
}

// (N+1)x(N+1) solve with partial pivoting; returns false if singular
__device__ bool solve_small(int m, double A[4][4], double b[4]) {
    for (int k = 0; k < m; ++k) {
        int p = k;
        double best = fabs(A[k][k]);
        for (int r = k + 1; r < m; ++r)
            if (fabs(A[r][k]) > best) { best = fabs(A[r][k]); p = r; }
        if (best == 0.0 || !(best == best)) return false;
        if (p != k) {
            for (int cc = 0; cc < m; ++cc) { const double t = A[k][cc]; A[k][cc] = A[p][cc]; A[p][cc] = t; }
            const double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        for (int r = k + 1; r < m; ++r) {
            const double f = A[r][k] / A[k][k];
            for (int cc = k; cc < m; ++cc) A[r][cc] -= f * A[k][cc];
            b[r] -= f * b[k];
        }
    }
    for (int k = m - 1; k >= 0; --k) {
        double sacc = b[k];
        for (int cc = k + 1; cc < m; ++cc) sacc -= A[k][cc] * b[cc];
        b[k] = sacc / A[k][k];
    }
    return true;
}
// The same with the size fixed at compile time.  The pivot row is brought up by conditional swaps with every row below in turn
// (first largest |entry| wins, as above) — the rows below end up in another order, which changes nothing: each is reduced with the
// same pivot row by the same operations — so no register array is indexed by a per-lane row number (the version above becomes a
// serialised loop over the distinct pivot rows of a wave at every access).
template <int M>
__device__ __forceinline__ bool solve_small_static(double A[4][4], double b[4]) {
#pragma unroll
    for (int k = 0; k < M; ++k) {
#pragma unroll
        for (int r = k + 1; r < M; ++r) {
            const bool sw = fabs(A[r][k]) > fabs(A[k][k]);
#pragma unroll
            for (int cc = 0; cc < M; ++cc) { const double u = A[k][cc], v = A[r][cc]; A[k][cc] = sw ? v : u; A[r][cc] = sw ? u : v; }
            const double u = b[k], v = b[r];
            b[k] = sw ? v : u; b[r] = sw ? u : v;
        }
        const double piv = fabs(A[k][k]);
        if (piv == 0.0 || !(piv == piv)) return false;
#pragma unroll
        for (int r = k + 1; r < M; ++r) {
            const double f = A[r][k] / A[k][k];
#pragma unroll
            for (int cc = k; cc < M; ++cc) A[r][cc] -= f * A[k][cc];
            b[r] -= f * b[k];
        }
    }
#pragma unroll
    for (int k = M - 1; k >= 0; --k) {
        double sacc = b[k];
#pragma unroll
        for (int cc = k + 1; cc < M; ++cc) sacc -= A[k][cc] * b[cc];
        b[k] = sacc / A[k][k];
    }
    return true;
}

// _closest_point (src/sdf.jl:239-272) on the patch of cell I, from x0; returns converged
template <int NV, int ND>
__device__ bool closest_on_patch(const ReinitArgs& a, const int I[3], const double xq[3], const double x0[3], double safeguard, double cp[3]) {
    const int N = ndim_of<ND>(a);
    double val, g[3], H[6];
    PatchValues<NV, ND> pv;
    pv.load(a, I);
    patch_eval(a, pv, I, x0, false, val, g, H);
    double g2 = 0.0, num = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < N) { g2 += g[d] * g[d]; num += (xq[d] - x0[d]) * g[d]; }
    double lam = g2 == 0.0 ? 0.0 : num / g2;
    double x[3] = {x0[0], x0[1], x0[2]};
    double best_res = __builtin_inf();
    cp[0] = x0[0]; cp[1] = x0[1]; cp[2] = x0[2];
    const double reg = 1.4901161193847656e-08;   // sqrt(eps(Float64))
    for (int it = 0; it < a.maxiters; ++it) {
        patch_eval(a, pv, I, x, true, val, g, H);
        double res[4] = {0, 0, 0, 0}, rn2 = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < N) { res[d] = x[d] - xq[d] + lam * g[d]; rn2 += res[d] * res[d]; }
        rn2 += val * val;
        const double rn = sqrt(rn2);
        if (rn < best_res) { best_res = rn; cp[0] = x[0]; cp[1] = x[1]; cp[2] = x[2]; }
        if (fabs(val) < a.ftol && rn < a.xtol) { cp[0] = x[0]; cp[1] = x[1]; cp[2] = x[2]; return true; }
        const double Hm[3][3] = {{H[0], H[3], H[4]}, {H[3], H[1], H[5]}, {H[4], H[5], H[2]}};
        double K[4][4], rhs[4];
        bool ok;
        if constexpr (ND > 0) {
#pragma unroll
            for (int r = 0; r < ND; ++r) {
#pragma unroll
                for (int cc = 0; cc < ND; ++cc) K[r][cc] = (r == cc ? 1.0 : 0.0) + lam * Hm[r][cc] + (r == cc ? reg : 0.0);
                K[r][ND] = g[r]; K[ND][r] = g[r];
                rhs[r] = -res[r];
            }
            K[ND][ND] = reg;
            rhs[ND] = -val;
            ok = solve_small_static<ND + 1>(K, rhs);
        } else {
            res[N] = val;
            for (int r = 0; r < N; ++r) {
                for (int cc = 0; cc < N; ++cc) K[r][cc] = (r == cc ? 1.0 : 0.0) + lam * Hm[r][cc] + (r == cc ? reg : 0.0);
                K[r][N] = g[r]; K[N][r] = g[r];
            }
            K[N][N] = reg;
            for (int r = 0; r <= N; ++r) rhs[r] = -res[r];
            ok = solve_small(N + 1, K, rhs);
        }
        if (!ok) return false;
        double nd2 = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < N) nd2 += rhs[d] * rhs[d];
        const double nd = sqrt(nd2);
        const double alpha = nd > 0.0 ? (safeguard / nd < 1.0 ? safeguard / nd : 1.0) : 1.0;
        double dist2 = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < N) { x[d] += alpha * rhs[d]; dist2 += (x[d] - x0[d]) * (x[d] - x0[d]); }
        lam += alpha * rhs[N];
        if (sqrt(dist2) > safeguard) return false;
    }
    return false;
}

// ---- 3. signed distance of every active node
// band fields: the active nodes as a compact list, so that the closest-point kernel runs with full waves instead of
// one lane in eight (one atomic per wave)
__global__ void __launch_bounds__(256) reinit_nodes_kernel(ReinitArgs a, long long* list, unsigned* count) {
    const long long total = (long long)a.n[0] * a.n[1] * a.n[2];
    const long long span = (total + 255) / 256 * 256;      // whole waves reach the ballot
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < span; t += (long long)gridDim.x * blockDim.x) {
        bool on = false;
        if (t < total) {
            const long long q = a.origin + (t % a.n[0]) + ((t / a.n[0]) % a.n[1]) * a.s1 + (t / ((long long)a.n[0] * a.n[1])) * a.s2;
            on = a.mask[q] != 0;
        }
        const unsigned long long bal = __ballot(on);
        if (!bal) continue;
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)bal) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(bal));
        base = __shfl(base, leader, 64);
        if (on && list) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = t;   // list == NULL: count only
    }
}

// The same list for a band, built from a linear scan of the mask bytes, 16 per load: the band is ~1 % of a 3-D grid and
// the node-indexed scan above spends its time on the other 99 % (one byte load and a 64-bit division per node).  One
// global atomic per workgroup (a global atomic per wave serialises: ~10 ns each).  Mask bytes outside the grid are 0.
__global__ void __launch_bounds__(256) reinit_band_nodes_kernel(ReinitArgs a, long long total, long long* list, long long cap, unsigned* count) {
    __shared__ unsigned blk_n, blk_base;
    const uint4* m4 = reinterpret_cast<const uint4*>(a.mask);        // hipMalloc'ed: 256-byte aligned
    const long long nvec = (total + 15) / 16;
    const long long span = (nvec + 255) / 256 * 256;
    const long long G = LSM_GHOST;
    const long long corner = a.origin - G - (a.ndim > 1 ? G * a.s1 : 0) - (a.ndim > 2 ? G * a.s2 : 0);
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < span; v += (long long)gridDim.x * blockDim.x) {
        unsigned w[4] = {0, 0, 0, 0};
        if (v < nvec) {
            if (16 * v + 16 <= total) { const uint4 x = m4[v]; w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; }
            else for (long long b = 16 * v; b < total; ++b) w[(b - 16 * v) / 4] |= (unsigned)a.mask[b] << (8 * ((b - 16 * v) % 4));
        }
        unsigned n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            n += ((w[k] & 0xffu) != 0) + ((w[k] & 0xff00u) != 0) + ((w[k] & 0xff0000u) != 0) + ((w[k] & 0xff000000u) != 0);
        if (!__syncthreads_or(n != 0)) continue;          // also the barrier that separates two rounds of the counters
        if (threadIdx.x == 0) blk_n = 0;
        __syncthreads();
        unsigned mine = 0;
        if (n) mine = atomicAdd(&blk_n, n);
        __syncthreads();
        if (threadIdx.x == 0) blk_base = atomicAdd(count, blk_n);
        __syncthreads();
        if (n && list) {                                   // list == NULL: count only; entries beyond `cap` are counted, not written
            unsigned at = blk_base + mine;
            for (int k = 0; k < 16; ++k) {
                if (!((w[k >> 2] >> (8 * (k & 3))) & 0xffu)) continue;
                const long long qq = 16 * v + k - corner;   // offset from the lowest ghost corner of the padded array
                long long p0 = qq, p1 = 0, p2 = 0;
                if (a.ndim > 2) { p2 = qq / a.s2; p0 = qq - p2 * a.s2; }
                if (a.ndim > 1) { p1 = p0 / a.s1; p0 = p0 - p1 * a.s1; }
                long long i0 = p0 - G, i1 = a.ndim > 1 ? p1 - G : 0, i2 = a.ndim > 2 ? p2 - G : 0;
                // the ghost entries of a band mask are 0 (include/lsm.h); should one not be, it lands on the nearest node
                i0 = i0 < 0 ? 0 : (i0 > a.n[0] - 1 ? a.n[0] - 1 : i0);
                i1 = i1 < 0 ? 0 : (i1 > a.n[1] - 1 ? a.n[1] - 1 : i1);
                i2 = i2 < 0 ? 0 : (i2 > a.n[2] - 1 ? a.n[2] - 1 : i2);
                if ((long long)at < cap) list[at] = i0 + (long long)a.n[0] * (i1 + (long long)a.n[1] * i2);
                ++at;
            }
        }
    }
}

constexpr int NSEED = 5;
constexpr int FINE_SHELLS = 6;
// The first-order closest-point estimate x - ϕ∇ϕ/|∇ϕ|² (centred differences), iterated on the node-wise linear model
// ϕ_J + ∇ϕ_J·(x - x_J), J the node nearest to the iterate: cheap, and within a cell of the interface even when ϕ is far from a
// distance function.  Returns false when there is no usable estimate.
template <int ND>
__device__ __forceinline__ bool first_order_foot(const ReinitArgs& a, const double xq[3], double hmin, double xe[3]) {
    xe[0] = xq[0]; xe[1] = xq[1]; xe[2] = xq[2];
    bool have = true;
    for (int it = 0; it < 5 && have; ++it) {
        int J[3] = {0, 0, 0};
        long long qj = a.origin;
        for (int d = 0; d < ND; ++d) {
            int j = (int)floor((xe[d] - a.lc[d]) / a.h[d] + 0.5) - a.goff[d];
            j = j < 0 ? 0 : (j > a.n[d] - 1 ? a.n[d] - 1 : j);
            J[d] = j;
            qj += (long long)j * (d == 0 ? 1 : (d == 1 ? a.s1 : a.s2));
        }
        if (a.mask && !a.mask[qj]) break;              // left the band: keep the last iterate
        const double vj = ld_val(a.phi, qj, a.f32);
        double g[3] = {0, 0, 0}, g2 = 0.0, val = vj;
        for (int d = 0; d < ND; ++d) {
            const long long sd = d == 0 ? 1 : (d == 1 ? a.s1 : a.s2);
            g[d] = (ld_val(a.phi, qj + sd, a.f32) - ld_val(a.phi, qj - sd, a.f32)) / (2.0 * a.h[d]);
            g2 += g[d] * g[d];
            val += g[d] * (xe[d] - (a.lc[d] + (double)(J[d] + a.goff[d]) * a.h[d]));
        }
        if (!(g2 > 0.0) || !(val == val)) { have = it > 0; break; }
        for (int d = 0; d < ND; ++d) xe[d] -= val * g[d] / g2;
        if (val * val < 0.0625 * hmin * hmin * g2) break;   // within a quarter cell of the model's zero
    }
    return have;
}
// the estimate's cell per node, one lane per node (21 bits per index; -1 = no usable estimate): the 16 lanes that share a node in the
// search below would all compute the same five dependent steps — a third of that kernel's instructions
template <int ND>
__global__ void __launch_bounds__(256) reinit_foot_kernel(ReinitArgs a, const long long* node_list, DevCount nlist, long long* foot) {
    const long long total = node_list ? count_of(nlist) : (long long)a.n[0] * a.n[1] * a.n[2];
    double hmin = a.h[0];
    for (int d = 1; d < ND; ++d) hmin = a.h[d] < hmin ? a.h[d] : hmin;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        const long long t = node_list ? node_list[w] : w;
        const int I[3] = {(int)(t % a.n[0]), (int)((t / a.n[0]) % a.n[1]), (int)(t / ((long long)a.n[0] * a.n[1]))};
        double xq[3] = {0, 0, 0}, xe[3];
        for (int d = 0; d < ND; ++d) xq[d] = a.lc[d] + (double)(I[d] + a.goff[d]) * a.h[d];
        long long out = -1;
        if (first_order_foot<ND>(a, xq, hmin, xe)) {
            int E[3];
            cell_of(a, xe, E);
            out = (long long)E[0] | ((long long)E[1] << 21) | ((long long)E[2] << 42);
        }
        foot[w] = out;
    }
}

// (a0) the guided search, GRP lanes per node.  One lane per node spends milliseconds on its chain of dependent loads
// (row word -> cell id -> sample count -> samples); sixteen lanes share the rows of the ball, keep one best each and
// merge with shuffles.  Nodes the guided search cannot settle (no usable estimate, ball wider than 10 cells) are
// flagged for the one-lane-per-node kernel below (seeds[0] = -2).
#ifndef LSM_REINIT_GRP
#define LSM_REINIT_GRP 16
#endif
#ifndef LSM_REINIT_QCAP
#define LSM_REINIT_QCAP 96
#endif
constexpr int GRP = LSM_REINIT_GRP;
constexpr int QCAP = LSM_REINIT_QCAP;
template <int ND>
__global__ void __launch_bounds__(256) reinit_search_group_kernel(ReinitArgs a, const int* cand_id, int S, const double* pts,
                                                                  const unsigned char* cnt, const unsigned long long* bits,
                                                                  const long long* node_list, DevCount nlist, const long long* foot,
                                                                  long long* seeds) {
    const long long total = node_list ? count_of(nlist) : (long long)a.n[0] * a.n[1] * a.n[2];
    double hmin = a.h[0];
    for (int d = 1; d < ND; ++d) hmin = a.h[d] < hmin ? a.h[d] : hmin;
    const int nc_[3] = {a.n[0] - 1, ND > 1 ? a.n[1] - 1 : 1, ND > 2 ? a.n[2] - 1 : 1};
    const int lane = threadIdx.x & 63, gl = lane % GRP, gbase = lane - gl;
    const long long ngroups = (long long)gridDim.x * (blockDim.x / GRP);
    __shared__ long long qcell[256 / GRP][QCAP];
    __shared__ unsigned qn[256 / GRP];
    for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / GRP; w < total; w += ngroups) {
        const long long t = node_list ? node_list[w] : w;
        const int I[3] = {(int)(t % a.n[0]), (int)((t / a.n[0]) % a.n[1]), (int)(t / ((long long)a.n[0] * a.n[1]))};
        double xq[3] = {0, 0, 0};
        for (int d = 0; d < ND; ++d) xq[d] = a.lc[d] + (double)(I[d] + a.goff[d]) * a.h[d];
        double bd = __builtin_inf();        // this lane's nearest sample so far
        long long bslot = -1;
        double bound = __builtin_inf();     // the group's (pruning only)
        auto scan_cell = [&](int c0, int c1, int c2) {
            const int c[3] = {c0, c1, c2};
            double bx = 0.0;                 // squared distance from the node to the cell
            for (int d = 0; d < ND; ++d) {
                const int gap = I[d] < c[d] ? c[d] - I[d] : (I[d] > c[d] + 1 ? I[d] - (c[d] + 1) : 0);
                bx += (double)gap * a.h[d] * ((double)gap * a.h[d]);
            }
            if (bx > bd || bx > bound) return;
            const int id = cand_id[cell_lin(a, c)];
            if (id < 0) return;
            const int m = cnt[id];
            for (int k = 0; k < m; ++k) {
                const long long slot = (long long)id * S + k;
                double d2 = 0.0;
                for (int d = 0; d < ND; ++d) { const double e = pts[3 * slot + d] - xq[d]; d2 += e * e; }
                if (d2 < bd) { bd = d2; bslot = slot; }
            }
        };
        auto group_min = [&](double v) {
            for (int off = GRP / 2; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, GRP); v = o < v ? o : v; }
            return v;
        };
        const long long fe = foot[w];                 // the estimate's cell (reinit_foot_kernel)
        const bool have = fe >= 0;
        const int E[3] = {(int)(fe & 0x1fffff), (int)((fe >> 21) & 0x1fffff), (int)((fe >> 42) & 0x1fffff)};
        if (have) {     // the 3^N cells around the estimate's cell, spread over the lanes (two each in 3-D)
            const int ncell = ND > 2 ? 27 : (ND > 1 ? 9 : 3);
            for (int r = gl; r < ncell; r += GRP) {
                const int c0 = E[0] + r % 3 - 1, c1 = ND > 1 ? E[1] + (r / 3) % 3 - 1 : 0, c2 = ND > 2 ? E[2] + r / 9 - 1 : 0;
                if (c0 < 0 || c0 >= nc_[0] || c1 < 0 || c1 >= nc_[1] || c2 < 0 || c2 >= nc_[2]) continue;
                scan_cell(c0, c1, c2);
            }
        }
        bound = group_min(bd);
        const double R0 = sqrt(bound);
        if (!(R0 <= 10.0 * hmin)) {            // also when bound == inf
            if (gl == 0) seeds[NSEED * w] = -2;
            continue;
        }
        // every occupied cell that meets the ball of radius R0 around the node: the lanes first collect them from the
        // occupancy rows (c1, c2) into a queue in LDS, then take one cell each — three levels of dependent loads per
        // node (row word, cell record, samples) instead of a chain per row
        {
            const int g = threadIdx.x / GRP;
            if (gl == 0) qn[g] = 0;
            __builtin_amdgcn_wave_barrier();
            auto gap = [&](int d, int c) { return c > I[d] ? c - I[d] : (c + 1 < I[d] ? I[d] - (c + 1) : 0); };
            // rows (c1, c2) whose gap to the node is at most floor(R0 / h) cells: a row one further lies more than R0 away
            const int k2 = ND > 2 ? (int)(R0 / a.h[2]) : 0, k1 = ND > 1 ? (int)(R0 / a.h[1]) : 0;
            const int n1 = ND > 1 ? 2 * k1 + 2 : 1, n2 = ND > 2 ? 2 * k2 + 2 : 1;
            const float inv_n1 = 1.0f / (float)n1, inv_h0 = (float)(1.0 / a.h[0]);
            for (int idx = gl; idx < n1 * n2; idx += GRP) {
                const int q2 = (int)(((float)idx + 0.5f) * inv_n1), q1 = idx - q2 * n1;     // idx / n1, idx % n1 (small integers: exact)
                const int c1 = ND > 1 ? I[1] - k1 - 1 + q1 : 0, c2 = ND > 2 ? I[2] - k2 - 1 + q2 : 0;
                if (c1 < 0 || c1 >= nc_[1] || c2 < 0 || c2 >= nc_[2]) continue;
                const double dz = ND > 2 ? gap(2, c2) * a.h[2] : 0.0, dy = ND > 1 ? gap(1, c1) * a.h[1] : 0.0;
                const double rem = bound - dz * dz - dy * dy;
                if (rem < 0.0) continue;
                // the x-range of the row that can meet the ball, rounded outwards (single precision is enough: every
                // cell of the range is tested against `rem` exactly below)
                const int k0 = (int)(__fsqrt_rn((float)rem) * inv_h0 * 1.0001f) + 1;
                int lo = I[0] - k0 - 1, hi = I[0] + k0;
                lo = lo < 0 ? 0 : lo; hi = hi >= nc_[0] ? nc_[0] - 1 : hi;
                const long long row = bits_row(a, c1, c2);
                // the cells around the estimate have been measured above: not again
                const bool seen_row = (ND < 2 || (c1 >= E[1] - 1 && c1 <= E[1] + 1)) && (ND < 3 || (c2 >= E[2] - 1 && c2 <= E[2] + 1));
                for (int w0 = lo >> 6; w0 <= (hi >> 6); ++w0) {
                    unsigned long long m = bits[row + w0];
                    if (w0 == (lo >> 6)) m &= ~0ull << (lo & 63);
                    if (w0 == (hi >> 6)) m &= ~0ull >> (63 - (hi & 63));
                    while (m) {
                        const int b = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int c0 = w0 * 64 + b;
                        if (seen_row && c0 >= E[0] - 1 && c0 <= E[0] + 1) continue;
                        const double dx = gap(0, c0) * a.h[0];
                        if (dx * dx > rem) continue;
                        const unsigned slot = atomicAdd(&qn[g], 1u);
                        if (slot < QCAP) qcell[g][slot] = (long long)c0 | ((long long)c1 << 21) | ((long long)c2 << 42);   // 21 bits per index
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned nq = *(volatile unsigned*)&qn[g];
            if (nq > QCAP) {                       // more occupied cells than the queue holds: the one-lane kernel takes it
                if (gl == 0) seeds[NSEED * w] = -2;
                continue;
            }
            for (unsigned i = gl; i < nq; i += GRP) {
                const long long c = *(volatile long long*)&qcell[g][i];
                scan_cell((int)(c & 0x1fffff), (int)((c >> 21) & 0x1fffff), (int)((c >> 42) & 0x1fffff));
            }
        }
        // the exact nearest sample; seeds[1] = -4: the further near samples have not been collected — the closest-point kernel asks the
        // shell search for them only where the solve from the first seed fails (src/sdf.jl:113-131: nn first, knn on the rare failure)
        {
            const double m = group_min(bd);
            long long out = -1;
            if (m < __builtin_inf()) {
                const unsigned long long bal = __ballot(bd == m);
                const int owner = gbase + __ffsll((long long)((bal >> gbase) & ((1ull << GRP) - 1ull))) - 1;
                const unsigned lo = (unsigned)__shfl((int)(unsigned)(unsigned long long)bslot, owner, 64);
                const unsigned hi = (unsigned)__shfl((int)(unsigned)((unsigned long long)bslot >> 32), owner, 64);
                out = (long long)(((unsigned long long)hi << 32) | lo);
            }
            if (gl == 0) { seeds[NSEED * w] = out; seeds[NSEED * w + 1] = -4; }
        }
    }
}

// (a) nearest samples of every active node -> seeds[NSEED * w .. ]   (latency-bound: keep it light on registers)
template <int ND>
__global__ void __launch_bounds__(256) reinit_search_kernel(ReinitArgs a, const int* cand_id, int S, const double* pts, const unsigned char* cnt,
                                                            const unsigned long long* sup, const unsigned long long* bits,
                                                            const long long* node_list, DevCount nlist, long long* seeds) {
    // one lane per node, for the nodes the group kernel flagged (seeds[0] == -2)
    const long long total = node_list ? count_of(nlist) : (long long)a.n[0] * a.n[1] * a.n[2];
    double hmin = a.h[0], hmax = a.h[0];
    for (int d = 1; d < ND; ++d) { hmin = a.h[d] < hmin ? a.h[d] : hmin; hmax = a.h[d] > hmax ? a.h[d] : hmax; }
    const int nc_[3] = {a.n[0] - 1, ND > 1 ? a.n[1] - 1 : 1, ND > 2 ? a.n[2] - 1 : 1};      // cells per dimension
    const int nb_[3] = {(nc_[0] + RB - 1) / RB, (nc_[1] + RB - 1) / RB, (nc_[2] + RB - 1) / RB};  // blocks per dimension
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        if (seeds[NSEED * w] != -2) continue;
        const long long t = node_list ? node_list[w] : w;
        const int I[3] = {(int)(t % a.n[0]), (int)((t / a.n[0]) % a.n[1]), (int)(t / ((long long)a.n[0] * a.n[1]))};
        double xq[3] = {0, 0, 0};
        for (int d = 0; d < ND; ++d) xq[d] = a.lc[d] + (double)(I[d] + a.goff[d]) * a.h[d];
        // the NSEED nearest samples, nearest first
        double bd[NSEED];
        long long bslot[NSEED];
        for (int k = 0; k < NSEED; ++k) { bd[k] = __builtin_inf(); bslot[k] = -1; }
        // squared distance from the node to the box of cells [c, c + w) per dimension (0 inside)
        auto box_d2 = [&](int c0, int c1, int c2, int w) {
            const int c[3] = {c0, c1, c2};
            double d2 = 0.0;
            for (int d = 0; d < ND; ++d) {
                const int gap = I[d] < c[d] ? c[d] - I[d] : (I[d] > c[d] + w ? I[d] - (c[d] + w) : 0);
                const double e = (double)gap * a.h[d];
                d2 += e * e;
            }
            return d2;
        };
        auto scan_cell = [&](int c0, int c1, int c2) {
            if (c0 < 0 || c0 >= nc_[0] || c1 < 0 || c1 >= nc_[1] || c2 < 0 || c2 >= nc_[2]) return;
            if (box_d2(c0, c1, c2, 1) > bd[0]) return;       // cannot hold a nearer sample than the best so far
            const int J[3] = {c0, c1, c2};
            const int id = cand_id[cell_lin(a, J)];
            if (id < 0) return;
            const int m = cnt[id];
            for (int k = 0; k < m; ++k) {
                const long long slot = (long long)id * S + k;
                double d2 = 0.0;
                for (int d = 0; d < ND; ++d) { const double e = pts[3 * slot + d] - xq[d]; d2 += e * e; }
                if (d2 < bd[NSEED - 1]) {                 // sorted insertion, once per sample
                    bool dup = false;
                    for (int p = 0; p < NSEED; ++p) dup = dup || bslot[p] == slot;
                    if (dup) continue;
                    int p = NSEED - 1;
                    while (p > 0 && bd[p - 1] > d2) { bd[p] = bd[p - 1]; bslot[p] = bslot[p - 1]; --p; }
                    bd[p] = d2; bslot[p] = slot;
                }
            }
        };
        // the cells c0 in [lo, hi] of row (c1, c2) that hold samples, straight from the occupancy bits (one or two word
        // loads instead of a probe per cell)
        auto scan_row = [&](int lo, int hi, int c1, int c2) {
            if (c1 < 0 || c1 >= nc_[1] || c2 < 0 || c2 >= nc_[2]) return;
            lo = lo < 0 ? 0 : lo; hi = hi >= nc_[0] ? nc_[0] - 1 : hi;
            const long long row = bits_row(a, c1, c2);
            for (int w0 = lo >> 6; w0 <= (hi >> 6); ++w0) {
                unsigned long long m = bits[row + w0];
                if (w0 == (lo >> 6)) m &= ~0ull << (lo & 63);
                if (w0 == (hi >> 6)) m &= ~0ull >> (63 - (hi & 63));
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    scan_cell(w0 * 64 + b, c1, c2);
                }
            }
        };
        bool done = false;
        // guided search: the first-order closest-point estimate x - ϕ∇ϕ/|∇ϕ|² (centred differences) lands next to the
        // nearest samples when ϕ is anywhere near a distance function; the cells around it give a tight upper bound,
        // and scanning every cell that meets the ball of that radius around the node then makes the result exact.
        {
            double xe[3];
            const bool have = first_order_foot<ND>(a, xq, hmin, xe);
            if (have) {
                int E[3];
                cell_of(a, xe, E);
                for (int c2 = E[2] - (ND > 2 ? 1 : 0); c2 <= E[2] + (ND > 2 ? 1 : 0); ++c2)
                    for (int c1 = E[1] - (ND > 1 ? 1 : 0); c1 <= E[1] + (ND > 1 ? 1 : 0); ++c1)
                        scan_row(E[0] - 1, E[0] + 1, c1, c2);
            }
            const double R0 = bslot[0] >= 0 ? sqrt(bd[0]) : __builtin_inf();
            if (R0 <= 10.0 * hmin) {
                // gap (in cells) between the node and cell c of dimension d
                auto gap = [&](int d, int c) { return c > I[d] ? c - I[d] : (c + 1 < I[d] ? I[d] - (c + 1) : 0); };
                const int k2 = ND > 2 ? (int)(R0 / a.h[2]) + 1 : 0, k1 = ND > 1 ? (int)(R0 / a.h[1]) + 1 : 0;
                for (int c2 = I[2] - k2 - (ND > 2 ? 1 : 0); c2 <= I[2] + k2; ++c2) {
                    const double dz = ND > 2 ? gap(2, c2) * a.h[2] : 0.0;
                    if (dz * dz > bd[0]) continue;
                    for (int c1 = I[1] - k1 - (ND > 1 ? 1 : 0); c1 <= I[1] + k1; ++c1) {
                        const double dy = ND > 1 ? gap(1, c1) * a.h[1] : 0.0;
                        const double rem = bd[0] - dz * dz - dy * dy;
                        if (rem < 0.0) continue;
                        const int k0 = (int)(sqrt(rem) / a.h[0]) + 1;
                        scan_row(I[0] - k0 - 1, I[0] + k0, c1, c2);
                    }
                }
                done = true;
            }
        }
        // otherwise: cells within Chebyshev radius s of the node, shell by shell; every unseen sample is then farther
        // than s*hmin
        for (int s = 1; s <= FINE_SHELLS && !done && bslot[0] < 0; ++s) {      // (with a sample in hand the ball scan below is cheaper)
            const int lo[3] = {I[0] - s, ND > 1 ? I[1] - s : 0, ND > 2 ? I[2] - s : 0};
            const int hi[3] = {I[0] + s - 1, ND > 1 ? I[1] + s - 1 : 0, ND > 2 ? I[2] + s - 1 : 0};
            for (int c2 = lo[2]; c2 <= hi[2]; ++c2)
                for (int c1 = lo[1]; c1 <= hi[1]; ++c1) {
                    const bool edge12 = (ND > 2 && (c2 == lo[2] || c2 == hi[2])) || (ND > 1 && (c1 == lo[1] || c1 == hi[1]));
                    for (int c0 = lo[0]; c0 <= hi[0]; c0 += (edge12 || hi[0] == lo[0]) ? 1 : (hi[0] - lo[0])) scan_cell(c0, c1, c2);   // shell cells only
                }
            done = bslot[0] >= 0 && sqrt(bd[0]) <= (double)s * hmin;
        }
        // far field: two levels of occupancy above the cells — blocks of RB^N cells, and per super-block of 8^N blocks the mask of its
        // occupied blocks (`sup`, 8 words).  With nothing found yet the nearest occupied super-block is scanned first; then every
        // super-block, and in it every occupied block, that meets the ball of the best distance so far.
        if (!done) {
            const int ns_[3] = {(nb_[0] + 7) / 8, (nb_[1] + 7) / 8, (nb_[2] + 7) / 8};
            constexpr int SW = RB * 8;                     // cells per super-block and dimension
            constexpr int NWORD = ND == 3 ? 8 : 1;
            auto scan_super = [&](int s0, int s1, int s2) {
                const long long sl = s0 + (long long)ns_[0] * (s1 + (long long)ns_[1] * s2);
                for (int wd = 0; wd < NWORD; ++wd) {
                    unsigned long long m = sup[8 * sl + wd];
                    while (m) {
                        const int l = wd * 64 + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int b0 = s0 * 8 + (l & 7), b1 = ND > 1 ? s1 * 8 + ((l >> 3) & 7) : 0, b2 = ND > 2 ? s2 * 8 + (l >> 6) : 0;
                        if (box_d2(b0 * RB, b1 * RB, b2 * RB, RB) > bd[0]) continue;
                        for (int c2 = b2 * RB; c2 < (ND > 2 ? (b2 + 1) * RB : 1); ++c2)
                            for (int c1 = b1 * RB; c1 < (ND > 1 ? (b1 + 1) * RB : 1); ++c1)
                                scan_row(b0 * RB, (b0 + 1) * RB - 1, c1, c2);
                    }
                }
            };
            long long first = -1;
            if (bslot[0] < 0) {
                double best = __builtin_inf();
                int f[3] = {0, 0, 0};
                for (int s2 = 0; s2 < ns_[2]; ++s2)
                    for (int s1 = 0; s1 < ns_[1]; ++s1)
                        for (int s0 = 0; s0 < ns_[0]; ++s0) {
                            const long long sl = s0 + (long long)ns_[0] * (s1 + (long long)ns_[1] * s2);
                            unsigned long long any = 0;
                            for (int wd = 0; wd < NWORD; ++wd) any |= sup[8 * sl + wd];
                            if (!any) continue;
                            const double d2 = box_d2(s0 * SW, s1 * SW, s2 * SW, SW);
                            if (d2 < best) { best = d2; first = sl; f[0] = s0; f[1] = s1; f[2] = s2; }
                        }
                if (first >= 0) scan_super(f[0], f[1], f[2]);
            }
            if (bslot[0] >= 0) {
                const double R = sqrt(bd[0]);
                int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                for (int d = 0; d < ND; ++d) {
                    const double kk = R / a.h[d];
                    const int k = kk < (double)nc_[d] ? (int)kk : nc_[d];      // cells whose gap to the node is at most floor(R / h)
                    int l = I[d] - k - 1, u = I[d] + k;
                    l = l < 0 ? 0 : l; u = u > nc_[d] - 1 ? nc_[d] - 1 : u;
                    lo[d] = l / SW; hi[d] = u / SW;
                }
                for (int s2 = lo[2]; s2 <= hi[2]; ++s2)
                    for (int s1 = lo[1]; s1 <= hi[1]; ++s1)
                        for (int s0 = lo[0]; s0 <= hi[0]; ++s0) {
                            if (s0 + (long long)ns_[0] * (s1 + (long long)ns_[1] * s2) == first) continue;
                            if (box_d2(s0 * SW, s1 * SW, s2 * SW, SW) > bd[0]) continue;
                            scan_super(s0, s1, s2);
                        }
            }
        }
        for (int k = 0; k < NSEED; ++k) seeds[NSEED * w + k] = bslot[k];
    }
}

// (b) closest point from the seeds, signed distance.  Pass 0 runs over every node: a node that comes with its nearest sample only
// (seeds[1] == -4) and whose solve from it does not converge is appended to `retry` and marked for the shell search (seeds[0] = -2),
// which collects its NSEED nearest samples; pass 1 runs over the `retry` list with them.
template <int NV, int ND>
__global__ void __launch_bounds__(128) reinit_newton_kernel(ReinitArgs a, int S, const double* pts, const long long* node_list, DevCount nlist,
                                                            long long* seeds, void* out, unsigned* nfail, unsigned* nfar, unsigned* retry,
                                                            unsigned* retry_count, int pass) {
    const long long total = pass ? (long long)*retry_count : (node_list ? count_of(nlist) : (long long)a.n[0] * a.n[1] * a.n[2]);
    double hmax = a.h[0];
    for (int d = 1; d < a.ndim; ++d) hmax = a.h[d] > hmax ? a.h[d] : hmax;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long w = pass ? (long long)retry[i] : i;
        const long long t = node_list ? node_list[w] : w;
        const int I[3] = {(int)(t % a.n[0]), (int)((t / a.n[0]) % a.n[1]), (int)(t / ((long long)a.n[0] * a.n[1]))};
        const long long q = a.origin + I[0] + I[1] * a.s1 + I[2] * a.s2;
        double xq[3] = {0, 0, 0};
        for (int d = 0; d < a.ndim; ++d) xq[d] = a.lc[d] + (double)(I[d] + a.goff[d]) * a.h[d];
        long long* bslot = seeds + NSEED * w;
        const bool first_only = pass == 0 && bslot[0] >= 0 && bslot[1] == -4;
        double cp[3] = {xq[0], xq[1], xq[2]};
        bool conv = false;
        if (bslot[0] < 0) {
            atomicAdd(nfar, 1u);
        } else {
            const double safeguard = 1.5 * hmax;
            double bestcp[3] = {0, 0, 0}, bestd = __builtin_inf();
            for (int k = 0; k < (first_only ? 1 : NSEED) && bslot[k] >= 0 && !conv; ++k) {
                const double seed[3] = {pts[3 * bslot[k]], pts[3 * bslot[k] + 1], pts[3 * bslot[k] + 2]};
                int J[3];
                cell_of<ND>(a, seed, J);
                double c3[3];
                conv = closest_on_patch<NV, ND>(a, J, xq, seed, safeguard, c3);
                double d2 = 0.0;
                for (int d = 0; d < a.ndim; ++d) d2 += (xq[d] - c3[d]) * (xq[d] - c3[d]);
                if (conv || d2 < bestd) { bestd = d2; bestcp[0] = c3[0]; bestcp[1] = c3[1]; bestcp[2] = c3[2]; }
            }
            if (first_only && !conv) {                    // the further near samples are needed: second pass
                retry[atomicAdd(retry_count, 1u)] = (unsigned)w;
                bslot[0] = -2;
                continue;
            }
            cp[0] = bestcp[0]; cp[1] = bestcp[1]; cp[2] = bestcp[2];
            if (!conv) atomicAdd(nfail, 1u);
        }
        double d2 = 0.0;
        for (int d = 0; d < a.ndim; ++d) d2 += (xq[d] - cp[d]) * (xq[d] - cp[d]);
        const double v = ld_val(a.phi, q, a.f32);
        const double sgn = v > 0 ? 1.0 : (v < 0 ? -1.0 : v);
        st_val(out, q, a.f32, bslot[0] < 0 ? v : sgn * sqrt(d2));
    }
}

// copy the new values of the active nodes back (the evaluation phase never writes ϕ: src/reinitializer.jl:21-24)
// — unless one of the call's lists did not fit its buffer (the host repeats the call with larger ones: ϕ must still be what it was)
__global__ void __launch_bounds__(256) reinit_commit_kernel(ReinitArgs a, const void* src, void* dst, const long long* node_list, DevCount nlist,
                                                            DevCount ncand) {
    if ((nlist.p && *nlist.p > nlist.cap) || (ncand.p && *ncand.p > ncand.cap)) return;
    const long long total = node_list ? count_of(nlist) : (long long)a.n[0] * a.n[1] * a.n[2];
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        const long long t = node_list ? node_list[w] : w;
        const long long q = a.origin + (t % a.n[0]) + ((t / a.n[0]) % a.n[1]) * a.s1 + (t / ((long long)a.n[0] * a.n[1])) * a.s2;
        if (a.mask && !a.mask[q]) continue;
        st_val(dst, q, a.f32, ld_val(src, q, a.f32));
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
// _interpolation_matrix (src/interpolation.jl:52-63): pseudo-inverse of the collocation matrix V (nv x nc) by the
// normal equations in long double (V is tiny and well conditioned), and the bound Λ = (max_t Σ_j |L_j(t)|)^N
static bool interp_matrix(int order, int ndim, double M[36], int* nv_out, double* lambda) {
    const int so = order % 2 ? order : order + 1, nc = order + 1, nv = so + 1;
    if (order < 1 || order > 5) return false;
    auto binom = [](int n, int k) { double r = 1; for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i; return r; };
    const long double a = (so - 1) / (2.0L * so), b = (so + 1) / (2.0L * so);
    long double V[6][6], G[6][12];
    for (int i = 0; i < nv; ++i) {
        const long double x = ((long double)i / so - a) / (b - a);
        for (int j = 0; j < nc; ++j) V[i][j] = binom(order, j) * powl(x, j) * powl(1 - x, order - j);
    }
    for (int r = 0; r < nc; ++r) {          // [VᵀV | I]
        for (int c = 0; c < nc; ++c) { long double s = 0; for (int i = 0; i < nv; ++i) s += V[i][r] * V[i][c]; G[r][c] = s; }
        for (int c = 0; c < nc; ++c) G[r][nc + c] = r == c ? 1 : 0;
    }
    for (int k = 0; k < nc; ++k) {          // Gauss–Jordan with partial pivoting
        int p = k;
        for (int r = k + 1; r < nc; ++r) if (fabsl(G[r][k]) > fabsl(G[p][k])) p = r;
        if (G[p][k] == 0) return false;
        for (int c = 0; c < 2 * nc; ++c) { long double t = G[k][c]; G[k][c] = G[p][c]; G[p][c] = t; }
        const long double piv = G[k][k];
        for (int c = 0; c < 2 * nc; ++c) G[k][c] /= piv;
        for (int r = 0; r < nc; ++r) {
            if (r == k) continue;
            const long double f = G[r][k];
            for (int c = 0; c < 2 * nc; ++c) G[r][c] -= f * G[k][c];
        }
    }
    for (int i = 0; i < nc; ++i)            // M = (VᵀV)⁻¹ Vᵀ
        for (int j = 0; j < nv; ++j) { long double s = 0; for (int c = 0; c < nc; ++c) s += G[i][nc + c] * V[j][c]; M[i * nv + j] = (double)s; }
    double lam1 = 0;
    for (int k = 0; k <= 400; ++k) {
        const double t = k / 400.0;
        double sum = 0;
        for (int j = 0; j < nv; ++j) {
            double l = 0;
            for (int i = 0; i < nc; ++i) l += M[i * nv + j] * binom(order, i) * pow(t, i) * pow(1 - t, order - i);
            sum += fabs(l);
        }
        lam1 = sum > lam1 ? sum : lam1;
    }
    *lambda = pow(lam1 * 1.01, ndim);
    *nv_out = nv;
    return true;
}

// ---- InterpolatedField(ϕ, order)(x): value, gradient and Hessian of the piecewise interpolant at arbitrary points
// (src/interpolation.jl:117-151,228-260): the patch of the cell that holds x (compute_index, clamped to the grid).
// pts: npts x ndim doubles (point-major); val: npts; grad: npts x ndim; hess: npts x ndim x ndim (row-major, symmetric);
// grad / hess may be NULL.
template <int NV>
__global__ void __launch_bounds__(128) interp_points_kernel(ReinitArgs a, long long npts, const double* pts, double* val, double* grad, double* hess) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npts; p += (long long)gridDim.x * blockDim.x) {
        double x[3] = {0, 0, 0};
        for (int d = 0; d < a.ndim; ++d) x[d] = pts[p * a.ndim + d];
        int I[3];
        cell_of(a, x, I);
        double v, g[3], H[6];
        patch_eval<NV>(a, I, x, hess != nullptr, v, g, H);
        val[p] = v;
        if (grad) for (int d = 0; d < a.ndim; ++d) grad[p * a.ndim + d] = g[d];
        if (hess) {
            const int N = a.ndim;
            const double full[3][3] = {{H[0], H[3], H[4]}, {H[3], H[1], H[5]}, {H[4], H[5], H[2]}};
            for (int r = 0; r < N; ++r)
                for (int c = 0; c < N; ++c) hess[(p * N + r) * N + c] = full[r][c];
        }
    }
}
static bool interp_matrix(int order, int ndim, double M[36], int* nv_out, double* lambda);
int interp_run(int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, const double lc[3], const double h[3], int order,
               const void* phi, int f32, long long npts, const double* pts, double* val, double* grad, double* hess, hipStream_t stream, const char** err) {
    ReinitArgs a;
    a.ndim = ndim;
    for (int d = 0; d < 3; ++d) { a.n[d] = n[d]; a.goff[d] = goff[d]; a.lc[d] = lc[d]; a.h[d] = h[d]; }
    a.s1 = s1; a.s2 = s2; a.origin = origin;
    a.order = order; a.upsample = 1; a.maxiters = 1; a.xtol = a.ftol = 0.0;
    a.phi = phi; a.f32 = f32; a.mask = nullptr;
    if (!interp_matrix(order, ndim, a.M, &a.nv, &a.lambda)) { *err = "InterpolatedField: order must be in 1..5"; return 1; }
    a.off = -((a.nv - 1) - 1) / 2;
    if (a.nv + a.off - 1 > LSM_GHOST + 1 || -a.off > LSM_GHOST) { *err = "InterpolatedField: stencil exceeds the ghost layers"; return 1; }
    for (int d = 0; d < ndim; ++d)
        if (n[d] < 2) { *err = "InterpolatedField: at least two nodes per dimension"; return 1; }
    if (npts <= 0) return 0;
    const unsigned gb = (unsigned)((npts + 127) / 128 > 65535 ? 65535 : (npts + 127) / 128);
    if (a.nv == 2) hipLaunchKernelGGL(interp_points_kernel<2>, dim3(gb), dim3(128), 0, stream, a, npts, pts, val, grad, hess);
    else if (a.nv == 4) hipLaunchKernelGGL(interp_points_kernel<4>, dim3(gb), dim3(128), 0, stream, a, npts, pts, val, grad, hess);
    else hipLaunchKernelGGL(interp_points_kernel<6>, dim3(gb), dim3(128), 0, stream, a, npts, pts, val, grad, hess);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---- the interface samples of a field and the structures that index them (candidate cells, samples per cell, occupancy
// bits and blocks): built once per reinitialize! call, or kept in a NewtonSDF object for point queries
struct SampleSet {
    ReinitArgs a;
    int S = 0;                 // start points per cell
    unsigned ncand = 0;        // candidate cells
    unsigned cap_nodes_e = 0, cap_cand_e = 0;   // entries the node list / the per-candidate buffers were launched with (launch_samples)
    DevCount dn{nullptr, 0, 0}, dc{nullptr, 0, 0};   // the two lengths as the kernels of this round see them
    long long nwork = 0;       // active nodes (band) or all nodes
    int* cand_id = nullptr;
    long long *cand_cell = nullptr, *maybe = nullptr, *node_list = nullptr;
    unsigned* counters = nullptr;      // [0] maybe cells, [1] nfail, [2] nfar, [3] band nodes, [4] start points, [5] candidate cells, [6] retried nodes
    double* pts = nullptr;
    unsigned char *valid = nullptr, *cnt = nullptr, *blk = nullptr;
    unsigned long long* sup = nullptr;      // occupied blocks per super-block (reinit_sup_kernel)
    unsigned long long* bits = nullptr;
    unsigned long long* starts = nullptr;   // start points to project: candidate id | slot << 32
    // bytes allocated behind each pointer (grow(): a buffer is re-allocated only when a call needs more — the set of a NewtonSDF
    // object is built once; the workspace of reinitialize! lives on the handle and stops allocating after the first calls)
    size_t cap_cand_id = 0, cap_cand_cell = 0, cap_maybe = 0, cap_node_list = 0, cap_counters = 0, cap_pts = 0, cap_valid = 0, cap_cnt = 0, cap_blk = 0,
           cap_bits = 0, cap_starts = 0, cap_sup = 0;
    // workspace only: cand_id == -1, bits == 0 and blk == 0 everywhere for a grid of `clean_cells` cells — the state every band call
    // starts from, and restores by un-marking its own candidate cells (reinit_unmark_kernel) instead of clearing arrays of the size of the grid
    long long clean_cells = -1;
    void release() {
        (void)hipFree(cand_id); (void)hipFree(cand_cell); (void)hipFree(maybe); (void)hipFree(node_list); (void)hipFree(counters);
        (void)hipFree(pts); (void)hipFree(valid); (void)hipFree(cnt); (void)hipFree(blk); (void)hipFree(sup); (void)hipFree(bits); (void)hipFree(starts);
        cand_id = nullptr; cand_cell = maybe = node_list = nullptr; counters = nullptr; pts = nullptr; valid = cnt = blk = nullptr; sup = nullptr; bits = nullptr; starts = nullptr;
        cap_cand_id = cap_cand_cell = cap_maybe = cap_node_list = cap_counters = cap_pts = cap_valid = cap_cnt = cap_blk = cap_bits = cap_starts = cap_sup = 0;
        cap_nodes_e = cap_cand_e = 0;
        clean_cells = -1;
    }
};
template <class T>
static hipError_t grow(T*& p, size_t& cap, size_t bytes, bool* fresh = nullptr) {
    if (fresh) *fresh = false;
    if (p && cap >= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 256;          // headroom: the band's size drifts from call to call
    const hipError_t e = hipMalloc((void**)&p, want);
    if (e == hipSuccess) { cap = want; if (fresh) *fresh = true; }
    return e;
}
// the device buffers of reinitialize! kept between calls on one handle (LsmHandle::reinit_ws): eleven hipMalloc / hipFree pairs per
// call were more than a third of a band call's 3.8 ms at 256³
struct ReinitWorkspace {
    SampleSet ss;
    long long* seeds = nullptr;
    unsigned* retry = nullptr;         // nodes whose solve from the nearest sample failed (second pass of the closest-point kernel)
    long long* foot = nullptr;         // cell of the first-order closest-point estimate per node
    size_t cap_seeds = 0, cap_retry = 0, cap_foot = 0;
    unsigned last_nodes = 0, last_cand = 0;      // band nodes and candidate cells of the previous call: what the next one is launched for
};
void reinit_workspace_free(ReinitWorkspace* w) {
    if (!w) return;
    w->ss.release();
    (void)hipFree(w->seeds);
    (void)hipFree(w->retry);
    (void)hipFree(w->foot);
    delete w;
}
// the second level of the far-field search: per super-block of 8^N blocks the mask of its occupied blocks, 8 words each (bit
// b0 + 8·(b1 + 8·b2) of them), written whole from the block bytes
__global__ void __launch_bounds__(256) reinit_sup_kernel(ReinitArgs a, const unsigned char* blk, unsigned long long* sup, long long nwords) {
    const int nb_[3] = {(a.n[0] - 1 + RB - 1) / RB, a.ndim > 1 ? (a.n[1] - 1 + RB - 1) / RB : 1, a.ndim > 2 ? (a.n[2] - 1 + RB - 1) / RB : 1};
    const long long s0n = (nb_[0] + 7) / 8, s1n = (nb_[1] + 7) / 8;
    const int b = threadIdx.x & 63;                      // a wave per word, a lane per block
    for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwords; w += ((long long)gridDim.x * blockDim.x) >> 6) {
        const long long sl = w >> 3;
        const int l = (int)(w & 7) * 64 + b;
        const int s0 = (int)(sl % s0n), s1 = (int)((sl / s0n) % s1n), s2 = (int)(sl / (s0n * s1n));
        const int B[3] = {s0 * 8 + (l & 7), s1 * 8 + ((l >> 3) & 7), s2 * 8 + (l >> 6)};
        const unsigned long long m = __ballot(B[0] < nb_[0] && B[1] < nb_[1] && B[2] < nb_[2] && blk[blk_lin(a, B)] != 0);
        if (b == 0) sup[w] = m;
    }
}
// band calls on a workspace: back to cand_id == -1, bits == 0, blk == 0 by visiting the candidate cells of the call that ends
__global__ void __launch_bounds__(256) reinit_unmark_kernel(ReinitArgs a, const long long* cand_cell, DevCount ncand_, int* cand_id, unsigned char* blk,
                                                            unsigned long long* bits) {
    const unsigned ncand = (unsigned)count_of(ncand_);
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < ncand; id += gridDim.x * blockDim.x) {
        const long long c = cand_cell[id];
        cand_id[c] = -1;
        int I[3];
        cell_unlin(a, c, I);
        const int B[3] = {I[0] / RB, I[1] / RB, I[2] / RB};
        blk[blk_lin(a, B)] = 0;
        bits[bits_row(a, I[1], I[2]) + I[0] / 64] = 0ull;        // every set bit of the word belongs to a candidate cell of this call
    }
}

static int setup_args(ReinitArgs& a, int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, const double lc[3],
                      const double h[3], int order, int upsample, int maxiters, double xtol, double ftol, const void* phi, int f32,
                      const unsigned char* mask, const char** err) {
    a.ndim = ndim;
    for (int d = 0; d < 3; ++d) { a.n[d] = n[d]; a.goff[d] = goff[d]; a.lc[d] = lc[d]; a.h[d] = h[d]; }
    a.s1 = s1; a.s2 = s2; a.origin = origin;
    a.order = order; a.upsample = upsample; a.maxiters = maxiters; a.xtol = xtol; a.ftol = ftol;
    a.phi = phi; a.f32 = f32; a.mask = mask;
    if (!interp_matrix(order, ndim, a.M, &a.nv, &a.lambda)) { *err = "reinitialize: order must be in 1..5"; return 1; }
    a.off = -((a.nv - 1) - 1) / 2;
    if (a.nv + a.off - 1 > LSM_GHOST + 1 || -a.off > LSM_GHOST) { *err = "reinitialize: stencil exceeds the ghost layers"; return 1; }
    return 0;
}

#define RE_HIP(call) do { if ((call) != hipSuccess) { *err = #call; ss.release(); return 2; } } while (0)
// Candidate cells -> interface samples -> per-cell counts, occupancy bits, blocks and super-blocks (steps 1 and 2 above) — launched
// without waiting for anything: the number of band nodes and of candidate cells stay on the device (DevCount), the buffers are sized
// for `want_nodes` / `want_cand` entries (what the previous call needed, with headroom) and the caller checks afterwards whether
// both fitted (samples_fit) — if not it launches again with what it has learnt.  `count_only`: stop behind the candidate-cell test
// (a first call: nothing is known yet, the rest of the round would work on a handful of cells).  ss.a is set.
// `keep_clean`: ss is a workspace whose cand_id / bits / blk are restored by the caller after the call (band fields only).
static int launch_samples(SampleSet& ss, long long total, hipStream_t stream, const char** err, bool keep_clean, unsigned want_nodes, unsigned want_cand,
                          bool count_only = false) {
    const ReinitArgs& a = ss.a;
    const int ndim = a.ndim;
    const int* n = a.n;
    long long nc = 1;
    for (int d = 0; d < ndim; ++d) nc *= n[d] - 1;
    ss.S = 1;
    for (int d = 0; d < ndim; ++d) ss.S *= a.upsample + 1;
    const int S = ss.S;
    bool fresh_id = false, fresh_blk = false, fresh_bits = false;
    RE_HIP(grow(ss.cand_id, ss.cap_cand_id, sizeof(int) * (size_t)nc, &fresh_id));
    RE_HIP(grow(ss.counters, ss.cap_counters, 8 * sizeof(unsigned)));
    RE_HIP(hipMemsetAsync(ss.counters, 0, 8 * sizeof(unsigned), stream));
    size_t nblk = 1, nsupw = 8;                          // block bytes; words of `sup` (8 per super-block of 8^N blocks)
    for (int d = 0; d < ndim; ++d) { nblk *= (size_t)((n[d] - 1 + RB - 1) / RB); nsupw *= (size_t)(((n[d] - 1 + RB - 1) / RB + 7) / 8); }
    size_t nwords = (size_t)((n[0] - 1 + 63) / 64);
    for (int d = 1; d < ndim; ++d) nwords *= (size_t)(n[d] - 1);
    RE_HIP(grow(ss.blk, ss.cap_blk, nblk, &fresh_blk));
    RE_HIP(grow(ss.sup, ss.cap_sup, sizeof(unsigned long long) * nsupw));
    RE_HIP(grow(ss.bits, ss.cap_bits, sizeof(unsigned long long) * nwords, &fresh_bits));
    const bool clean = keep_clean && a.mask && ss.clean_cells == nc && !fresh_id && !fresh_blk && !fresh_bits;
    ss.clean_cells = -1;                       // until the caller has restored the state at the end of the call
    // band fields: the compact list of the active nodes first — candidate cells, the distance computation and the
    // commit all run over it (the band is ~1 % of a 3-D grid)
    const long long nodes = (long long)n[0] * n[1] * n[2];
    if (nodes > 0xffffffffll) { *err = "reinitialize: more than 2^32 nodes"; ss.release(); return 1; }
    if (a.mask) {
        RE_HIP(grow(ss.node_list, ss.cap_node_list, sizeof(long long) * (size_t)(want_nodes ? want_nodes : 1)));
        ss.cap_nodes_e = (unsigned)std::min<size_t>(ss.cap_node_list / sizeof(long long), 0xffffffffu);
        const long long nvec = (total + 15) / 16;
        const unsigned gl = (unsigned)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
        hipLaunchKernelGGL(reinit_band_nodes_kernel, dim3(gl), dim3(256), 0, stream, a, total, ss.node_list, (long long)ss.cap_nodes_e, ss.counters + 3);
        if (!clean) RE_HIP(hipMemsetAsync(ss.cand_id, 0xFF, sizeof(int) * (size_t)nc, stream));     // -1 everywhere; the cells kernel visits the band only
        ss.dn = DevCount{ss.counters + 3, ss.cap_nodes_e, 0};
    } else {
        ss.cap_nodes_e = (unsigned)nodes;
        ss.dn = DevCount{nullptr, 0, nodes};
    }
    // the cells looked at: those whose lowest corner is a band node, or all of them
    const long long ncell_work = a.mask ? (long long)ss.cap_nodes_e : nc, ncell_grid = a.mask ? (long long)(want_nodes ? want_nodes : 1) : nc;
    const unsigned gb = (unsigned)((ncell_grid + 255) / 256 > 65535 ? 65535 : (ncell_grid + 255) / 256);
    RE_HIP(grow(ss.maybe, ss.cap_maybe, sizeof(long long) * (size_t)(ncell_work ? ncell_work : 1)));
    hipLaunchKernelGGL(reinit_cells_kernel, dim3(gb), dim3(256), 0, stream, a, ss.cand_id, ss.maybe, ss.counters, ss.node_list, ss.dn);
    // samples: S slots per candidate cell, for at most cap_cand_e of them
    const size_t wantc = want_cand ? want_cand : 1;
    RE_HIP(grow(ss.pts, ss.cap_pts, sizeof(double) * 3 * wantc * S));
    RE_HIP(grow(ss.valid, ss.cap_valid, wantc * S));
    RE_HIP(grow(ss.cnt, ss.cap_cnt, wantc));
    RE_HIP(grow(ss.starts, ss.cap_starts, sizeof(unsigned long long) * wantc * S));
    const size_t capc = std::min(std::min(ss.cap_pts / (sizeof(double) * 3 * S), ss.cap_valid / S), std::min(ss.cap_cnt, ss.cap_starts / (sizeof(unsigned long long) * S)));
    ss.cap_cand_e = want_cand ? (unsigned)std::min<size_t>(capc, 0xffffffffu) : 0u;
    ss.dc = DevCount{ss.counters + 5, ss.cap_cand_e, 0};
    // the list of candidate cells (as long as the "maybe" list could be)
    RE_HIP(grow(ss.cand_cell, ss.cap_cand_cell, sizeof(long long) * (size_t)(ncell_work ? ncell_work : 1)));
    {
        const dim3 g2((unsigned)((ncell_grid / 2 + 256) / 256 > 65535 ? 65535 : (ncell_grid / 2 + 256) / 256)), b2(256);
#define LSM_CELLS2(NV_, NC_, ND_) hipLaunchKernelGGL((reinit_cells2_kernel<NV_, NC_, ND_>), g2, b2, 0, stream, a, ss.maybe, ss.counters, ss.cand_id, ss.cand_cell, ss.counters + 5, ss.cap_cand_e)
        const int ncf = a.order + 1;
        if (a.nv == 4 && ncf == 4 && ndim == 3) LSM_CELLS2(4, 4, 3);
        else if (a.nv == 4 && ncf == 4 && ndim == 2) LSM_CELLS2(4, 4, 2);
        else if (a.nv == 4 && ncf == 3 && ndim == 3) LSM_CELLS2(4, 3, 3);
        else if (a.nv == 4 && ncf == 3 && ndim == 2) LSM_CELLS2(4, 3, 2);
        else if (a.nv == 2 && ncf == 2 && ndim == 3) LSM_CELLS2(2, 2, 3);
        else if (a.nv == 2 && ncf == 2 && ndim == 2) LSM_CELLS2(2, 2, 2);
        else LSM_CELLS2(0, 0, 0);
#undef LSM_CELLS2
    }
    if (count_only) return 0;          // the caller wants the two counts first (nothing is known about this field yet)
    if (ss.cap_cand_e) RE_HIP(hipMemsetAsync(ss.valid, 0, (size_t)ss.cap_cand_e * S, stream));
    if (!clean) {
        RE_HIP(hipMemsetAsync(ss.blk, 0, nblk, stream));
        RE_HIP(hipMemsetAsync(ss.bits, 0, sizeof(unsigned long long) * nwords, stream));
    }
    if (ss.cap_cand_e) {
        const long long work = (long long)want_cand * S;
        const unsigned gw = (unsigned)((work + 255) / 256 > 262144 ? 262144 : (work + 255) / 256);
        hipLaunchKernelGGL(reinit_starts_kernel, dim3((gw + 3) / 4), dim3(256), 0, stream, a, ss.cand_cell, ss.cand_id, ss.dc, S, ss.starts, ss.counters + 4);
        // the projection kernel walks the registered points with a stride (their number stays on the device: about a third of `work` in 3-D)
        const unsigned gs = gw / 2 + 1;
#define LSM_SAMPLE(NV_, ND_) hipLaunchKernelGGL((reinit_sample_kernel<NV_, ND_>), dim3(gs), dim3(256), 0, stream, a, ss.cand_cell, ss.cand_id, S, ss.starts, \
                                                ss.counters + 4, ss.pts, ss.valid)
        // (the dimension stays a run-time value here: the version with it fixed needs 288 registers in 3-D — one wave per SIMD — and
        // is slower capped at 256, 1.74 against 1.65 ms for the 256³ band call)
        if (a.nv == 2) LSM_SAMPLE(2, 0);
        else if (a.nv == 4) LSM_SAMPLE(4, 0);
        else LSM_SAMPLE(6, 0);
#undef LSM_SAMPLE
        hipLaunchKernelGGL(reinit_compact_kernel, dim3(S <= 32 ? (want_cand + 7) / 8 : (want_cand + 255) / 256), dim3(256), 0, stream, a, ss.cand_cell, ss.dc, S, ss.pts,
                           ss.valid, ss.cnt, ss.blk, ss.bits);
    }
    hipLaunchKernelGGL(reinit_sup_kernel, dim3((unsigned)((nsupw + 3) / 4 > 65535 ? 65535 : (nsupw + 3) / 4)), dim3(256), 0, stream, a, ss.blk, ss.sup, (long long)nsupw);
    return 0;
}
// after the stream has been synchronised: did the band nodes and the candidate cells fit the buffers they were launched with?  Sets
// ss.nwork / ss.ncand to the true counts either way.
static bool samples_fit(SampleSet& ss, const unsigned cn[8]) {
    const long long nodes = (long long)ss.a.n[0] * ss.a.n[1] * ss.a.n[2];
    ss.nwork = ss.a.mask ? (long long)cn[3] : nodes;
    ss.ncand = cn[5];
    return (!ss.a.mask || cn[3] <= ss.cap_nodes_e) && cn[5] <= ss.cap_cand_e;
}
// what to size the buffers for, from what was needed last (the band drifts from call to call)
static unsigned headroom(unsigned n) { return n ? (unsigned)std::min<unsigned long long>((unsigned long long)n + n / 8 + 256, 0xffffffffull) : 0u; }

// returns 0 on success; out_counts = {samples kept, nodes whose solve did not converge, nodes with no sample at all}
int reinit_run(int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, long long total, const double lc[3],
               const double h[3], int order, int upsample, int maxiters, double xtol, double ftol, void* phi, int f32, const unsigned char* mask,
               void* out_field, hipStream_t stream, long long out_counts[3], const char** err, ReinitWorkspace** wsp) {
    // the buffers live in the caller's workspace (the handle's) when there is one: no allocation after the first calls
    ReinitWorkspace local;
    if (wsp && !*wsp) *wsp = new ReinitWorkspace();
    ReinitWorkspace& W = wsp ? **wsp : local;
    SampleSet& ss = W.ss;
    auto done = [&](int rc) {
        if (!wsp) { ss.release(); (void)hipFree(W.seeds); (void)hipFree(W.retry); (void)hipFree(W.foot); W.seeds = nullptr; W.retry = nullptr; W.foot = nullptr; }
        return rc;
    };
    if (int r = setup_args(ss.a, ndim, n, goff, s1, s2, origin, lc, h, order, upsample, maxiters, xtol, ftol, phi, f32, mask, err)) return done(r);
    const ReinitArgs& a = ss.a;
    long long nc = 1;
    for (int d = 0; d < ndim; ++d) nc *= n[d] - 1;
    unsigned cn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // One round = everything, launched without a host round trip: node list, candidate cells, samples, nearest samples, closest points,
    // commit, un-marking.  The commit kernel leaves ϕ alone when a list did not fit; the round is then repeated with larger buffers
    // (the first call on a workspace: one round to count the band, one to count the candidate cells, one that fits).
    for (int round = 0;; ++round) {
        // nothing known about the band or about its candidate cells (the first call on the handle): a round that only counts them
        const bool count_only = (a.mask && W.last_nodes == 0) || W.last_cand == 0;
        if (int r = launch_samples(ss, total, stream, err, wsp != nullptr, std::max(headroom(W.last_nodes), ss.cap_nodes_e), std::max(headroom(W.last_cand), ss.cap_cand_e),
                                   count_only))
            return r;     // (launch_samples released ss)
        const int S = ss.S;
        const long long nwork = ss.cap_nodes_e;          // what the per-node buffers and grids are sized for
        if (nwork && !count_only) {
            if (grow(W.seeds, W.cap_seeds, sizeof(long long) * NSEED * (size_t)nwork) != hipSuccess) { *err = "hipMalloc(seeds)"; ss.release(); return done(2); }
            if (grow(W.retry, W.cap_retry, sizeof(unsigned) * (size_t)nwork) != hipSuccess) { *err = "hipMalloc(retry)"; ss.release(); return done(2); }
            if (grow(W.foot, W.cap_foot, sizeof(long long) * (size_t)nwork) != hipSuccess) { *err = "hipMalloc(foot)"; ss.release(); return done(2); }
            long long* seeds = W.seeds;
            const long long ngrid = a.mask ? (long long)std::max(headroom(W.last_nodes), 1u) : nwork;      // nodes expected
            const unsigned gsr = (unsigned)((ngrid + 255) / 256 > 262144 ? 262144 : (ngrid + 255) / 256);
            const long long grp_blocks = (ngrid * GRP + 255) / 256;
            const dim3 gg((unsigned)(grp_blocks > 1048576 ? 1048576 : grp_blocks));
#define LSM_BY_ND(KERNEL, GRID, ...) do { \
            if (ndim == 3) hipLaunchKernelGGL(KERNEL<3>, GRID, dim3(256), 0, stream, __VA_ARGS__); \
            else if (ndim == 2) hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, stream, __VA_ARGS__); \
            else hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(256), 0, stream, __VA_ARGS__); } while (0)
#define LSM_NEWTON_K(NV_, ND_, PASS, GRID) hipLaunchKernelGGL((reinit_newton_kernel<NV_, ND_>), GRID, dim3(128), 0, stream, a, S, ss.pts, ss.node_list, ss.dn, seeds, out_field, \
                                                             ss.counters + 1, ss.counters + 2, W.retry, ss.counters + 6, PASS)
#define LSM_NEWTON(PASS, GRID) do { \
            if (a.nv == 4 && ndim == 3) LSM_NEWTON_K(4, 3, PASS, GRID); \
            else if (a.nv == 4 && ndim == 2) LSM_NEWTON_K(4, 2, PASS, GRID); \
            else if (a.nv == 2 && ndim == 3) LSM_NEWTON_K(2, 3, PASS, GRID); \
            else if (a.nv == 2 && ndim == 2) LSM_NEWTON_K(2, 2, PASS, GRID); \
            else if (a.nv == 2) LSM_NEWTON_K(2, 0, PASS, GRID); \
            else if (a.nv == 4) LSM_NEWTON_K(4, 0, PASS, GRID); \
            else LSM_NEWTON_K(6, 0, PASS, GRID); } while (0)
            // nearest sample per node (exact); what the guided search cannot settle goes to the shell search
            LSM_BY_ND(reinit_foot_kernel, dim3(gsr), a, ss.node_list, ss.dn, W.foot);
            LSM_BY_ND(reinit_search_group_kernel, gg, a, ss.cand_id, S, ss.pts, ss.cnt, ss.bits, ss.node_list, ss.dn, W.foot, seeds);
            LSM_BY_ND(reinit_search_kernel, dim3(gsr), a, ss.cand_id, S, ss.pts, ss.cnt, ss.sup, ss.bits, ss.node_list, ss.dn, seeds);
            const unsigned gn = (unsigned)((ngrid + 127) / 128 > 262144 ? 262144 : (ngrid + 127) / 128);
            LSM_NEWTON(0, dim3(gn));
            // second pass for the nodes whose solve from the nearest sample did not converge (usually none: two short launches)
            LSM_BY_ND(reinit_search_kernel, dim3(gsr), a, ss.cand_id, S, ss.pts, ss.cnt, ss.sup, ss.bits, ss.node_list, ss.dn, seeds);
            LSM_NEWTON(1, dim3(gn > 1024 ? 1024 : gn));
#undef LSM_NEWTON
#undef LSM_NEWTON_K
#undef LSM_BY_ND
            hipLaunchKernelGGL(reinit_commit_kernel, dim3((unsigned)((ngrid + 255) / 256 > 65535 ? 65535 : (ngrid + 255) / 256)), dim3(256), 0, stream, a,
                               out_field, phi, ss.node_list, ss.dn, ss.dc);
        }
        hipError_t e = hipMemcpyAsync(cn, ss.counters, sizeof(cn), hipMemcpyDeviceToHost, stream);
        // a workspace serving a band goes back to its clean state by un-marking this round's candidate cells (behind everything that read
        // them; cells beyond the buffers' capacity were never marked)
        const bool restore = wsp && a.mask && e == hipSuccess;
        if (restore) {
            const unsigned gu = (std::max(headroom(W.last_cand), 256u) + 255) / 256;
            hipLaunchKernelGGL(reinit_unmark_kernel, dim3(gu), dim3(256), 0, stream, a, ss.cand_cell, ss.dc, ss.cand_id, ss.blk, ss.bits);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { *err = "reinitialize: device error"; if (wsp) ss.release(); return done(2); }
        if (restore) ss.clean_cells = nc;
        const bool fit = samples_fit(ss, cn);
        const bool nodes_fit = !a.mask || cn[3] <= ss.cap_nodes_e;
        W.last_nodes = a.mask ? cn[3] : 0u;
        W.last_cand = nodes_fit ? cn[5] : 0u;         // (counted over a truncated node list: not the field's)
        if (fit && !count_only) break;
        if (count_only && nodes_fit && cn[5] == 0) {           // no candidate cell at all: ϕ stays, every node is one "without a sample"
            cn[1] = 0; cn[2] = (unsigned)ss.nwork;
            break;
        }
        if (round >= 4) { *err = "reinitialize: the lists did not fit their buffers after five rounds"; if (wsp) ss.release(); return done(2); }
    }
    out_counts[0] = ss.ncand; out_counts[1] = cn[1]; out_counts[2] = cn[2];
    return done(0);
}
#undef RE_HIP

// ---- NewtonSDF(ϕ; order, upsample, maxiters, xtol, ftol) (src/sdf.jl:57-78) as an object: the interface samples of a
// private copy of ϕ (the reference deep-copies the field) kept on the device, and signed-distance queries at arbitrary
// points (src/sdf.jl:80-84,113-127): exact nearest sample (the reference: KD-tree nn) -> Newton–Lagrange closest point on
// the seed cell's patch, further near samples as fall-back seeds -> sign(dot(x - cp, ∇p(cp))) · ‖x - cp‖.
struct SdfObject {
    SampleSet ss;
    void* phi_copy = nullptr;
    unsigned char* mask_copy = nullptr;
    hipStream_t stream = nullptr;
};

// One thread per query point.  The nearest sample: Chebyshev shells of cells around the (clamped) cell of x, occupied
// cells from the occupancy bits; every cell not yet visited after shell r lies at least r·hmin away from x, so the search
// stops once the best sample is nearer than that — exact.  The next shell is scanned as well for the fall-back seeds.  After four
// shells the search goes on over occupied blocks (8^N cells) and super-blocks (8^N blocks) within the ball of the NSEED-th best
// distance: a query far from the interface, or outside the grid, costs the blocks near its closest point, not r³ cells.
template <int NV>
__global__ void __launch_bounds__(128) sdf_points_kernel(ReinitArgs a, const int* cand_id, int S, const double* pts, const unsigned char* cnt,
                                                         const unsigned long long* bits, const unsigned long long* sup, long long npts, const double* xs,
                                                         double* dist, double* cps, unsigned* nfail) {
    double hmin = a.h[0], hmax = a.h[0];
    for (int d = 1; d < a.ndim; ++d) { hmin = a.h[d] < hmin ? a.h[d] : hmin; hmax = a.h[d] > hmax ? a.h[d] : hmax; }
    const int nc_[3] = {a.n[0] - 1, a.ndim > 1 ? a.n[1] - 1 : 1, a.ndim > 2 ? a.n[2] - 1 : 1};
    int rmax = nc_[0];
    for (int d = 1; d < a.ndim; ++d) rmax = nc_[d] > rmax ? nc_[d] : rmax;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npts; p += (long long)gridDim.x * blockDim.x) {
        double xq[3] = {0, 0, 0};
        for (int d = 0; d < a.ndim; ++d) xq[d] = xs[p * a.ndim + d];
        int C[3];
        cell_of(a, xq, C);
        double bd[NSEED];
        long long bs[NSEED];
        for (int k = 0; k < NSEED; ++k) { bd[k] = __builtin_inf(); bs[k] = -1; }
        auto scan_cell = [&](int c0, int c1, int c2) {
            const int J[3] = {c0, c1, c2};
            const int id = cand_id[cell_lin(a, J)];
            if (id < 0) return;
            const int m = cnt[id];
            for (int k = 0; k < m; ++k) {
                const long long slot = (long long)id * S + k;
                double d2 = 0.0;
                for (int d = 0; d < a.ndim; ++d) { const double e = pts[3 * slot + d] - xq[d]; d2 += e * e; }
                if (!(d2 < bd[NSEED - 1])) continue;
                bool dup = false;                         // (the block scan below revisits cells of the shells)
                for (int q = 0; q < NSEED; ++q) dup = dup || bs[q] == slot;
                if (dup) continue;
                int q = NSEED - 1;                        // sorted insertion
                while (q > 0 && d2 < bd[q - 1]) { bd[q] = bd[q - 1]; bs[q] = bs[q - 1]; --q; }
                bd[q] = d2; bs[q] = slot;
            }
        };
        auto scan_row = [&](int lo, int hi, int c1, int c2) {      // cells lo..hi of row (c1, c2), by occupancy words
            if (c1 < 0 || c1 >= nc_[1] || c2 < 0 || c2 >= nc_[2]) return;
            lo = lo < 0 ? 0 : lo; hi = hi >= nc_[0] ? nc_[0] - 1 : hi;
            if (lo > hi) return;
            const long long row = bits_row(a, c1, c2);
            for (int w0 = lo >> 6; w0 <= (hi >> 6); ++w0) {
                unsigned long long m = bits[row + w0];
                if (w0 == (lo >> 6)) m &= ~0ull << (lo & 63);
                if (w0 == (hi >> 6)) m &= ~0ull >> (63 - (hi & 63));
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    scan_cell(w0 * 64 + b, c1, c2);
                }
            }
        };
        int extra = 1;                                    // shells still to scan after the nearest sample is settled
        constexpr int NEAR_SHELLS = 4;                    // beyond them: blocks and super-blocks (below)
        for (int r = 0; r <= (rmax < NEAR_SHELLS ? rmax : NEAR_SHELLS) && extra >= 0; ++r) {
            const int r1 = a.ndim > 1 ? r : 0, r2 = a.ndim > 2 ? r : 0;
            for (int d2i = -r2; d2i <= r2; ++d2i)
                for (int d1i = -r1; d1i <= r1; ++d1i) {
                    const bool face = (a.ndim > 2 && (d2i == -r || d2i == r)) || (a.ndim > 1 && (d1i == -r || d1i == r));
                    if (face || r == 0) scan_row(C[0] - r, C[0] + r, C[1] + d1i, C[2] + d2i);
                    else { scan_row(C[0] - r, C[0] - r, C[1] + d1i, C[2] + d2i); scan_row(C[0] + r, C[0] + r, C[1] + d1i, C[2] + d2i); }
                }
            if (bs[0] >= 0 && bd[0] <= ((double)r * hmin) * ((double)r * hmin)) --extra;
        }
        // far from the interface (or from the grid): the two occupancy levels above the cells, as in reinit_search_kernel — with
        // nothing found yet the nearest occupied super-block first, then every super-block, and in it every occupied block, that
        // meets the ball of the best distance so far
        if (extra >= 0 && rmax > NEAR_SHELLS) {
            const int nb_[3] = {(nc_[0] + RB - 1) / RB, (nc_[1] + RB - 1) / RB, (nc_[2] + RB - 1) / RB};
            const int ns_[3] = {(nb_[0] + 7) / 8, (nb_[1] + 7) / 8, (nb_[2] + 7) / 8};
            constexpr int SW = RB * 8;
            const int nword = a.ndim == 3 ? 8 : 1;
            auto box_d2 = [&](int c0, int c1, int c2, int w) {           // squared distance from x to the cells [c, c + w) per dimension
                const int c[3] = {c0, c1, c2};
                double d2 = 0.0;
                for (int d = 0; d < a.ndim; ++d) {
                    const int top = c[d] + w < nc_[d] ? c[d] + w : nc_[d];
                    const double lo = a.lc[d] + (double)(c[d] + a.goff[d]) * a.h[d], hi = a.lc[d] + (double)(top + a.goff[d]) * a.h[d];
                    const double e = xq[d] < lo ? lo - xq[d] : (xq[d] > hi ? xq[d] - hi : 0.0);
                    d2 += e * e;
                }
                return d2;
            };
            auto scan_super = [&](int s0, int s1, int s2) {
                const long long sl = s0 + (long long)ns_[0] * (s1 + (long long)ns_[1] * s2);
                for (int wd = 0; wd < nword; ++wd) {
                    unsigned long long m = sup[8 * sl + wd];
                    while (m) {
                        const int l = wd * 64 + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int b0 = s0 * 8 + (l & 7), b1 = a.ndim > 1 ? s1 * 8 + ((l >> 3) & 7) : 0, b2 = a.ndim > 2 ? s2 * 8 + (l >> 6) : 0;
                        if (box_d2(b0 * RB, b1 * RB, b2 * RB, RB) > bd[NSEED - 1]) continue;       // (the fall-back seeds want the NSEED nearest)
                        for (int c2 = b2 * RB; c2 < (a.ndim > 2 ? (b2 + 1) * RB : 1); ++c2)
                            for (int c1 = b1 * RB; c1 < (a.ndim > 1 ? (b1 + 1) * RB : 1); ++c1)
                                scan_row(b0 * RB, (b0 + 1) * RB - 1, c1, c2);
                    }
                }
            };
            long long first = -1;
            if (bs[0] < 0) {
                double best = __builtin_inf();
                int f[3] = {0, 0, 0};
                for (int s2 = 0; s2 < ns_[2]; ++s2)
                    for (int s1 = 0; s1 < ns_[1]; ++s1)
                        for (int s0 = 0; s0 < ns_[0]; ++s0) {
                            const long long sl = s0 + (long long)ns_[0] * (s1 + (long long)ns_[1] * s2);
                            unsigned long long any = 0;
                            for (int wd = 0; wd < nword; ++wd) any |= sup[8 * sl + wd];
                            if (!any) continue;
                            const double d2 = box_d2(s0 * SW, s1 * SW, s2 * SW, SW);
                            if (d2 < best) { best = d2; first = sl; f[0] = s0; f[1] = s1; f[2] = s2; }
                        }
                if (first >= 0) scan_super(f[0], f[1], f[2]);
            }
            if (bs[0] >= 0)
                for (int s2 = 0; s2 < ns_[2]; ++s2)
                    for (int s1 = 0; s1 < ns_[1]; ++s1)
                        for (int s0 = 0; s0 < ns_[0]; ++s0) {
                            if (s0 + (long long)ns_[0] * (s1 + (long long)ns_[1] * s2) == first) continue;
                            if (box_d2(s0 * SW, s1 * SW, s2 * SW, SW) > bd[NSEED - 1]) continue;
                            scan_super(s0, s1, s2);
                        }
        }
        double cp[3] = {xq[0], xq[1], xq[2]}, g[3] = {0, 0, 0};
        bool conv = false, have = false;
        if (bs[0] >= 0) {
            const double safeguard = 1.5 * hmax;
            double bestd = __builtin_inf();
            for (int k = 0; k < NSEED && bs[k] >= 0 && !conv; ++k) {
                const double seed[3] = {pts[3 * bs[k]], pts[3 * bs[k] + 1], pts[3 * bs[k] + 2]};
                int J[3];
                cell_of(a, seed, J);
                double c3[3];
                conv = closest_on_patch<NV, 0>(a, J, xq, seed, safeguard, c3);
                double d2 = 0.0;
                for (int d = 0; d < a.ndim; ++d) d2 += (xq[d] - c3[d]) * (xq[d] - c3[d]);
                if (conv || d2 < bestd) {
                    bestd = d2; cp[0] = c3[0]; cp[1] = c3[1]; cp[2] = c3[2];
                    double val, H[6];
                    patch_eval<NV>(a, J, c3, false, val, g, H);      // ∇p(cp) on the seed cell's patch (src/sdf.jl:102-104)
                    have = true;
                }
            }
        }
        if (!conv) atomicAdd(nfail, 1u);
        double d2 = 0.0, dot = 0.0;
        for (int d = 0; d < a.ndim; ++d) { d2 += (xq[d] - cp[d]) * (xq[d] - cp[d]); dot += (xq[d] - cp[d]) * g[d]; }
        const double sgn = dot > 0 ? 1.0 : (dot < 0 ? -1.0 : dot);
        dist[p] = have ? sgn * sqrt(d2) : __builtin_nan("");      // no interface sample at all: NaN
        if (cps) for (int d = 0; d < a.ndim; ++d) cps[p * a.ndim + d] = cp[d];
    }
}

// compacts the valid samples into out (nsamples x ndim, point-major); count only when out == NULL
__global__ void __launch_bounds__(256) sdf_samples_kernel(int ndim, unsigned ncand, int S, const double* pts, const unsigned char* cnt, double* out,
                                                          unsigned long long* count) {
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < ncand; id += gridDim.x * blockDim.x) {
        const int m = cnt[id];
        if (!m) continue;
        const unsigned long long at = atomicAdd(count, (unsigned long long)m);
        if (!out) continue;
        for (int k = 0; k < m; ++k)
            for (int d = 0; d < ndim; ++d) out[(at + k) * ndim + d] = pts[3 * ((long long)id * S + k) + d];
    }
}

int sdf_build(int ndim, const int n[3], const int goff[3], long long s1, long long s2, long long origin, long long total, const double lc[3],
              const double h[3], int order, int upsample, int maxiters, double xtol, double ftol, const void* phi, int f32,
              const unsigned char* mask, hipStream_t stream, SdfObject** out, long long* nsamples, const char** err) {
    SdfObject* o = new SdfObject();
    o->stream = stream;
    const size_t bytes = (size_t)total * (f32 ? 4 : 8);
    auto bail = [&](const char* what) { *err = what; (void)hipFree(o->phi_copy); (void)hipFree(o->mask_copy); o->ss.release(); delete o; return 2; };
    if (hipMalloc(&o->phi_copy, bytes) != hipSuccess) return bail("hipMalloc(phi copy)");
    if (hipMemcpyAsync(o->phi_copy, phi, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return bail("copy of phi");
    if (mask) {
        if (hipMalloc((void**)&o->mask_copy, (size_t)total) != hipSuccess) return bail("hipMalloc(mask copy)");
        if (hipMemcpyAsync(o->mask_copy, mask, (size_t)total, hipMemcpyDeviceToDevice, stream) != hipSuccess) return bail("copy of mask");
    }
    if (int r = setup_args(o->ss.a, ndim, n, goff, s1, s2, origin, lc, h, order, upsample, maxiters, xtol, ftol, o->phi_copy, f32, o->mask_copy, err)) {
        (void)hipFree(o->phi_copy); (void)hipFree(o->mask_copy); delete o; return r;
    }
    // rounds of launch_samples until the node list and the candidate cells fit their buffers (a fresh object: two or three)
    unsigned want_nodes = 0, want_cand = 0;
    for (int round = 0;; ++round) {
        if (int r = launch_samples(o->ss, total, stream, err, false, want_nodes, want_cand)) { (void)hipFree(o->phi_copy); (void)hipFree(o->mask_copy); delete o; return r; }
        unsigned cn[8];
        if (hipMemcpyAsync(cn, o->ss.counters, sizeof(cn), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return bail("candidate counts");
        if (samples_fit(o->ss, cn)) break;
        if (round >= 3) return bail("NewtonSDF: the lists did not fit their buffers after four rounds");
        want_nodes = std::max(want_nodes, mask ? cn[3] : 0u);
        want_cand = std::max(want_cand, cn[5]);
    }
    unsigned long long* cnt = nullptr;
    unsigned long long hc = 0;
    if (hipMalloc((void**)&cnt, 8) != hipSuccess || hipMemsetAsync(cnt, 0, 8, stream) != hipSuccess) return bail("hipMalloc(count)");
    if (o->ss.ncand)
        hipLaunchKernelGGL(sdf_samples_kernel, dim3((o->ss.ncand + 255) / 256), dim3(256), 0, stream, ndim, o->ss.ncand, o->ss.S, o->ss.pts, o->ss.cnt,
                           (double*)nullptr, cnt);
    hipError_t e = hipMemcpyAsync(&hc, cnt, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(cnt);
    if (e != hipSuccess) return bail("sample count");
    *nsamples = (long long)hc;
    *out = o;
    return 0;
}
int sdf_eval(SdfObject* o, long long npts, const double* xs, double* dist, double* cps, long long* nfail, const char** err) {
    const ReinitArgs& a = o->ss.a;
    if (npts <= 0) { if (nfail) *nfail = 0; return 0; }
    (void)hipMemsetAsync(o->ss.counters + 1, 0, sizeof(unsigned), o->stream);
    const unsigned gb = (unsigned)((npts + 127) / 128 > 65535 ? 65535 : (npts + 127) / 128);
#define LSM_SDF(NV_) hipLaunchKernelGGL(sdf_points_kernel<NV_>, dim3(gb), dim3(128), 0, o->stream, a, o->ss.cand_id, o->ss.S, o->ss.pts, o->ss.cnt, o->ss.bits, o->ss.sup, npts, xs, dist, cps, o->ss.counters + 1)
    if (a.nv == 2) LSM_SDF(2); else if (a.nv == 4) LSM_SDF(4); else LSM_SDF(6);
#undef LSM_SDF
    unsigned nf = 0;
    hipError_t e = hipMemcpyAsync(&nf, o->ss.counters + 1, sizeof(unsigned), hipMemcpyDeviceToHost, o->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(o->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { *err = "NewtonSDF evaluation: device error"; return 2; }
    if (nfail) *nfail = nf;
    return 0;
}
int sdf_samples(SdfObject* o, double* out, const char** err) {
    unsigned long long* cnt = nullptr;
    if (hipMalloc((void**)&cnt, 8) != hipSuccess || hipMemsetAsync(cnt, 0, 8, o->stream) != hipSuccess) { *err = "hipMalloc(count)"; return 2; }
    if (o->ss.ncand)
        hipLaunchKernelGGL(sdf_samples_kernel, dim3((o->ss.ncand + 255) / 256), dim3(256), 0, o->stream, o->ss.a.ndim, o->ss.ncand, o->ss.S, o->ss.pts,
                           o->ss.cnt, out, cnt);
    const hipError_t e = hipStreamSynchronize(o->stream);
    (void)hipFree(cnt);
    if (e != hipSuccess) { *err = "NewtonSDF samples: device error"; return 2; }
    return 0;
}
void sdf_free(SdfObject* o) {
    if (!o) return;
    o->ss.release();
    (void)hipFree(o->phi_copy);
    (void)hipFree(o->mask_copy);
    delete o;
}

}  // namespace lsm
