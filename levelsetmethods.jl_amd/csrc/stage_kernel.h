// stage_kernel.h — the fused RK-stage kernel: one loop body of _advance!
// (src/timestepping.jl:128-137,143-164,170-202) for all terms of a pass, one thread per grid node.
//
// Decomposition (CDNA4, 64-wide waves):
//   * a workgroup owns a TX×TY tile of the leading dimension(s) and MARCHES along the last
//     dimension (z in 3-D, y in 2-D): the stencil line along the march axis lives in registers
//     (2G+1 doubles per thread), the current plane's tile + halo lives in LDS for the x/y
//     neighbours.  Each ψ value is fetched from HBM/L2 once per tile (+halo), every neighbour
//     read is an LDS ds_read_b64 or a register.
//   * the LDS planes form a ring of 2·LEAD+2 slots (LEAD = 1 when the curvature term needs the
//     edge-diagonal neighbours of planes m±1): ONE s_barrier per plane.
//   * every global load of a plane — the next plane's centre and halo values, this plane's ϕⁿ, coefficient
//     fields, frozen sign, the band mask byte of the next plane — is issued BEFORE the plane's barrier and
//     consumed behind its arithmetic; the store is the last instruction of the iteration.  Vector loads return in
//     order, so the memory instructions of the loop are kept free of control flow (zero-range descriptors and
//     out-of-range offsets instead of branches): the compiler's wait counts are exact and nothing is waited for
//     that has not had a plane's arithmetic (≈230 vector instructions for WENO5 + Eikonal) to arrive.
//   * field accesses are raw buffer loads/stores (scalar plane descriptor + 32-bit lane offset), march-axis
//     table entries scalar loads; "plain" variants (template parameter AK) fix what the general kernel reads from
//     its arguments, which keeps the scalar state inside the SGPR file and the kernel at 5 waves per SIMD.
//   * upwind selection never diverges: waves with one sign of u_d (almost all) take a scalar branch to a version
//     with the stencil direction fixed at compile time; the others select per lane (sign-flipped LDS stride in
//     x,y; v_cndmask on the register line along the march axis).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "lsm_internal.h"
#include "stage_math.h"

namespace lsm {
namespace LSM_NS {

#if LSM_STRICT
constexpr bool ADV_SCALED = false;
#else
constexpr bool ADV_SCALED = true;    // the advection velocity is carried as u_d/h_d (see coeff_prep)
#endif

constexpr int halo_of(int ADV, int NM, int CURV, int EIK) {
    int g = 0;
    if (ADV == 2) g = 3;
    if (ADV == 1 && g < 1) g = 1;
    if ((NM || EIK) && g < 2) g = 2;
    if (CURV && g < 1) g = 1;
    return g;
}

// a wave-uniform pointer pinned to SGPRs (the buffer descriptors below must be scalar)
template <class T>
LSM_DEV T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// read-only table entry at a wave-uniform index: a scalar load (s_load_dwordx2) instead of a vector load, so that
// waiting for it does not wait for the plane prefetch in flight (vector loads return in order)
LSM_DEV double ld_uniform(const double* t, int i) {
    typedef const __attribute__((address_space(4))) double* cptr;
    return *((cptr)(unsigned long long)t + i);
}
// Field access = wave-uniform base (a plane of the padded array; any 64-bit address) + a 32-bit unsigned BYTE
// offset inside the plane, as a raw buffer access: the descriptor is built from the base on the scalar unit, the
// offset goes in as it is — no 64-bit vector address arithmetic (global_load needs one v_lshl_add_u64 per access
// once LLVM has hoisted the offset's zero-extension out of the plane loop).  The descriptor's range is the 2 GiB
// above the base (lsm_stage refuses layouts whose planes are larger), so nothing is clamped that the kernel's own
// index logic allows, and LSM_OOB_OFFSET is out of range under every reading of the range rule: such a load
// returns 0 without touching memory.
// ST is the storage type of the field (double, or float for LSM_DTYPE_F32: values widen exactly on load,
// all arithmetic is fp64, the result is rounded to nearest on store).
typedef unsigned lsm_v2u __attribute__((ext_vector_type(2)));
LSM_DEV __amdgpu_buffer_rsrc_t plane_rsrc(const void* base, int range = (int)0x80000000u) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, range, 0x00020000);
}
constexpr unsigned LSM_OOB_OFFSET = 0xC0000000u;
// AUX = cache policy bits of the buffer instruction (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Everything is loaded and stored with the
// default policy: non-temporal ϕⁿ loads and result stores were measured in round 2 and were slower everywhere (upwind +25 %).
template <class ST, int AUX = 0>
LSM_DEV double ldg(const ST* base, unsigned boff, int range = (int)0x80000000u) {   // range 0: returns 0, no access
    if constexpr (sizeof(ST) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(plane_rsrc(base, range), boff, 0, AUX));
    else return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane_rsrc(base, range), boff, 0, AUX));
}
template <class ST, int AUX = 0>
LSM_DEV void stg(ST* base, unsigned boff, double v) {
    if constexpr (sizeof(ST) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lsm_v2u, v), plane_rsrc(base), boff, 0, AUX);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), plane_rsrc(base), boff, 0, AUX);
}
// two neighbouring elements in one access (16 bytes for double, 8 for float): the pair kernels below (stage_tile2)
typedef unsigned lsm_v4u __attribute__((ext_vector_type(4)));
template <class ST, int AUX = 0>
LSM_DEV void ldg2(const ST* base, unsigned boff, double& x, double& y, int range = (int)0x80000000u) {
    if constexpr (sizeof(ST) == 8) {
        const lsm_v4u v = __builtin_bit_cast(lsm_v4u, __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc(base, range), boff, 0, AUX));
        x = __builtin_bit_cast(double, lsm_v2u{v.x, v.y});
        y = __builtin_bit_cast(double, lsm_v2u{v.z, v.w});
    } else {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(plane_rsrc(base, range), boff, 0, AUX));
        x = (double)__builtin_bit_cast(float, (unsigned)u);
        y = (double)__builtin_bit_cast(float, (unsigned)(u >> 32));
    }
}
template <class ST, int AUX = 0>
LSM_DEV void stg2(ST* base, unsigned boff, double x, double y) {
    if constexpr (sizeof(ST) == 8) {
        const lsm_v2u a = __builtin_bit_cast(lsm_v2u, x), b = __builtin_bit_cast(lsm_v2u, y);
        __builtin_amdgcn_raw_buffer_store_b128(lsm_v4u{a.x, a.y, b.x, b.y}, plane_rsrc(base), boff, 0, AUX);
    } else {
        const unsigned long long u = (unsigned long long)__builtin_bit_cast(unsigned, (float)x) | ((unsigned long long)__builtin_bit_cast(unsigned, (float)y) << 32);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lsm_v2u, u), plane_rsrc(base), boff, 0, AUX);
    }
}
// byte load with an explicit range: range 0 (no array) returns 0 without touching memory
LSM_DEV unsigned ldg_u8(const unsigned char* base, unsigned boff, int range) {
    return __builtin_amdgcn_raw_buffer_load_b8(
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, __builtin_amdgcn_readfirstlane(range), 0x00020000), boff, 0, 0);
}

// _eval_field (src/levelsetterms.jl:42-43) for the catalogued coefficient kinds (include/lsm.h),
// split into the part that is constant along the march axis (hoisted out of the plane loop: table
// look-ups of the leading dimensions, the in-plane rotation) and the per-plane remainder.
// Operation order is exactly ((T1*T2)*T3)*g and lc + i*h, as in the oracle.
// SCALED (FAST build, advection velocity only): the values produced are u_d/h_d — the factor 1/h_d and
// the time factor g(t) are folded into the hoisted part, so the per-node remainder is one multiply.
// KIND >= 0 fixes the coefficient kind at compile time (the "plain" kernel variants), -1 reads it from the arguments
template <int NDIM, int NCOMP, bool SCALED = false, int KIND = -1>
LSM_DEV void coeff_prep(const CoeffArgs& c, const StageArgs& a, int gi0, int gi1, double pre[3]) {
    pre[0] = pre[1] = pre[2] = 0.0;
    const int kind = KIND >= 0 ? KIND : c.kind;
    if (kind == LSM_COEFF_CONST) {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) pre[k] = SCALED ? c.v[k] * a.inv_h[k] : c.v[k];
    } else if (kind == LSM_COEFF_ROTATION) {
        const double x1 = a.lc[0] + (double)gi0 * a.h[0];
        pre[1] = c.v[0] * (x1 - c.v[1]);
        if (SCALED) pre[1] = pre[1] * a.inv_h[1];
        if (NDIM == 3) {
            const double x2 = a.lc[1] + (double)gi1 * a.h[1];
            pre[0] = -(c.v[0] * (x2 - c.v[2]));
            if (SCALED) pre[0] = pre[0] * a.inv_h[0];
        }
    } else if (kind == LSM_COEFF_SEPARABLE) {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) {
            const double* T = c.sep[k];
            double p = T[gi0];
            if (NDIM == 3) p = p * T[a.gn[0] + gi1];
            pre[k] = SCALED ? p * (c.tfac * a.inv_h[k]) : p;
        }
    }
}
template <int NDIM, int NCOMP, bool SCALED = false, int KIND = -1>
LSM_DEV void coeff_eval(const CoeffArgs& c, const StageArgs& a, const double pre[3], const double tz[3], int gim, long long plane_off,
                        unsigned ocol, double out[3]) {
    const int kind = KIND >= 0 ? KIND : c.kind;
    if (kind == LSM_COEFF_CONST) {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) out[k] = pre[k];
    } else if (kind == LSM_COEFF_ROTATION) {
        if (NDIM == 2) {
            const double x2 = a.lc[1] + (double)gim * a.h[1];
            out[0] = -(c.v[0] * (x2 - c.v[2]));
            if (SCALED) out[0] = out[0] * a.inv_h[0];
        } else {
            out[0] = pre[0];
        }
        if (NCOMP > 1) out[1] = pre[1];
        if (NCOMP > 2) out[2] = 0.0;
    } else if (kind == LSM_COEFF_SEPARABLE) {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) {
            double p = pre[k];
            if (NDIM > 1) p = p * tz[k];
            out[k] = SCALED ? p : p * c.tfac;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) {
            out[k] = ldg(uniform_ptr(c.f[k] + plane_off), ocol);
            if (SCALED) out[k] = out[k] * a.inv_h[k];
        }
    }
}

// the march-axis table entries of SEPARABLE coefficients for one plane (wave-uniform: scalar loads, fetched one
// plane ahead by the march loop)
struct PlaneTab {
    double adv[3], nm[3], curv[3];
};
template <int NDIM, int NCOMP, int KIND>
LSM_DEV void plane_tab_one(const CoeffArgs& c, const StageArgs& a, int gim, double tz[3]) {
    tz[0] = tz[1] = tz[2] = 1.0;
    if (NDIM > 1 && (KIND >= 0 ? KIND : c.kind) == LSM_COEFF_SEPARABLE) {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) tz[k] = ld_uniform(c.sep[k], (NDIM == 3 ? a.gn[0] + a.gn[1] : a.gn[0]) + gim);
    }
}
// AK: -1 = everything read from the arguments; >= 0 = a "plain" variant — the advection coefficient is of kind AK, the
// NormalMotion / curvature coefficients are constants, there is no band mask, no second output, the terms come in
// slot order.  Plain variants carry none of the other paths' scalars (the general kernel exceeds the SGPR file).
template <int NDIM, int ADV, int NM, int CURV, int AK>
LSM_DEV void plane_tab(const StageArgs& a, int gim, PlaneTab& t) {
    if constexpr (ADV != 0) plane_tab_one<NDIM, NDIM, AK>(a.adv, a, gim, t.adv);
    if constexpr (NM != 0) plane_tab_one<NDIM, 1, (AK >= 0 ? (int)LSM_COEFF_CONST : -1)>(a.nm, a, gim, t.nm);
    if constexpr (CURV != 0) plane_tab_one<NDIM, 1, (AK >= 0 ? (int)LSM_COEFF_CONST : -1)>(a.curv, a, gim, t.curv);
}

// where a node reads/writes its pointwise operands: uniform plane offset + per-thread in-plane offset
struct NodeIO {
    long long plane_off;   // offset of the plane's lowest (ghost) corner (wave-uniform)
    unsigned ocol;         // in-plane BYTE offset of this thread from that corner, in the field's storage type
    unsigned ocold;        // the same for the fp64 side arrays (coefficient fields, frozen sign) and, /8, the band mask
    int gim;               // global march index
};

// the node's pointwise operands, loaded BEFORE the plane's barrier so that their latency hides under the
// arithmetic (vector loads return in order: a load issued after the barrier would make its consumer wait for the
// next plane's prefetch as well)
struct NodeOps {
    double u[3];     // advection velocity (FAST: u_d/h_d)
    double vnm;      // NormalMotion speed
    double bcurv;    // curvature coefficient
    double s0;       // Eikonal frozen sign
    double phin;     // ϕⁿ (or the accumulator of a multi-pass stage)
    double out2;     // previous value of the second output
    // plain variants with a catalogued advection coefficient: the lanes with the sign bit of u_d set, known without a
    // per-plane compare — sign(pre_d·table entry) = sign(pre_d) xor sign(entry), pre_d fixed along the march axis
    bool have_negs;
    unsigned long long negs[3];
};
template <int NDIM, int ADV, int NM, int CURV, int EIK, class ST, int AK>
LSM_DEV void node_operands(const StageArgs& a, const NodeIO& io, const double pre_adv[3], const double pre_nm[3],
                           const double pre_curv[3], const PlaneTab& pt, NodeOps& op) {
    op.u[0] = op.u[1] = op.u[2] = 0.0;
    op.vnm = op.bcurv = op.s0 = op.phin = op.out2 = 0.0;
    op.have_negs = false;
    if constexpr (ADV != 0) coeff_eval<NDIM, NDIM, ADV_SCALED, AK>(a.adv, a, pre_adv, pt.adv, io.gim, io.plane_off, io.ocold, op.u);   // FAST: u_d/h_d
    if constexpr (NM != 0) {
        double vv[3];
        coeff_eval<NDIM, 1, false, (AK >= 0 ? (int)LSM_COEFF_CONST : -1)>(a.nm, a, pre_nm, pt.nm, io.gim, io.plane_off, io.ocold, vv);
        op.vnm = vv[0];
    }
    if constexpr (CURV != 0) {
        double bb[3];
        coeff_eval<NDIM, 1, false, (AK >= 0 ? (int)LSM_COEFF_CONST : -1)>(a.curv, a, pre_curv, pt.curv, io.gim, io.plane_off, io.ocold, bb);
        op.bcurv = bb[0];
    }
    if constexpr (EIK == 1) op.s0 = ldg(uniform_ptr(a.s0 + io.plane_off), io.ocold);
    // always issued (a load inside a branch costs the loop its exact wait counts); when the base is ψ the
    // descriptor's range is 0: the load returns 0 and touches no memory
    op.phin = ldg<ST, 0>(uniform_ptr(reinterpret_cast<const ST*>(a.phin) + io.plane_off), io.ocol,
                  __builtin_amdgcn_readfirstlane(a.base_mode == LSM_BASE_PSI ? 0 : (int)0x80000000u));
    if (AK < 0 && a.out2 && a.out2_accum) op.out2 = ldg(uniform_ptr(reinterpret_cast<const ST*>(a.out2) + io.plane_off), io.ocol);
}

// Everything one node needs: pointers into the LDS ring at the node's own position and the
// register line along the march axis.
// RN > 0: the register line is a RING of RN = 2G+1 entries — plane m+k sits in zl[(R + G + k) % RN], R = the
// iteration's rotation, a compile-time constant of an RN-fold unrolled plane loop: advancing a plane overwrites the
// oldest entry instead of shifting the line (2G v_mov_b64 per plane, 4 issue cycles each on gfx950).
template <int NDIM, int G, int W, int R = 0, int RN = 0>
struct NodeView {
    const double* T0;   // plane m   (own position)
    const double* Tm;   // plane m-1 (curvature only)
    const double* Tp;   // plane m+1 (curvature only)
    const double* zl;   // register line, z(0) = centre (MARCH only)
    double c;           // centre value
    // neighbour at offset k along the march axis
    LSM_DEV double z(int k) const { return zl[RN > 0 ? (R + G + k + RN) % (RN > 0 ? RN : 1) : G + k]; }
    // neighbour at offset k along dimension D
    template <int D>
    LSM_DEV double at(int k) const {
        if constexpr (D == 0) return T0[k];
        else if constexpr (D == 1 && NDIM == 3) return T0[k * W];
        else return z(k);
    }
    // neighbour at (sa along A, sb along B), A<B, for the mixed second difference
    template <int A, int B>
    LSM_DEV double corner(int sa, int sb) const {
        if constexpr (NDIM == 3 && A == 0 && B == 1) return T0[sa + sb * W];
        else {
            const double* P = sb > 0 ? Tp : Tm;
            if constexpr (A == 0) return P[sa];
            else return P[sa * W];
        }
    }
};

#if !LSM_STRICT
// FAST: u_d·weno5^{∓}(ϕ) = |u_d|/h_d · W(undivided, upwind-ordered differences) for either sign of u_d
// (the flipped stencil negates every difference and W is odd), so neither h nor the result needs a
// sign select; the stencil itself is chosen with a sign-bit mask and v_bfi_b32 (≈1.9 ns) instead of
// v_cmp + v_cndmask (≈3 ns each on gfx950).  u_d = +0 takes the left-biased stencil, where the
// reference takes the right-biased one — the term is |0|·W = 0 either way.
LSM_DEV double sel64(int m, double x, double y) {   // m = all-ones ? x : y
    return __hiloint2double((__double2hiint(x) & m) | (__double2hiint(y) & ~m), (__double2loint(x) & m) | (__double2loint(y) & ~m));
}
// Almost every wave sees ONE sign of u_d (the sign changes on a few surfaces of the domain): such a wave branches
// (scalar) to a version with the stencil direction fixed at compile time — LDS reads at immediate offsets from one
// address register, the march line used in place — and only waves that straddle a sign change run the version that
// selects per lane (≈9 integer/select instructions per dimension).  All three evaluate the same expression.
#ifndef LSM_NO_UNIFORM_PATHS
#define LSM_UNIFORM_PATHS 1
#else
#define LSM_UNIFORM_PATHS 0
#endif
template <int NDIM, int D, int G, int W, bool PQ, class NV>
LSM_DEV double weno_term(const NV& nv, const StageArgs& a, double v, bool have_negs, unsigned long long negs_known, double& P, double& Q) {
    const double epsf = 1.0e-99 * a.h2[D];
    auto lds = [&](auto Sc) {            // x / y neighbours, direction Sc::value = ±1 at compile time
        constexpr int ss = (D == 0 ? 1 : W) * decltype(Sc)::value;
        const double q0 = nv.T0[-3 * ss], q1 = nv.T0[-2 * ss], q2 = nv.T0[-ss], q4 = nv.T0[ss], q5 = nv.T0[2 * ss];
        return weno5_undivided_pq<PQ>(q1 - q0, q2 - q1, nv.c - q2, q4 - nv.c, q5 - q4, epsf, P, Q);
    };
    auto reg = [&](auto Sc) {            // march-axis neighbours from the register line
        constexpr int ss = decltype(Sc)::value;
        return weno5_undivided_pq<PQ>(nv.z(-2 * ss) - nv.z(-3 * ss), nv.z(-ss) - nv.z(-2 * ss), nv.c - nv.z(-ss),
                                      nv.z(ss) - nv.c, nv.z(2 * ss) - nv.z(ss), epsf, P, Q);
    };
    constexpr bool IN_LDS = D == 0 || (D == 1 && NDIM == 3);
    double w;
    const unsigned long long negs = have_negs ? negs_known : __builtin_amdgcn_ballot_w64(__double2hiint(v) < 0);
    if (LSM_UNIFORM_PATHS && negs == 0) {
        if constexpr (IN_LDS) w = lds(std::integral_constant<int, 1>{});
        else w = reg(std::integral_constant<int, 1>{});
    } else if (LSM_UNIFORM_PATHS && negs == __builtin_amdgcn_ballot_w64(true)) {
        if constexpr (IN_LDS) w = lds(std::integral_constant<int, -1>{});
        else w = reg(std::integral_constant<int, -1>{});
    } else {
        const int m = ~(__double2hiint(v) >> 31);   // all ones when the sign bit of u_d is clear
        double q[6];
        if constexpr (IN_LDS) {
            constexpr int st = D == 0 ? 1 : W;
            const int ss = (st & m) | (-st & ~m);
            q[0] = nv.T0[-3 * ss]; q[1] = nv.T0[-2 * ss]; q[2] = nv.T0[-ss];
            q[3] = nv.c;
            q[4] = nv.T0[ss]; q[5] = nv.T0[2 * ss];
        } else {
            q[0] = sel64(m, nv.z(-3), nv.z(3));
            q[1] = sel64(m, nv.z(-2), nv.z(2));
            q[2] = sel64(m, nv.z(-1), nv.z(1));
            q[3] = nv.c;
            q[4] = sel64(m, nv.z(1), nv.z(-1));
            q[5] = sel64(m, nv.z(2), nv.z(-2));
        }
        w = weno5_undivided_pq<PQ>(q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], epsf, P, Q);
    }
    return __builtin_fabs(v) * w;          // v = u_d/h_d
}
#endif

template <int NDIM, int D, int G, int W, class NV>
LSM_DEV double weno_dim(const NV& nv, const StageArgs& a, double v) {
    const bool up = v > 0;
    double q[6];
    if constexpr (D == 0 || (D == 1 && NDIM == 3)) {
        constexpr int st = D == 0 ? 1 : W;
        const int ss = up ? st : -st;
        q[0] = nv.T0[-3 * ss]; q[1] = nv.T0[-2 * ss]; q[2] = nv.T0[-ss];
        q[3] = nv.c;
        q[4] = nv.T0[ss]; q[5] = nv.T0[2 * ss];
    } else {
        q[0] = up ? nv.z(-3) : nv.z(3);
        q[1] = up ? nv.z(-2) : nv.z(2);
        q[2] = up ? nv.z(-1) : nv.z(1);
        q[3] = nv.c;
        q[4] = up ? nv.z(1) : nv.z(-1);
        q[5] = up ? nv.z(2) : nv.z(-2);
    }
    const double hs = up ? a.h[D] : -a.h[D];
    const double ihs = up ? a.inv_h[D] : -a.inv_h[D];
    return weno5_upwind(q, hs, ihs, 1.0e-99 * a.h2[D]);
}

template <int NDIM, int ADV, int NM, int CURV, int EIK, int G, int W, class ST, bool PLAIN, class NV>
LSM_DEV void node_update(const StageArgs& a, const NV& nv, const NodeOps& op, double& r_out, double& r_out2) {
    const double c = nv.c;
    double Ladv = 0.0, Lnm = 0.0, Lcurv = 0.0, Leik = 0.0;
    double A[3] = {0, 0, 0}, B[3] = {0, 0, 0};   // second-order ENO pairs (NormalMotion, Eikonal)
#if LSM_STRICT
    constexpr bool ENO_FROM_WENO = false;
#else
    // FAST: a WENO5 line already holds the differences of its ENO pair (stage_math.h, weno5_undivided_pq)
    constexpr bool ENO_FROM_WENO = ADV == 2 && (NM || EIK);
#endif

    // ---- AdvectionTerm: Σ_d u_d (u_d>0 ? D⁻|weno5⁻ : D⁺|weno5⁺) — src/levelsetterms.jl:73-82
    if constexpr (ADV != 0) {
        const double* u = op.u;
        auto one = [&](auto Dc) {
            constexpr int D = decltype(Dc)::value;
            const double v = u[D];
            double der;
            if constexpr (ADV == 2) {
#if LSM_STRICT
                der = weno_dim<NDIM, D, G, W>(nv, a, v);
#else
                return weno_term<NDIM, D, G, W, ENO_FROM_WENO>(nv, a, v, op.have_negs, op.negs[D], A[D], B[D]);
#endif
            } else {
#if LSM_STRICT
                der = v > 0 ? (c - nv.template at<D>(-1)) / a.h[D] : (nv.template at<D>(1) - c) / a.h[D];
#else
                der = v > 0 ? (c - nv.template at<D>(-1)) : (nv.template at<D>(1) - c);      // v = u_d/h_d
#endif
            }
            return v * der;
        };
        Ladv = one(std::integral_constant<int, 0>{});
        if constexpr (NDIM > 1) Ladv = Ladv + one(std::integral_constant<int, 1>{});
        if constexpr (NDIM > 2) Ladv = Ladv + one(std::integral_constant<int, 2>{});
    }

    // ---- second-order ENO pairs shared by NormalMotion and Eikonal — src/levelsetterms.jl:161-163,255-257
    if constexpr ((NM || EIK) && !ENO_FROM_WENO) {
        auto pr = [&](auto Dc) {
            constexpr int D = decltype(Dc)::value;
            eno2_pair(nv.template at<D>(-2), nv.template at<D>(-1), c, nv.template at<D>(1), nv.template at<D>(2), a.h[D],
                      a.h2[D], a.inv_h[D], A[D], B[D]);
        };
        pr(std::integral_constant<int, 0>{});
        if constexpr (NDIM > 1) pr(std::integral_constant<int, 1>{});
        if constexpr (NDIM > 2) pr(std::integral_constant<int, 2>{});
    }

    // ---- NormalMotionTerm — src/levelsetterms.jl:156-170
    if constexpr (NM) {
        const double v = op.vnm;
#if LSM_STRICT
        double gp = 0.0, gm = 0.0;
#pragma unroll
        for (int d = 0; d < NDIM; ++d) {
            double x = positive(A[d]) * positive(A[d]) + negative(B[d]) * negative(B[d]);
            double y = negative(A[d]) * negative(A[d]) + positive(B[d]) * positive(B[d]);
            if (d == 0) { gp = x; gm = y; } else { gp = gp + x; gm = gm + y; }
        }
        Lnm = positive(v) * lsm_sqrt(gp) + negative(v) * lsm_sqrt(gm);
#else
        double g2 = 0.0;
        const unsigned long long vpos = __builtin_amdgcn_ballot_w64(v > 0);   // one sign per wave is the rule (constant speeds)
        if (LSM_UNIFORM_PATHS && vpos == __builtin_amdgcn_ballot_w64(true)) {
            g2 = godunov_sum<NDIM, true>(A, B, a.inv_h2, a.uniform_h != 0);
        } else if (LSM_UNIFORM_PATHS && vpos == 0) {
            g2 = godunov_sum<NDIM, false>(A, B, a.inv_h2, a.uniform_h != 0);
        } else {
            const double sg = v > 0 ? 1.0 : -1.0;
#pragma unroll
            for (int d = 0; d < NDIM; ++d) g2 += godunov_term(sg, A[d], B[d], a.inv_h2[d]);
        }
        Lnm = v * fast_norm(g2);
#endif
    }

    // ---- CurvatureTerm: b κ |∇ϕ| — src/levelsetterms.jl:111-121, src/levelsetops.jl:197-244
    if constexpr (CURV) {
        const double bb[1] = {op.bcurv};
#if LSM_STRICT
        double gr[3] = {0, 0, 0}, Hd[3] = {0, 0, 0};
        double H01 = 0, H02 = 0, H12 = 0;
        auto first = [&](auto Dc) {
            constexpr int D = decltype(Dc)::value;
            const double p1 = nv.template at<D>(1), m1 = nv.template at<D>(-1);
            gr[D] = (p1 - m1) / (2 * a.h[D]);
            Hd[D] = (p1 - 2 * c + m1) / a.h2[D];
        };
        first(std::integral_constant<int, 0>{});
        if constexpr (NDIM > 1) first(std::integral_constant<int, 1>{});
        if constexpr (NDIM > 2) first(std::integral_constant<int, 2>{});
        // D2(ϕ,I,(a,b)) = (D⁰_b(I+e_a) - D⁰_b(I-e_a)) / (2h_a), upper triangle a<b — src/derivatives.jl:144-149
        auto mixed = [&](auto Ac, auto Bc) {
            constexpr int A_ = decltype(Ac)::value, B_ = decltype(Bc)::value;
            const double pp = nv.template corner<A_, B_>(1, 1), pm = nv.template corner<A_, B_>(1, -1);
            const double mp = nv.template corner<A_, B_>(-1, 1), mm = nv.template corner<A_, B_>(-1, -1);
            return ((pp - pm) / (2 * a.h[B_]) - (mp - mm) / (2 * a.h[B_])) / (2 * a.h[A_]);
        };
        if constexpr (NDIM > 1) H01 = mixed(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if constexpr (NDIM > 2) {
            H02 = mixed(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            H12 = mixed(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        }
        double q = gr[0] * gr[0];
        if constexpr (NDIM > 1) q = q + gr[1] * gr[1];
        if constexpr (NDIM > 2) q = q + gr[2] * gr[2];
        double lap = Hd[0];
        if constexpr (NDIM > 1) lap = lap + Hd[1];
        if constexpr (NDIM > 2) lap = lap + Hd[2];
        // gᵀHg in the order of LinearAlgebra.dot(x, ::Symmetric, y) (upper triangle by columns)
        double r = 0.0;
        r += gr[0] * (Hd[0] * gr[0]);
        if constexpr (NDIM > 1) {
            r += gr[1] * (Hd[1] * gr[1]);
            r += gr[0] * (H01 * gr[1]) + gr[1] * (H01 * gr[0]);
        }
        if constexpr (NDIM > 2) {
            r += gr[2] * (Hd[2] * gr[2]);
            r += gr[0] * (H02 * gr[2]) + gr[2] * (H02 * gr[0]);
            r += gr[1] * (H12 * gr[2]) + gr[2] * (H12 * gr[1]);
        }
        const double kappa = q < 2.220446049250313e-16 ? 0.0 : (lap * q - r) / pow(q, 1.5);
        Lcurv = bb[0] * kappa * lsm_sqrt(q);
#else
        // FAST.  κ|∇ϕ| = (Δϕ·q - gᵀHg)/q (the q^(3/2) and the sqrt cancel) from UNDIVIDED central differences
        //   G_d = ϕ₊ - ϕ₋ = 2h_d·D⁰_d,   S_d = ϕ₊ - 2ϕ + ϕ₋ = h_d²·D2⁰_d,   M_ab = (ϕ₊₊ - ϕ₊₋) - (ϕ₋₊ - ϕ₋₋) = 4h_a h_b·D2_ab
        // with gᵀHg = Σ g_d² H_dd + 2 Σ_{a<b} g_a g_b H_ab written on the squares q already needs: the expression is
        // homogeneous, so on equal spacings every factor cancels to ONE 1/h² (other grids scale G, S, M first).
        // 46 vector instructions instead of 64 in 3-D; within 1e-13 like every FAST path (the reference's own value
        // depends on pow and on the summation order of a 3-argument dot).
        double Gd[3] = {0, 0, 0}, Sd[3] = {0, 0, 0};
        double M01 = 0, M02 = 0, M12 = 0;
        auto first = [&](auto Dc) {
            constexpr int D = decltype(Dc)::value;
            const double p1 = nv.template at<D>(1), m1 = nv.template at<D>(-1);
            Gd[D] = p1 - m1;
            Sd[D] = __builtin_fma(-2.0, c, p1 + m1);
        };
        first(std::integral_constant<int, 0>{});
        if constexpr (NDIM > 1) first(std::integral_constant<int, 1>{});
        if constexpr (NDIM > 2) first(std::integral_constant<int, 2>{});
        auto mixed = [&](auto Ac, auto Bc) {
            constexpr int A_ = decltype(Ac)::value, B_ = decltype(Bc)::value;
            const double pp = nv.template corner<A_, B_>(1, 1), pm = nv.template corner<A_, B_>(1, -1);
            const double mp = nv.template corner<A_, B_>(-1, 1), mm = nv.template corner<A_, B_>(-1, -1);
            return (pp - pm) - (mp - mm);
        };
        if constexpr (NDIM > 1) M01 = mixed(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if constexpr (NDIM > 2) {
            M02 = mixed(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            M12 = mixed(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        }
        double scale = a.inv_h2[0], thr = 4.0 * 2.220446049250313e-16 * a.h2[0];   // q < eps(T)  <=>  ΣG² < 4h²·eps
        if (!a.uniform_h) {
            scale = 1.0;
            thr = 4.0 * 2.220446049250313e-16;
#pragma unroll
            for (int d = 0; d < NDIM; ++d) { Gd[d] = Gd[d] * a.inv_h[d]; Sd[d] = Sd[d] * a.inv_h2[d]; }
            if constexpr (NDIM > 1) M01 = M01 * (a.inv_h[0] * a.inv_h[1]);
            if constexpr (NDIM > 2) { M02 = M02 * (a.inv_h[0] * a.inv_h[2]); M12 = M12 * (a.inv_h[1] * a.inv_h[2]); }
        }
        const double g0 = Gd[0] * Gd[0], g1 = Gd[1] * Gd[1], g2 = Gd[2] * Gd[2];
        double q4 = g0, lap = Sd[0], diag = g0 * Sd[0], cross = 0.0;
        if constexpr (NDIM > 1) { q4 = q4 + g1; lap = lap + Sd[1]; diag = __builtin_fma(g1, Sd[1], diag); cross = (Gd[0] * Gd[1]) * M01; }
        if constexpr (NDIM > 2) {
            q4 = q4 + g2; lap = lap + Sd[2]; diag = __builtin_fma(g2, Sd[2], diag);
            cross = __builtin_fma(Gd[0] * Gd[2], M02, cross);
            cross = __builtin_fma(Gd[1] * Gd[2], M12, cross);
        }
        const double num = __builtin_fma(lap, q4, -__builtin_fma(0.5, cross, diag));
        Lcurv = q4 < thr ? 0.0 : (bb[0] * scale) * (num * fast_rcp(q4));
#endif
    }

    // ---- EikonalReinitializationTerm: S (|∇ϕ| - 1) — src/levelsetterms.jl:234-265
    if constexpr (EIK != 0) {
        const double s = EIK == 1 ? op.s0 : c;
#if LSM_STRICT
        const bool vpos = s > 0;
        double mA = 0.0, mB = 0.0;
#pragma unroll
        for (int d = 0; d < NDIM; ++d) {
            double x, y;
            godunov_sel(vpos, A[d], B[d], x, y);
            if (d == 0) { mA = x; mB = y; } else { mA = mA + x; mB = mB + y; }
        }
        const double n2 = mA + mB;
#else
        double n2 = 0.0;
        // the sign of ϕ is the same across most waves (away from the interface): a scalar branch to the version
        // without the sign factor
        const unsigned long long poss = __builtin_amdgcn_ballot_w64(s > 0);
        if (LSM_UNIFORM_PATHS && poss == __builtin_amdgcn_ballot_w64(true)) {
            n2 = godunov_sum<NDIM, true>(A, B, a.inv_h2, a.uniform_h != 0);
        } else if (LSM_UNIFORM_PATHS && poss == 0) {
            n2 = godunov_sum<NDIM, false>(A, B, a.inv_h2, a.uniform_h != 0);
        } else {
            const double sg = s > 0 ? 1.0 : -1.0;
#pragma unroll
            for (int d = 0; d < NDIM; ++d) n2 += godunov_term(sg, A[d], B[d], a.inv_h2[d]);
        }
#endif
#if LSM_STRICT
        const double nrm = lsm_sqrt(n2);
#else
        const double nrm = fast_norm(n2);
#endif
        double S;
        if constexpr (EIK == 1) {
            S = s;
        } else {
#if LSM_STRICT
            const double den = lsm_sqrt(c * c + (nrm * nrm) * (a.dxmin * a.dxmin));
            S = den == 0.0 ? 0.0 : lsm_div(c, den);
#else
            const double d2 = __builtin_fma(n2, a.dxmin * a.dxmin, c * c);
            S = c * fast_rsqrt(__builtin_fmax(d2, 1.0e-300));   // d2 == 0 implies c == 0: S = 0
#endif
        }
        Leik = S * (nrm - 1);
    }

    // ---- stage combination — src/timestepping.jl:129-136,147-163,172-200
    double base;
#if LSM_STRICT
    if (a.base_mode == LSM_BASE_PSI) base = c;
    else if (a.base_mode == LSM_BASE_RK3_S2) base = 0.75 * op.phin + 0.25 * c;
    else if (a.base_mode == LSM_BASE_RK3_S3) base = (op.phin + 2 * c) / 3;
    else base = op.phin;
#else
    // FAST: the four bases as ONE expression base_a·ϕⁿ + base_b·ψ with (base_a, base_b) = (0,1), (¾,¼), (⅓,⅔), (1,0):
    // no control flow between the loads issued before the barrier and this first use of ϕⁿ, which therefore stays
    // BEHIND the arithmetic (the scheduling barrier pins it there) — its latency is hidden, not waited for.
    __builtin_amdgcn_sched_barrier(0);
    base = __builtin_fma(a.base_a, op.phin, a.base_b * c);
#endif
    double b2 = !PLAIN && a.out2_accum ? op.out2 : c;
    const bool has2 = !PLAIN && a.out2 != nullptr;
    if (PLAIN || a.natural) {
        // the terms in slot order, each once (the usual case): no look-up of the order
        auto acc = [&](double L) {
            base -= a.cdt * L;
            if (has2) b2 -= a.cdt2 * L;
        };
        if constexpr (ADV != 0) acc(Ladv);
        if constexpr (NM != 0) acc(Lnm);
        if constexpr (CURV != 0) acc(Lcurv);
        if constexpr (EIK != 0) acc(Leik);
    } else {
        for (int k = 0; k < a.nterms; ++k) {
            const int o = a.order[k];
            double L = Ladv;
            if (NM && o == SLOT_NM) L = Lnm;
            if (CURV && o == SLOT_CURV) L = Lcurv;
            if (EIK && o == SLOT_EIK) L = Leik;
            base -= a.cdt * L;
            b2 -= a.cdt2 * L;
        }
    }
    r_out = base;
    r_out2 = b2;
}
template <class ST, bool PLAIN>
LSM_DEV void node_store(const StageArgs& a, const NodeIO& io, bool on, double r_out, double r_out2) {
    if (!on) return;
    stg<ST, 0>(uniform_ptr(reinterpret_cast<ST*>(a.out) + io.plane_off), io.ocol, r_out);
    if (!PLAIN && a.out2) stg(uniform_ptr(reinterpret_cast<ST*>(a.out2) + io.plane_off), io.ocol, r_out2);
}

#ifndef LSM_WAVES_PER_EU
#define LSM_WAVES_PER_EU 1
#endif
#ifndef LSM_BAND_PATCH
#define LSM_BAND_PATCH 1    // A/B (build) switch: 8 x 8 patches per wave in band mode
#endif
#ifndef LSM_ZROT
#define LSM_ZROT 1
#endif
#define LSM_BARRIER() __syncthreads()
// One tile (a 32×8 or 64×8 column of `mc` planes; a row segment in 2-D; 256 nodes in 1-D) of one stage: the body of the
// stage kernels below.  `slot` only names the tile in the diagnostic build's stamp buffer.
template <int NDIM, int ADV, int NM, int CURV, int EIK, int TX, int TY, int MC, class ST, int AK, bool MASKED>
__device__ __forceinline__ void stage_tile(const StageArgs& a, unsigned tile_id, bool tail_tile, unsigned slot) {
    constexpr bool PLAIN = AK >= 0;                 // coefficient kinds, term order and the single output fixed at compile time
    constexpr bool NOMASK = PLAIN && !MASKED;       // ... and no band mask (MASKED: a plain variant over a narrow band)
    constexpr int CK = PLAIN ? (int)LSM_COEFF_CONST : -1;   // kind of the NormalMotion / curvature coefficients
    constexpr int G = halo_of(ADV, NM, CURV, EIK);
    constexpr bool HAS_Y = NDIM == 3, MARCH = NDIM >= 2;
    constexpr int LEAD = (CURV && MARCH) ? 1 : 0;
    constexpr int NSLOT = MARCH ? 2 * LEAD + 2 : 1;
    // the march line as a register ring with the plane loop unrolled 2G+1-fold: the WENO5 kernels in 3-D (7 moves a plane)
    // (the ring for the lighter 3-D kernels too, with 0 / 1 / 2 more planes of ψ in flight: ±3 %, no pattern — round 2, not kept)
    constexpr bool ZROT = LSM_ZROT && !LSM_STRICT && NDIM == 3 && ADV == 2;
    constexpr int W = TX + 2 * G;
    constexpr int H = HAS_Y ? TY + 2 * G : 1;
    constexpr int HW = H * W;
    constexpr int NT = TX * TY;
    constexpr int NHX = 2 * G * TY;
    constexpr int WY = CURV ? W : TX;
    constexpr int NHY = HAS_Y ? 2 * G * WY : 0;
    constexpr int NH = NHX + NHY;
    constexpr int HPT = (NH + NT - 1) / NT;
    static_assert(!HAS_Y || TY > 1 || true, "");
    __shared__ double tile[NSLOT * HW];
    // (Round 4, measured and not kept: faces with ExtrapolationBC{P}, P >= 1, resolved here — the face tiles summing their weighted
    // ghosts from the nodes they hold in LDS behind a second barrier per plane, so that a step's fill writes the march axis' ghost
    // planes only (54 -> 14 µs at 512³).  Correct (FAST tolerance, tests/test_gpu_parity.py keeps the cases), and slower: the
    // NormalMotion + curvature stage 0.741 -> 0.808 ms, BASELINE config 3 2.38 -> 2.49 ms per step — the second barrier takes the
    // slack between a tile's waves and the sums cost every variant registers.  profiles/r4/exp_weighted_faces_in_kernel_ab.log.)

#ifdef LSM_STAMP
    const unsigned long long st_rb = __builtin_amdgcn_s_memrealtime();
#endif
    if (a.tile_list) tile_id = (unsigned)a.tile_list[tile_id];     // narrow band: the compact list of active tiles
    else if (a.tile_active && !a.tile_active[tile_id]) return;   // narrow band: no band node in this tile
    // tiles are numbered x fastest (y-fastest and 2 x 2 ... 8 x 8 blocks of tiles were measured in rounds 2 and 3: +2 % and ±0.1 %)
    const unsigned tbx = tile_id % a.nb[0], tby = (tile_id / a.nb[0]) % a.nb[1];
    const unsigned tbm = tile_id / (a.nb[0] * a.nb[1]);

    const int tid = threadIdx.x;
    int tx = tid % TX, ty = tid / TX;
    // Narrow band, 32×8 tile: a wave takes an 8 × 8 patch of the plane instead of two 32-node rows.  A wave without a band node
    // skips the plane's arithmetic, and the band is a shell ~8 nodes thick: where its normal points along x a row crosses it
    // in 8 of its 32 nodes — every row wave works at a quarter of its lanes — while of the four 8-wide patches one or two hold
    // all of them.  (Where the normal points along y or the march axis both shapes do the same.)  Dense launches keep the rows:
    // 256-byte segments per wave load.
    if constexpr (NDIM == 3 && TX == 32 && TY == 8 && !NOMASK) {
        if (LSM_BAND_PATCH && a.mask) { tx = 8 * (tid >> 6) + (tid & 7); ty = (tid & 63) >> 3; }
    }
    const int bx0 = tbx * TX, by0 = HAS_Y ? tby * TY : 0;
    const int gx = bx0 + tx, gy = by0 + ty;
    const int nx = a.n[0], ny = HAS_Y ? a.n[1] : 1;
    const int nm = MARCH ? a.n[NDIM - 1] : 1;
    const long long sy = HAS_Y ? a.s1 : 0;
    const long long sm = MARCH ? (NDIM == 3 ? a.s2 : a.s1) : 0;
    const bool active = gx < nx && gy < ny;
    // inactive threads of a partial tile still feed the tile: they load their own (ghost) position
    // while it lies inside the padded array; coefficient lookups use the interior-clamped index.
    const int cx = gx < nx ? gx : nx - 1, cy = gy < ny ? gy : ny - 1;
    // StageArgs::xredirect / yredirect: the node a load of index i (possibly a ghost) is served from.  Only tiles that reach
    // past a face take the map (a scalar branch); the kinds are read into scalars first (indexing the argument block with a
    // per-lane side would loop over the lanes).
    const int xk0 = a.xkind[0], xk1 = a.xkind[1], yk0 = a.ykind[0], yk1 = a.ykind[1];
    const bool xmap = a.xredirect && (bx0 < G || bx0 + TX + G > nx);
    const bool ymap = HAS_Y && a.yredirect && (by0 < G || by0 + TY + G > ny);
    auto bsrc = [](int i, int n, int k0, int k1) {
        const bool left = i < 0;
        const int k = left ? -i : i - (n - 1), kind = left ? k0 : k1;
        const int per = left ? (n - 1) - k : k, ext = left ? 0 : n - 1, sym = left ? k : (n - 1) - k;
        return kind == LSM_BC_PERIODIC ? per : (kind == LSM_BC_EXTRAPOLATION ? ext : sym);
    };
    auto xsrc = [&](int i) { return (!xmap || (i >= 0 && i < nx)) ? i : bsrc(i, nx, xk0, xk1); };
    auto ysrc = [&](int i) { return (!ymap || (i >= 0 && i < ny)) ? i : bsrc(i, ny, yk0, yk1); };
    const int lxg = xsrc(gx < nx + G ? gx : nx + G - 1);
    const int lyg = HAS_Y ? ysrc(gy < ny + G ? gy : ny + G - 1) : 0;
    // unsigned in-plane offset from the plane's lowest (ghost) corner: with a wave-uniform base this
    // selects the SGPR-base + 32-bit-VGPR-offset addressing mode (no 64-bit vector address arithmetic)
    const long long corner = a.origin - G - (HAS_Y ? (long long)G * sy : 0);
    const unsigned ocole = (unsigned)(lxg + G) + (unsigned)(lyg + (HAS_Y ? G : 0)) * (unsigned)sy;         // elements
    const unsigned ocol = (unsigned)sizeof(ST) * ocole, ocold = 8u * ocole;                                // bytes
    const int mcl = a.mc > 0 ? a.mc : MC;
    const int mc = tail_tile ? a.mc_tail : mcl;
    const int m0 = MARCH ? (tail_tile ? a.mb + (int)a.nbig * mcl + (int)(tbm - a.nbig) * mc : a.mb + (int)tbm * mc) : 0;
    const int m1 = MARCH ? (m0 + mc < a.me ? m0 + mc : a.me) : 1;
    // StageArgs::mredirect (NeumannBC on a face of the march axis): a ghost plane IS the boundary plane (degree-0 extrapolation copies
    // it, src/boundaryconditions.jl:134-144 with P = 0), so the march clamps at the boundary plane instead of at the last ghost
    // plane and that face's ghost planes are never read — nor filled.  Two scalars; the plane pointers still advance by one plane.
    const int mlo = (MARCH && a.mredirect[0]) ? 0 : -G, mhi = (MARCH && a.mredirect[1]) ? nm - 1 : nm + G - 1;
    auto clampM = [&](int p) { return p < mlo ? mlo : (p > mhi ? mhi : p); };
    // wave-uniform plane base (SGPRs) + 32-bit per-thread offset: no vector address arithmetic in the loop
    auto plane = [&](int p) { return uniform_ptr(reinterpret_cast<const ST*>(a.psi) + (corner + (long long)clampM(p) * sm)); };

    // halo elements owned by this thread: LDS offset within a plane, global offset within a plane
    int hl[HPT > 0 ? HPT : 1];
    unsigned hg[HPT > 0 ? HPT : 1];
    bool hv[HPT > 0 ? HPT : 1];
#pragma unroll
    for (int h = 0; h < HPT; ++h) {
        const int e = tid + h * NT;
        hv[h] = e < NH;
        int lx, ly;
        if (e < NHX) {
            const int r = e % (2 * G);
            ly = e / (2 * G) + (HAS_Y ? G : 0);
            lx = r < G ? r : r + TX;
        } else {
            const int e2 = e - NHX;
            const int row = e2 / WY;
            lx = (CURV ? 0 : G) + e2 % WY;
            ly = row < G ? row : row + TY;
        }
        int X = bx0 - G + lx;
        X = xsrc(X > nx + G - 1 ? nx + G - 1 : X);
        int Y = 0;
        if (HAS_Y) {
            Y = by0 - G + ly;
            Y = ysrc(Y > ny + G - 1 ? ny + G - 1 : Y);
        }
        hl[h] = ly * W + lx;
        hg[h] = (unsigned)sizeof(ST) * ((unsigned)(X + G) + (unsigned)(Y + (HAS_Y ? G : 0)) * (unsigned)sy);   // bytes
        if (!hv[h]) hg[h] = LSM_OOB_OFFSET;   // beyond the descriptor's range: the buffer load returns 0 and touches no memory
    }
    const int lpos = (ty + (HAS_Y ? G : 0)) * W + tx + G;

    // coefficient parts that are constant along the march axis
    double pre_adv[3] = {0, 0, 0}, pre_nm[3] = {0, 0, 0}, pre_curv[3] = {0, 0, 0};
    {
        const int gi0 = cx + a.goff[0], gi1 = HAS_Y ? cy + a.goff[1] : 0;
        if constexpr (ADV != 0) coeff_prep<NDIM, NDIM, ADV_SCALED, AK>(a.adv, a, gi0, gi1, pre_adv);
        if constexpr (NM != 0) coeff_prep<NDIM, 1, false, CK>(a.nm, a, gi0, gi1, pre_nm);
        if constexpr (CURV != 0) coeff_prep<NDIM, 1, false, CK>(a.curv, a, gi0, gi1, pre_curv);
    }

    if constexpr (!MARCH) {
        const ST* P = plane(0);
        const double c = ldg(P, ocol);
        const NodeIO io{corner, ocol, ocold, 0};
        NodeOps op;
        PlaneTab pt;
        plane_tab<NDIM, ADV, NM, CURV, AK>(a, 0, pt);
        node_operands<NDIM, ADV, NM, CURV, EIK, ST, AK>(a, io, pre_adv, pre_nm, pre_curv, pt, op);
        tile[lpos] = c;
#pragma unroll
        for (int h = 0; h < HPT; ++h)
            if (hv[h]) tile[hl[h]] = ldg(P, hg[h]);
        __syncthreads();
        NodeView<NDIM, G, W> nv{tile + lpos, nullptr, nullptr, nullptr, c};
        double r1, r2;
        node_update<NDIM, ADV, NM, CURV, EIK, G, W, ST, PLAIN>(a, nv, op, r1, r2);
        // narrow band: only band nodes are updated (src/timestepping.jl loops over active_nodeindices)
        node_store<ST, PLAIN>(a, io, active && (NOMASK || !a.mask || ldg_u8(uniform_ptr(a.mask + corner), ocold >> 3, (int)0x80000000u) != 0), r1, r2);
    } else {
        // the prologue is a chain of memory round trips (≈12 µs of a workgroup's ≈165 on a busy chip): issue everything that
        // does not depend on an earlier answer at once — the first plane's table entries (scalar loads), the march line and
        // the halo elements of the first LDS planes — and only then wait and fill the LDS planes
        PlaneTab pt;
        plane_tab<NDIM, ADV, NM, CURV, AK>(a, m0 + a.goff[NDIM - 1], pt);
        double zl[2 * G + 1];
#pragma unroll
        for (int j = 0; j <= 2 * G; ++j) zl[j] = ldg(plane(m0 - G + j), ocol);
        double hp[2 * LEAD + 1][HPT > 0 ? HPT : 1];
#pragma unroll
        for (int pl = -LEAD; pl <= LEAD; ++pl) {
            const ST* P = plane(m0 + pl);
#pragma unroll
            for (int h = 0; h < HPT; ++h) hp[pl + LEAD][h] = ldg(P, hg[h]);   // lanes without a halo element: out-of-range offset, no access
        }
#pragma unroll
        for (int pl = -LEAD; pl <= LEAD; ++pl) {
            const int slot = pl + LEAD;
            tile[slot * HW + lpos] = zl[G + pl];
#pragma unroll
            for (int h = 0; h < HPT; ++h)
                if (hv[h]) tile[slot * HW + hl[h]] = hp[slot][h];
        }
        // narrow band: only band nodes are updated (src/timestepping.jl loops over active_nodeindices).  The mask byte
        // of a plane is fetched one plane ahead, unconditionally: without a mask the descriptor's range is 0 and the
        // load returns 0 without an access (a load inside a branch would cost the loop its exact wait counts).
        const bool nomask = NOMASK || a.mask == nullptr;
        const int mrange = nomask ? 0 : (int)0x80000000u;
        long long po = corner + (long long)m0 * sm;     // plane m of the pointwise operands (ϕⁿ, outputs, mask, side arrays)
        unsigned mk_next = 0;
        if constexpr (!NOMASK) mk_next = ldg_u8(uniform_ptr(a.mask + po), ocold >> 3, mrange);
        const ST* Pnx = plane(m0 + G);            // plane m+G of ψ, advanced (and clamped) before each use
        const ST* Pn = plane(m0 + LEAD);
        const int plast = mhi;
        const bool any_active = __builtin_amdgcn_ballot_w64(active) != 0;
        // sign bits of the march-invariant parts of u_d (plain variants; see NodeOps::negs)
        constexpr bool SIGNS_KNOWN = PLAIN && ADV == 2 && AK != LSM_COEFF_FIELD && !LSM_STRICT;
        unsigned long long neg_pre[3] = {0, 0, 0};
        const unsigned long long all_lanes = __builtin_amdgcn_ballot_w64(true);
        if constexpr (SIGNS_KNOWN) {
#pragma unroll
            for (int d = 0; d < NDIM; ++d) neg_pre[d] = __builtin_amdgcn_ballot_w64(__double2hiint(pre_adv[d]) < 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the prologue's loads have landed; the loop counts its own
#ifdef LSM_STAMP
        // diagnostic build only (MI355X_MICROARCH.md, DVFS item 6): the clock this kernel holds = Δs_memtime ÷ Δs_memrealtime × 100 MHz
        // around the plane loop, one pair per workgroup, written where nothing else in the kernel reads
        const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        // One plane of the march.  ROT > 0: the register line is a ring (see NodeView) and the loop below is unrolled
        // ROT-fold with the rotation Rc a compile-time constant of each copy.
        constexpr int ROT = ZROT ? 2 * G + 1 : 0, RM = ROT > 0 ? ROT : 1;
        auto plane_iter = [&](auto Rc, int m) {
            constexpr int R = decltype(Rc)::value;
            // issue the next plane's loads early; they land in LDS after this plane's arithmetic
            Pnx = uniform_ptr(m + 1 + G <= plast ? Pnx + sm : Pnx);
            Pn = uniform_ptr(m + 1 + LEAD <= plast ? Pn + sm : Pn);
            const double nxt = ldg(Pnx, ocol);
            double hn[HPT > 0 ? HPT : 1];
#pragma unroll
            for (int h = 0; h < HPT; ++h) hn[h] = ldg(Pn, hg[h]);   // lanes without a halo element: out-of-range offset, no access
            const unsigned mk = mk_next;
            if constexpr (!NOMASK) mk_next = ldg_u8(uniform_ptr(a.mask + (po + sm)), ocold >> 3, mrange);
            // this plane's pointwise operands (coefficients, ϕⁿ): consumed after the arithmetic
            const NodeIO io{po, ocol, ocold, m + a.goff[NDIM - 1]};
            NodeOps op;
            node_operands<NDIM, ADV, NM, CURV, EIK, ST, AK>(a, io, pre_adv, pre_nm, pre_curv, pt, op);
            if constexpr (SIGNS_KNOWN) {
                op.have_negs = true;
#pragma unroll
                for (int d = 0; d < NDIM; ++d) {
                    // CONST, ROTATION (3-D): u_d = pre_d; ROTATION (2-D) dimension 0 is recomputed per plane (not known
                    // here); SEPARABLE: u_d = pre_d · entry of the march-axis table (wave-uniform)
                    if (AK == LSM_COEFF_SEPARABLE) op.negs[d] = neg_pre[d] ^ (__double2hiint(pt.adv[d]) < 0 ? all_lanes : 0ull);
                    else op.negs[d] = neg_pre[d];
                }
                if (AK == LSM_COEFF_ROTATION) op.have_negs = NDIM == 3;
            }
            LSM_BARRIER();
            const bool on = active && (nomask || mk != 0);
            double r1 = 0.0, r2 = 0.0;
            if (NOMASK ? any_active : __builtin_amdgcn_ballot_w64(on) != 0) {   // a wave without a node to update skips the arithmetic (band mode)
                const int rel = m - m0;
                const double* T0 = tile + ((rel + LEAD) % NSLOT) * HW + lpos;
                const double* Tm = tile + ((rel + LEAD + NSLOT - 1) % NSLOT) * HW + lpos;
                const double* Tp = tile + ((rel + LEAD + 1) % NSLOT) * HW + lpos;
                NodeView<NDIM, G, W, R, ROT> nv{T0, Tm, Tp, zl, zl[ROT > 0 ? (R + G) % RM : G]};
                node_update<NDIM, ADV, NM, CURV, EIK, G, W, ST, PLAIN>(a, nv, op, r1, r2);
            }
            // next plane's table entries: scalar loads, issued behind the LDS reads (they share a counter with them)
            plane_tab<NDIM, ADV, NM, CURV, AK>(a, (m + 1 < m1 ? m + 1 : m) + a.goff[NDIM - 1], pt);
            // advance the register line (ring: overwrite the oldest entry; otherwise shift), write the next plane to its ring slot
            if constexpr (ROT > 0) {
                zl[R % RM] = nxt;
            } else {
#pragma unroll
                for (int j = 0; j < 2 * G; ++j) zl[j] = zl[j + 1];
                zl[2 * G] = nxt;
            }
            const int wslot = (m - m0 + 1 + 2 * LEAD) % NSLOT;
            tile[wslot * HW + lpos] = zl[ROT > 0 ? (R + 1 + G + LEAD) % RM : G + LEAD];
#pragma unroll
            for (int h = 0; h < HPT; ++h)
                if (hv[h]) tile[wslot * HW + hl[h]] = hn[h];
            node_store<ST, PLAIN>(a, io, on, r1, r2);
            po += sm;
        };
        if constexpr (ROT > 0) {
            int m = m0;
            while (m < m1) {
                // ROT copies of the body; each leaves the loop when the chunk is done (scalar branch)
                bool go = true;
                auto step = [&](auto Rc) {
                    if (go) {
                        plane_iter(Rc, m);
                        ++m;
                        go = m < m1;
                    }
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
                if constexpr (ROT > 3) step(std::integral_constant<int, 3>{});
                if constexpr (ROT > 4) step(std::integral_constant<int, 4>{});
                if constexpr (ROT > 5) step(std::integral_constant<int, 5>{});
                if constexpr (ROT > 6) step(std::integral_constant<int, 6>{});
                if constexpr (ROT > 7) step(std::integral_constant<int, 7>{});
            }
        } else {
            for (int m = m0; m < m1; ++m) plane_iter(std::integral_constant<int, 0>{}, m);
        }
#ifdef LSM_STAMP
        if (a.stamp && threadIdx.x == 0) {
            const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
            a.stamp[4 * (slot % 16384u)] = st_t1 - st_t0;
            a.stamp[4 * (slot % 16384u) + 1] = st_rb;        // absolute 100 MHz stamps: kernel entry, plane loop start, plane loop end —
            a.stamp[4 * (slot % 16384u) + 2] = st_r0;        // the timeline of the launch (tools/timeline_probe.py)
            a.stamp[4 * (slot % 16384u) + 3] = st_r1;
        }
#endif
    }
}

// Tile order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2: every XCD walks a
// CONTIGUOUS range of tiles (x fastest, then y, then march chunk), so tiles sharing halo columns / rows share an L2 (speed
// only, never correctness).  Narrow band with a compact (ordered) tile list: the same dealing over the list.  With a graded
// tail (StageArgs::mc_tail) an XCD's list is its share of the long chunks followed by its share of the short ones.
struct TileOrder {
    unsigned ntiles, nbigt, big_per, small_per;
    __host__ __device__ explicit TileOrder(const StageArgs& a) {
        ntiles = a.tile_list ? a.ntile_list : a.nb[0] * a.nb[1] * a.nb[2];
        nbigt = a.mc_tail > 0 ? a.nb[0] * a.nb[1] * a.nbig : ntiles;
        big_per = (nbigt + 7u) / 8u;
        small_per = (ntiles - nbigt + 7u) / 8u;
    }
    __host__ __device__ unsigned per_xcd() const { return big_per + small_per; }
    // entry j of XCD x's list: tile id (and whether it is a tail tile), or false when the list is shorter
    __host__ __device__ bool entry(unsigned x, unsigned j, unsigned& id, bool& tail) const {
        const unsigned nb = x * big_per < nbigt ? (nbigt - x * big_per < big_per ? nbigt - x * big_per : big_per) : 0u;
        if (j < nb) { id = x * big_per + j; tail = false; return true; }
        const unsigned js = j - nb, so = nbigt + x * small_per;
        if (so + js < ntiles && js < small_per) { id = so + js; tail = true; return true; }
        return false;
    }
};

template <int NDIM, int ADV, int NM, int CURV, int EIK, int TX, int TY, int MC, class ST, int AK, bool MASKED>
__global__ void __launch_bounds__(TX* TY, (LSM_ZROT && !LSM_STRICT && NDIM == 3 && ADV == 2 && AK >= 0 && !MASKED && !CURV && !NM) ? 5 : LSM_WAVES_PER_EU)
    stage_kernel(const StageArgs a) {   // the plain dense WENO5 kernels sit two registers above the 5-waves-per-SIMD step: capped there
    const TileOrder ord(a);
    unsigned tile_id;
    bool tail_tile;
    if (NDIM == 3 && !MASKED && a.tail_wgs && blockIdx.x >= 8u * ord.big_per) {      // dynamic tail (StageArgs::tail_ctr)
        __shared__ unsigned s_ticket;
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(a.tail_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // exactly tail_wgs workgroups draw: the last ticket leaves the counter at 0 for the slot's next launch
            if (t + 1u == a.tail_wgs) __hip_atomic_store(a.tail_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ticket = t;
        }
        __syncthreads();
        const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)s_ticket);
        if (t >= ord.ntiles - ord.nbigt) return;
        tile_id = ord.nbigt + t;
        tail_tile = true;
    } else if (!ord.entry(blockIdx.x % 8u, blockIdx.x / 8u, tile_id, tail_tile)) {
        return;   // whole workgroup leaves before any barrier
    }
    stage_tile<NDIM, ADV, NM, CURV, EIK, TX, TY, MC, ST, AK, MASKED>(a, tile_id, tail_tile, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------------
// Two nodes per thread along x (round 3): the memory-bound members of the family — a single upwind / NormalMotion / Eikonal
// term with constant coefficients on a dense 3-D field — move their field with 16-byte accesses: a hand-written copy
// of the same padded array runs at 6.2–6.3 TB/s with 16 bytes per lane against 5.7–5.9 with 8 (tools/copy_bw.hip), and a tile
// twice as wide halves the share of the x halo.  Thread (tx, ty) owns nodes (2tx, 2tx+1) of row ty of a 2TX × TY tile and
// marches like stage_tile (shifting register lines, one barrier per plane, every load of a plane issued before its barrier);
// node_update runs once per node on the same LDS tile, so the arithmetic — and every bit of the result — is stage_tile's.
// Plain variants only (FAST build, constants, one output, no band mask), n[0] even; everything else takes stage_tile.
template <int ADV, int NM, int CURV, int EIK, int TX, int TY, int MC, class ST>
__device__ __forceinline__ void stage_tile2(const StageArgs& a, unsigned tile_id) {
    constexpr int NDIM = 3;
    constexpr int AK = LSM_COEFF_CONST;
    // No curvature here.  A variant with the three-plane LDS ring curvature's edge diagonals need (planes m-1, m, m+1 resident, as in
    // stage_tile) was written in round 3 and gave wrong values at the FIRST node of each pair from the second plane of a chunk
    // on: that node kept the chunk's first centre value — the compiled march held no move into its z(+1) register, although the
    // source shifted both nodes' lines alike, and splitting the shift per node did not change the code.  Source or compiler was
    // not decided; the variant gained 4 % on a kernel that is not on any BASELINE config's path.  It is fenced, not carried:
    static_assert(!CURV, "stage_tile2 serves the axis-aligned single terms; curvature stays with stage_tile (see the comment above)");
    constexpr int G = halo_of(ADV, NM, CURV, EIK);
    constexpr int NSLOT = 2;
    constexpr int TXN = 2 * TX;
    constexpr int W = TXN + 2 * G, H = TY + 2 * G, HW = H * W, NT = TX * TY;
    constexpr int NHX = 2 * G * TY;
    constexpr int WY = TXN;
    constexpr int NHY = 2 * G * WY;
    constexpr int NH = NHX + NHY;
    constexpr int HPT = (NH + NT - 1) / NT;
    __shared__ double tile[NSLOT * HW];

    const unsigned tbx = tile_id % a.nb[0], tby = (tile_id / a.nb[0]) % a.nb[1], tbm = tile_id / (a.nb[0] * a.nb[1]);
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int bx0 = tbx * TXN, by0 = tby * TY;
    const int gx = bx0 + 2 * tx, gy = by0 + ty;
    const int nx = a.n[0], ny = a.n[1], nm = a.n[2];
    const long long sy = a.s1, sm = a.s2;
    const bool active = gx < nx && gy < ny;          // n[0] is even: the two nodes of a pair are in range together
    const int xk0 = a.xkind[0], xk1 = a.xkind[1], yk0 = a.ykind[0], yk1 = a.ykind[1];
    const bool xmap = a.xredirect && (bx0 < G || bx0 + TXN + G > nx);
    const bool ymap = a.yredirect && (by0 < G || by0 + TY + G > ny);
    auto bsrc = [](int i, int n, int k0, int k1) {
        const bool left = i < 0;
        const int k = left ? -i : i - (n - 1), kind = left ? k0 : k1;
        const int per = left ? (n - 1) - k : k, ext = left ? 0 : n - 1, sym = left ? k : (n - 1) - k;
        return kind == LSM_BC_PERIODIC ? per : (kind == LSM_BC_EXTRAPOLATION ? ext : sym);
    };
    auto xsrc = [&](int i) { return (!xmap || (i >= 0 && i < nx)) ? i : bsrc(i, nx, xk0, xk1); };
    auto ysrc = [&](int i) { return (!ymap || (i >= 0 && i < ny)) ? i : bsrc(i, ny, yk0, yk1); };
    // a tile whose own columns all lie in the grid loads its pairs in one access; a partial tile (wave-uniform branch) loads the
    // two nodes of its out-of-range pairs one by one from wherever the ghost layers / the redirect put them
    const bool whole = bx0 + TXN <= nx;
    const int lyg = ysrc(gy < ny + G ? gy : ny + G - 1);
    const int lx0 = xsrc(gx < nx + G ? gx : nx + G - 1), lx1 = xsrc(gx + 1 < nx + G ? gx + 1 : nx + G - 1);
    const long long corner = a.origin - G - (long long)G * sy;
    const unsigned orow = (unsigned)(lyg + G) * (unsigned)sy;
    const unsigned ocole0 = (unsigned)(lx0 + G) + orow, ocole1 = (unsigned)(lx1 + G) + orow;
    const unsigned ocol0 = (unsigned)sizeof(ST) * ocole0, ocol1 = (unsigned)sizeof(ST) * ocole1, ocold0 = 8u * ocole0;
    const int mc = a.mc > 0 ? a.mc : MC;
    const int m0 = a.mb + (int)tbm * mc;
    const int m1 = m0 + mc < a.me ? m0 + mc : a.me;
    const int mlo = a.mredirect[0] ? 0 : -G, mhi = a.mredirect[1] ? nm - 1 : nm + G - 1;      // (stage_tile: StageArgs::mredirect)
    auto clampM = [&](int p) { return p < mlo ? mlo : (p > mhi ? mhi : p); };
    auto plane = [&](int p) { return uniform_ptr(reinterpret_cast<const ST*>(a.psi) + (corner + (long long)clampM(p) * sm)); };
    auto ldpair = [&](const ST* P, double& x, double& y) {
        if (whole) ldg2(P, ocol0, x, y);
        else { x = ldg(P, ocol0); y = ldg(P, ocol1); }
    };

    int hl[HPT];
    unsigned hg[HPT];
    bool hv[HPT];
#pragma unroll
    for (int h = 0; h < HPT; ++h) {
        const int e = tid + h * NT;
        hv[h] = e < NH;
        int lx, ly;
        if (e < NHX) {
            const int r = e % (2 * G);
            ly = e / (2 * G) + G;
            lx = r < G ? r : r + TXN;
        } else {
            const int e2 = e - NHX;
            const int row = e2 / WY;
            lx = G + e2 % WY;
            ly = row < G ? row : row + TY;
        }
        int X = bx0 - G + lx;
        X = xsrc(X > nx + G - 1 ? nx + G - 1 : X);
        int Y = by0 - G + ly;
        Y = ysrc(Y > ny + G - 1 ? ny + G - 1 : Y);
        hl[h] = ly * W + lx;
        hg[h] = (unsigned)sizeof(ST) * ((unsigned)(X + G) + (unsigned)(Y + G) * (unsigned)sy);
        if (!hv[h]) hg[h] = LSM_OOB_OFFSET;
    }
    const int lpos = (ty + G) * W + 2 * tx + G;

    double pre_adv[3] = {0, 0, 0}, pre_nm[3] = {0, 0, 0};
    if constexpr (ADV != 0) coeff_prep<NDIM, NDIM, ADV_SCALED, AK>(a.adv, a, 0, 0, pre_adv);
    if constexpr (NM != 0) coeff_prep<NDIM, 1, false, AK>(a.nm, a, 0, 0, pre_nm);

    double zA[2 * G + 1], zB[2 * G + 1];
#pragma unroll
    for (int j = 0; j <= 2 * G; ++j) ldpair(plane(m0 - G + j), zA[j], zB[j]);
    double hp[HPT];
    {
        const ST* P = plane(m0);
#pragma unroll
        for (int h = 0; h < HPT; ++h) hp[h] = ldg(P, hg[h]);
    }
    tile[lpos] = zA[G];
    tile[lpos + 1] = zB[G];
#pragma unroll
    for (int h = 0; h < HPT; ++h)
        if (hv[h]) tile[hl[h]] = hp[h];
    long long po = corner + (long long)m0 * sm;
    const ST* Pnx = plane(m0 + G);
    const ST* Pn = plane(m0);
    const int plast = mhi;
    const bool any_active = __builtin_amdgcn_ballot_w64(active) != 0;
    const int prange = __builtin_amdgcn_readfirstlane(a.base_mode == LSM_BASE_PSI ? 0 : (int)0x80000000u);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int m = m0; m < m1; ++m) {
        Pnx = uniform_ptr(m + 1 + G <= plast ? Pnx + sm : Pnx);
        Pn = uniform_ptr(m + 1 <= plast ? Pn + sm : Pn);
        double nA, nB;
        ldpair(Pnx, nA, nB);
        double hn[HPT];
#pragma unroll
        for (int h = 0; h < HPT; ++h) hn[h] = ldg(Pn, hg[h]);
        // this plane's pointwise operands: consumed after the arithmetic
        NodeOps opA, opB;
        opA.u[0] = pre_adv[0]; opA.u[1] = pre_adv[1]; opA.u[2] = pre_adv[2];
        opA.vnm = pre_nm[0]; opA.bcurv = 0.0; opA.s0 = 0.0; opA.phin = 0.0; opA.out2 = 0.0; opA.have_negs = false;
        opA.negs[0] = opA.negs[1] = opA.negs[2] = 0ull;
        opB = opA;
        if constexpr (EIK == 1) ldg2(uniform_ptr(a.s0 + po), ocold0, opA.s0, opB.s0);
        // always issued; when the base is ψ the descriptor's range is 0: the load returns 0 and touches no memory
        ldg2<ST, 0>(uniform_ptr(reinterpret_cast<const ST*>(a.phin) + po), ocol0, opA.phin, opB.phin, prange);
        LSM_BARRIER();
        double rA = 0.0, rB = 0.0, r2 = 0.0;
        if (any_active) {
            const int rel = m - m0;
            const double* T0 = tile + (rel % NSLOT) * HW + lpos;
            NodeView<NDIM, G, W> nvB{T0 + 1, T0 + 1, T0 + 1, zB, zB[G]};
            node_update<NDIM, ADV, NM, CURV, EIK, G, W, ST, true>(a, nvB, opB, rB, r2);
            NodeView<NDIM, G, W> nvA{T0, T0, T0, zA, zA[G]};
            node_update<NDIM, ADV, NM, CURV, EIK, G, W, ST, true>(a, nvA, opA, rA, r2);
        }
#pragma unroll
        for (int j = 0; j < 2 * G; ++j) { zA[j] = zA[j + 1]; zB[j] = zB[j + 1]; }
        zA[2 * G] = nA; zB[2 * G] = nB;
        const int wslot = (m - m0 + 1) % NSLOT;
        tile[wslot * HW + lpos] = zA[G];
        tile[wslot * HW + lpos + 1] = zB[G];
#pragma unroll
        for (int h = 0; h < HPT; ++h)
            if (hv[h]) tile[wslot * HW + hl[h]] = hn[h];
        if (active) stg2<ST, 0>(uniform_ptr(reinterpret_cast<ST*>(a.out) + po), ocol0, rA, rB);
        po += sm;
    }
}

template <int ADV, int NM, int CURV, int EIK, int TX, int TY, int MC, class ST>
__global__ void __launch_bounds__(TX* TY, LSM_WAVES_PER_EU) stage_kernel2(const StageArgs a) {
    const TileOrder ord(a);
    unsigned tile_id;
    bool tail_tile;
    if (!ord.entry(blockIdx.x % 8u, blockIdx.x / 8u, tile_id, tail_tile)) return;
    stage_tile2<ADV, NM, CURV, EIK, TX, TY, MC, ST>(a, tile_id);
}

template <int NDIM>
struct TileCfg;
template <>
struct TileCfg<1> { static constexpr int TX = 256, TY = 1, MC = 1; };
template <>
#ifndef LSM_TX2
#define LSM_TX2 256
#endif
#ifndef LSM_MC2
#define LSM_MC2 8
#endif
struct TileCfg<2> { static constexpr int TX = LSM_TX2, TY = 1, MC = LSM_MC2; };
#ifndef LSM_TX3
#define LSM_TX3 32
#endif
#ifndef LSM_TY3
#define LSM_TY3 8
#endif
#ifndef LSM_MC3
#define LSM_MC3 64
#endif
template <>
struct TileCfg<3> { static constexpr int TX = LSM_TX3, TY = LSM_TY3, MC = LSM_MC3; };

// The light members of the family in 3-D — a single upwind / NormalMotion / Eikonal term: little arithmetic per byte —
// run 8–14 % faster on a 64×8 tile (512 threads; halo 1.5 instead of 1.7-fold, twice the bytes per row) when the field
// is dense (the narrow band's tile flags describe the standard 32×8 brick).  Measured: tools/terms_ab.sh, DESIGN.md §3.1.
#ifndef LSM_WIDE_TILES
#define LSM_WIDE_TILES 1
#endif
template <int NDIM, int ADV, int NM, int CURV, int EIK>
constexpr bool wide_tile_combo() {
    return LSM_WIDE_TILES && !LSM_STRICT && NDIM == 3 && !CURV && ADV != 2 && (ADV != 0) + (NM != 0) + (EIK != 0) == 1;
}
struct WideTile3 { static constexpr int TX = 64, TY = 8, MC = LSM_MC3; };

template <int NDIM, int ADV, int NM, int CURV, int EIK, class T>
void launch_tiled(const StageArgs& a, hipStream_t s);

// the pair kernels (stage_tile2): which combinations, and the launch
template <int NDIM, int ADV, int NM, int CURV, int EIK>
constexpr bool pair_combo() {
    // (NormalMotion + curvature, config 3's pair, was tried: issue-bound, 0.663 against 0.644 ms per 512³ stage — not taken.  The
    // single curvature term gained 4 % but its three-plane LDS ring gave wrong values at the even nodes behind the first plane of a
    // chunk; not found in the time there was, so curvature stays with stage_tile.)
    return !LSM_STRICT && NDIM == 3 && ADV != 2 && !CURV && (ADV != 0) + (NM != 0) + (EIK != 0) == 1;
}
template <int ADV, int NM, int CURV, int EIK>
bool launch_pairs(const StageArgs& a, hipStream_t s) {
    const int env = a.tune->pairs;                                            // 0 = one node per thread everywhere
    const bool consts = (!ADV || a.adv.kind == LSM_COEFF_CONST) && (!NM || a.nm.kind == LSM_COEFF_CONST) && (!CURV || a.curv.kind == LSM_COEFF_CONST);
    if (!env || a.mask || a.tile_active || a.tile_list || a.mc > 0 || a.out2 || !a.natural || !consts || (a.n[0] & 1) || a.n[0] < 128 || a.me <= a.mb ||
        a.tune->stage_generic)
        return false;
    static_assert(!CURV, "the pair kernels serve the axis-aligned single terms");
    constexpr int TX = 64, TY = 8, MC = LSM_MC3;
    StageArgs b = a;
    b.nb[0] = (a.n[0] + 2 * TX - 1) / (2 * TX);
    b.nb[1] = (a.n[1] + TY - 1) / TY;
    int mc = MC;
    while (mc > 8 && (long long)b.nb[0] * b.nb[1] * ((a.me - a.mb + mc - 1) / mc) < 2048) mc /= 2;
    b.mc = mc;
    b.nb[2] = (a.me - a.mb + mc - 1) / mc;
    b.nbig = 0; b.mc_tail = 0; b.tail_wgs = 0;
    const unsigned ntiles = b.nb[0] * b.nb[1] * b.nb[2];
    const dim3 grid(((ntiles + 7u) / 8u) * 8u), block(TX * TY);
    if (b.f32) hipLaunchKernelGGL((stage_kernel2<ADV, NM, CURV, EIK, TX, TY, MC, float>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((stage_kernel2<ADV, NM, CURV, EIK, TX, TY, MC, double>), grid, block, 0, s, b);
    return true;
}

template <int NDIM, int ADV, int NM, int CURV, int EIK>
void launch_one(const StageArgs& a, hipStream_t s) {
    if constexpr (pair_combo<NDIM, ADV, NM, CURV, EIK>()) {
        if (launch_pairs<ADV, NM, CURV, EIK>(a, s)) return;
    }
    if constexpr (wide_tile_combo<NDIM, ADV, NM, CURV, EIK>()) {
        if (!a.mask && !a.tile_active && !a.tile_list && a.mc <= 0 && a.n[0] >= 64 && !a.out2) {
            launch_tiled<NDIM, ADV, NM, CURV, EIK, WideTile3>(a, s);
            return;
        }
    }
    launch_tiled<NDIM, ADV, NM, CURV, EIK, TileCfg<NDIM>>(a, s);
}

template <int NDIM, int ADV, int NM, int CURV, int EIK, class T>
void launch_tiled(const StageArgs& a, hipStream_t s) {
    dim3 block(T::TX * T::TY);
    StageArgs b = a;
    b.nb[0] = (a.n[0] + T::TX - 1) / T::TX;
    b.nb[1] = NDIM == 3 ? (a.n[1] + T::TY - 1) / T::TY : 1;
    if (NDIM >= 2 && a.me <= a.mb) return;
    int mc = a.mc > 0 ? a.mc : T::MC;
    if (a.mc <= 0 && NDIM == 3 && a.tune->stage_mc > 0) mc = a.tune->stage_mc;       // planes per march chunk
    if (a.mc <= 0 && NDIM == 2 && a.tune->stage_mc2 > 0) mc = a.tune->stage_mc2;     // rows per march chunk in 2-D
    // small grids: a workgroup marching 64 planes leaves most of the 256 CUs idle (48^3 = 12 workgroups, a serial walk
    // of 48 planes each).  Shorter chunks — down to 8 planes — until there are ~8 workgroups per CU; each chunk pays its
    // 2G+1 planes of prologue, which is why large grids keep the long march.
    if (a.mc <= 0 && NDIM == 3 && !a.mask && !(a.tune->stage_mc > 0))
        while (mc > 8 && (long long)b.nb[0] * b.nb[1] * ((a.me - a.mb + mc - 1) / mc) < 2048) mc /= 2;
    b.mc = mc;
    b.nb[2] = NDIM >= 2 ? (a.me - a.mb + mc - 1) / mc : 1;
    b.nbig = 0;
    b.mc_tail = 0;
    // graded tail: a launch ends with every workgroup slot finishing its last chunk at a different moment — on average half a
    // chunk's duration of the whole chip is lost.  Cut the last layer into short chunks and run them last.
    if (NDIM == 3 && a.mc <= 0 && !a.mask && !a.tile_list && !a.tile_active && b.nb[2] >= 4) {
        const int tail_env = a.tune->stage_tail;
        if (tail_env > 0 && tail_env < mc) {
            b.nbig = b.nb[2] - 1;
            b.mc_tail = tail_env;
            const int left = (a.me - a.mb) - (int)b.nbig * mc;
            b.nb[2] = b.nbig + (unsigned)((left + tail_env - 1) / tail_env);
        }
    }
    const unsigned ntiles = b.tile_list ? b.ntile_list : b.nb[0] * b.nb[1] * b.nb[2];
    if (ntiles == 0) return;
    dim3 grid(((ntiles + 7u) / 8u) * 8u, 1, 1);
    b.tail_wgs = 0;
    if (b.mc_tail > 0) {
        const unsigned nbigt = b.nb[0] * b.nb[1] * b.nbig;
        grid.x = 8u * ((nbigt + 7u) / 8u + (ntiles - nbigt + 7u) / 8u);
        const int dyn_env = a.tune->stage_tail_dyn;
        if (dyn_env > 0 && a.tail_ring && a.tail_slot_host && nbigt % 8u == 0) {
            const unsigned ntail = ntiles - nbigt;
            b.tail_wgs = ((ntail + ntail * (unsigned)dyn_env / 100u) + 7u) / 8u * 8u;      // dyn_env % spare workgroups
            b.tail_ctr = a.tail_ring + (*a.tail_slot_host)++ % LSM_TAIL_SLOTS;
            grid.x = nbigt + b.tail_wgs;
        }
    }
    // plain variant (see plane_tab): dense field, one output, terms in slot order, constant speed / curvature
    // coefficients, and a catalogued advection coefficient
#if !LSM_STRICT
    int ak = -1;
    const bool plain = !b.out2 && b.natural && (!NM || b.nm.kind == LSM_COEFF_CONST) && (!CURV || b.curv.kind == LSM_COEFF_CONST) &&
                       !b.tune->stage_generic;
    const bool masked = b.mask != nullptr;          // narrow band: the plain variants exist with and without the band mask
    if (plain) ak = ADV ? b.adv.kind : (int)LSM_COEFF_CONST;
    if (ak == LSM_COEFF_FIELD) ak = -1;
#endif
#define LSM_LAUNCH1(STT, AKK, MK) hipLaunchKernelGGL((stage_kernel<NDIM, ADV, NM, CURV, EIK, T::TX, T::TY, T::MC, STT, AKK, MK>), grid, block, 0, s, b)
#if LSM_STRICT
#define LSM_LAUNCH(STT, AKK) LSM_LAUNCH1(STT, AKK, false)
#else
#define LSM_LAUNCH(STT, AKK) do { if ((AKK) >= 0 && masked) LSM_LAUNCH1(STT, AKK, ((AKK) >= 0)); else LSM_LAUNCH1(STT, AKK, false); } while (0)
#endif
#if LSM_STRICT
    if (b.f32) LSM_LAUNCH(float, -1); else LSM_LAUNCH(double, -1);
#else
    if (ak == LSM_COEFF_CONST) { if (b.f32) LSM_LAUNCH(float, LSM_COEFF_CONST); else LSM_LAUNCH(double, LSM_COEFF_CONST); }
    else if (ADV && ak == LSM_COEFF_ROTATION) { if (b.f32) LSM_LAUNCH(float, (ADV ? (int)LSM_COEFF_ROTATION : 0)); else LSM_LAUNCH(double, (ADV ? (int)LSM_COEFF_ROTATION : 0)); }
    else if (ADV && ak == LSM_COEFF_SEPARABLE) { if (b.f32) LSM_LAUNCH(float, (ADV ? (int)LSM_COEFF_SEPARABLE : 0)); else LSM_LAUNCH(double, (ADV ? (int)LSM_COEFF_SEPARABLE : 0)); }
    else { if (b.f32) LSM_LAUNCH(float, -1); else LSM_LAUNCH(double, -1); }
#endif
#undef LSM_LAUNCH
#undef LSM_LAUNCH1
}

// the instantiated fused combinations (keep in sync with combo_available in lsm_api.hip)
#define LSM_FOR_EACH_COMBO(X) \
    X(1, 0, 0, 0)             \
    X(2, 0, 0, 0)             \
    X(0, 1, 0, 0)             \
    X(0, 0, 1, 0)             \
    X(0, 0, 0, 1)             \
    X(0, 0, 0, 2)             \
    X(2, 0, 0, 2)             \
    X(2, 0, 0, 1)             \
    X(0, 1, 1, 0)             \
    X(2, 0, 1, 0)             \
    X(2, 1, 0, 0)

template <int NDIM>
int launch_ndim(const Combo& c, const StageArgs& a, hipStream_t s) {
#if !LSM_STRICT
    // narrow band on a compact piece list: one lane per band node (stage_brick.h) where that kernel applies
    if constexpr (NDIM == 3) {
        if (a.mask && a.brick_list && launch_stage_brick(c, a, s) == 0) return 0;
    }
#endif
#define LSM_X(ADV, NM, CURV, EIK)                                         \
    if (c.adv == ADV && c.nm == NM && c.curv == CURV && c.eik == EIK) { \
        launch_one<NDIM, ADV, NM, CURV, EIK>(a, s);                       \
        return 0;                                                         \
    }
    LSM_FOR_EACH_COMBO(LSM_X)
#undef LSM_X
    return -1;
}

}  // namespace LSM_NS
}  // namespace lsm
