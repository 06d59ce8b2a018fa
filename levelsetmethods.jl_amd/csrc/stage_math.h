// stage_math.h — per-node arithmetic of the level-set terms for gfx950 (fp64 VALU).
//
// Included twice, from stage_fast.hip (LSM_STRICT=0, built with -ffp-contract=fast) and from
// stage_strict.hip (LSM_STRICT=1, built with -ffp-contract=off):
//
//  STRICT  the reference's operation order verbatim (src/derivatives.jl:28-175,
//          src/levelsetterms.jl:73-265, src/levelsetops.jl:197-244): true IEEE divisions, no
//          contraction.  Used to prove indexing / ghost / stage logic bit for bit against the oracle.
//  FAST    same mathematics reorganised for the fp64 vector pipe, which — not HBM — bounds this
//          kernel (≈230 vector instructions per node-stage of WENO5 advection + Eikonal against 24 bytes):
//            * differences stay undivided (Δ, not Δ/h); WENO5 is homogeneous of degree one, so the
//              1/h is applied once per dimension and the ε floor becomes 1e-99·h² (raised to 1e-75, see below);
//            * the three WENO weights use ONE reciprocal: Σ_k c_k Π_{j≠k}(S_j+ε)² d_k / Σ_k c_k Π_{j≠k}(S_j+ε)²
//              instead of 6 divisions (src/derivatives.jl:73-78);
//            * divisions/sqrt are v_rcp_f64 / v_rsq_f64 seeds + Newton/Goldschmidt FMA steps (≤2 ulp);
//            * the two minmods of an ENO pair as clamps sharing max(w₃,0), min(w₃,0); Godunov sums without selects;
//              the ENO pair of a WENO5 line from that line's own differences.
//          Valid for neighbour differences in ~[1e-35, 1e+35] (beyond that the weight products
//          leave the fp64 range; exactly flat data is handled); within 1e-13·max|ϕ| per stage.
#pragma once
#include <hip/hip_runtime.h>

namespace lsm {
namespace LSM_NS {

#define LSM_DEV __device__ __forceinline__

LSM_DEV double positive(double x) { return x > 0.0 ? x : 0.0; }   // src/levelsetterms.jl:180
LSM_DEV double negative(double x) { return x < 0.0 ? x : 0.0; }   // src/levelsetterms.jl:181

#if LSM_STRICT

LSM_DEV double jl_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (b > a ? b : a); }

// _weno5 — src/derivatives.jl:61-81, verbatim
LSM_DEV double weno5_core(double v1, double v2, double v3, double v4, double v5) {
    double dphi1 = (1.0 / 3) * v1 - (7.0 / 6) * v2 + (11.0 / 6) * v3;
    double dphi2 = -(1.0 / 6) * v2 + (5.0 / 6) * v3 + (1.0 / 3) * v4;
    double dphi3 = (1.0 / 3) * v3 + (5.0 / 6) * v4 - (1.0 / 6) * v5;
    double a1 = v1 - 2 * v2 + v3, b1 = v1 - 4 * v2 + 3 * v3;
    double a2 = v2 - 2 * v3 + v4, b2 = v2 - v4;
    double a3 = v3 - 2 * v4 + v5, b3 = 3 * v3 - 4 * v4 + v5;
    double S1 = (13.0 / 12) * (a1 * a1) + (1.0 / 4) * (b1 * b1);
    double S2 = (13.0 / 12) * (a2 * a2) + (1.0 / 4) * (b2 * b2);
    double S3 = (13.0 / 12) * (a3 * a3) + (1.0 / 4) * (b3 * b3);
    double m = jl_max(jl_max(jl_max(jl_max(v1 * v1, v2 * v2), v3 * v3), v4 * v4), v5 * v5);
    double eps = 1.0e-6 * m + 1.0e-99;
    double t1 = S1 + eps, t2 = S2 + eps, t3 = S3 + eps;
    double al1 = 0.1 / (t1 * t1);
    double al2 = 0.6 / (t2 * t2);
    double al3 = 0.3 / (t3 * t3);
    double w1 = al1 / (al1 + al2 + al3);
    double w2 = al2 / (al1 + al2 + al3);
    double w3 = al3 / (al1 + al2 + al3);
    return w1 * dphi1 + w2 * dphi2 + w3 * dphi3;
}

// limiter (minmod) — src/levelsetterms.jl:184-187
LSM_DEV double limiter(double x, double y) {
    if (!(x * y > 0.0)) return 0.0;
    return __builtin_fabs(x) <= __builtin_fabs(y) ? x : y;
}

// upwind-biased WENO5 derivative from the six values q0..q5 = ϕ[I + s·(-3..2)e] (s = +1 if u>0
// else -1) and hs = s·h:  v_k = (q_k - q_{k-1})/hs reproduces D⁻ (s=+1) / D⁺ (s=-1) of
// src/derivatives.jl:89-121 bit for bit, since (a-b)/(-h) == (b-a)/h in IEEE arithmetic.
LSM_DEV double weno5_upwind(const double q[6], double hs, double /*inv_hs*/, double /*eps_floor*/) {
    return weno5_core((q[1] - q[0]) / hs, (q[2] - q[1]) / hs, (q[3] - q[2]) / hs, (q[4] - q[3]) / hs,
                      (q[5] - q[4]) / hs);
}

// second-order ENO one-sided derivatives (src/levelsetterms.jl:161-163,255-257)
LSM_DEV void eno2_pair(double m2, double m1, double c, double p1, double p2, double h, double h2, double /*inv_h*/,
                       double& A, double& B) {
    double dm = (c - m1) / h;
    double dp = (p1 - c) / h;
    double d20 = (p1 - 2 * c + m1) / h2;
    double d2mm = (m2 - 2 * m1 + c) / h2;
    double d2pp = (c - 2 * p1 + p2) / h2;
    A = dm + 0.5 * h * limiter(d2mm, d20);
    B = dp - 0.5 * h * limiter(d2pp, d20);
}
LSM_DEV double lsm_sqrt(double x) { return __builtin_sqrt(x); }
LSM_DEV double lsm_div(double a, double b) { return a / b; }

#else  // ------------------------------------------------------------------ FAST

// Reciprocal / reciprocal square root: hardware seed (measured on MI355X: 2^-24.4 / 2^-24.2 relative,
// tools/seedacc.hip) + ONE Newton step -> 2.1e-15 / 4.1e-15.  That is ample here: these values only
// scale correction terms that are themselves O(Δϕ), so their error enters a stage as
// c·Δt·|u|/h · 4e-15·|Δϕ| <~ 1e-17·max|ϕ|, four orders below the 1e-13 parity bar.
LSM_DEV double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
LSM_DEV double lsm_div(double a, double b) { return a * fast_rcp(b); }

LSM_DEV double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-0.5 * x * r, r, 0.5);
    return __builtin_fma(r, e, r);
}

// sqrt for x >= 0: v_rsq_f64 seed, one Goldschmidt step and the final residual correction (which
// is itself a Newton step: ~1e-16).  The argument is floored at 1e-300 instead of branching on zero:
// sqrt(0) returns 1e-150 (a select costs ≈4 fp64 issue slots on gfx950, an fmax one).
// sqrt(x) = x·rsqrt(x) for the Godunov norms (x >= 0; 0 -> 0): two operations fewer than lsm_sqrt, 4e-15
LSM_DEV double fast_norm(double x) { return x * fast_rsqrt(__builtin_fmax(x, 1.0e-300)); }

LSM_DEV double lsm_sqrt(double x0) {
    const double x = __builtin_fmax(x0, 1.0e-300);
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r, hh = 0.5 * r;
    double e = __builtin_fma(-hh, g, 0.5);
    g = __builtin_fma(g, e, g);
    hh = __builtin_fma(hh, e, hh);
    double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, hh, g);
}

// Jiang–Shu WENO5 on UNDIVIDED one-sided differences e1..e5 (ordered from the upwind end inward),
// returning h·(reference value), in the second-difference form
//     Σ ω_k dϕ_k = dϕ₂ + (ω₁/3)(A₁-A₂) + (ω₃/6)(A₂-A₃),   A_k = e_k - 2e_{k+1} + e_{k+2},
// (algebraically identical to src/derivatives.jl:63-80 because Σω = 1), with the three weights from
// ONE reciprocal: ω_k ∝ c_k Π_{j≠k} (S_j+ε)².  eps_floor = 1e-99·h², raised to 1e-75: the reference's
// 1e-99 floor only matters for differences below ~1e-49·h; raising it to 1e-75 changes results only for differences
// below ~3e-35 (by less than their own size) and spares the guard against a vanishing denominator.
// With PQ, the second-order ENO pair of the same line (eno2_pair below) is returned as well, in the
// upwind frame of the stencil: P = e3 + ½·minmod(w2, w3), Q = e4 - ½·minmod(w4, w3) — the differences
// it needs are exactly e3, e4, w2, w3, w4.  In the flipped frame (P, Q) = (-B, -A), which the Godunov
// sum max(σA,0)² + min(σB,0)² does not see, so a fused WENO5 + NormalMotion/Eikonal node computes its
// ENO pairs for free.
LSM_DEV double minmod_fast(double x, double y);
template <bool PQ>
LSM_DEV double weno5_undivided_pq(double e1, double e2, double e3, double e4, double e5, double eps_floor, double& P, double& Q) {
    const double w1 = e2 - e1, w2 = e3 - e2, w3 = e4 - e3, w4 = e5 - e4;
    if constexpr (PQ) {
        // minmod(x, w3) = clamp of x to the interval between 0 and w3: the two share max(w3,0), min(w3,0)
        const double hi = __builtin_fmax(w3, 0.0), lo = __builtin_fmin(w3, 0.0);
        P = __builtin_fma(0.5, __builtin_fmax(__builtin_fmin(w2, hi), lo), e3);
        Q = __builtin_fma(-0.5, __builtin_fmax(__builtin_fmin(w4, hi), lo), e4);
    }
    const double A1 = w2 - w1, A2 = w3 - w2, A3 = w4 - w3;
    const double B1 = __builtin_fma(2.0, w2, A1);      // e1 - 4e2 + 3e3
    const double B2 = w2 + w3;                         // -(e2 - e4)
    const double B3 = __builtin_fma(-2.0, w3, A3);     // 3e3 - 4e4 + e5
    const double m = __builtin_fmax(__builtin_fmax(__builtin_fmax(__builtin_fabs(e1), __builtin_fabs(e2)),
                                                   __builtin_fmax(__builtin_fabs(e3), __builtin_fabs(e4))),
                                    __builtin_fabs(e5));
    // S_k + ε scaled by 12/13 (the weights only see ratios): A² + (3/13)·B² + (12/13)·ε — one multiply less per k
    const double eps = __builtin_fma((12.0 / 13) * 1.0e-6 * m, m, __builtin_fmax(eps_floor, 1.0e-75));   // the max is loop-invariant
    const double r1 = __builtin_fma(A1, A1, __builtin_fma((3.0 / 13) * B1, B1, eps));
    const double r2 = __builtin_fma(A2, A2, __builtin_fma((3.0 / 13) * B2, B2, eps));
    const double r3 = __builtin_fma(A3, A3, __builtin_fma((3.0 / 13) * B3, B3, eps));
    const double s1 = r1 * r1, s2 = r2 * r2, s3 = r3 * r3;
    const double W1 = s2 * s3, W2 = s1 * s3, W3 = s1 * s2;   // ∝ α_k / c_k
    const double c1W1 = (0.1 / 3) * W1, c3W3 = (0.3 / 6) * W3;
    // 0.1 W1 + 0.6 W2 + 0.3 W3 > 0: eps_floor >= 1e-75 keeps every r_k >= 1e-75 and the products >= 1e-302
    // (exactly flat data gives 0·(1/den) = 0, as the reference)
    const double den = __builtin_fma(3.0, c1W1, __builtin_fma(6.0, c3W3, 0.6 * W2));
    const double rc = fast_rcp(den);
    const double dphi2 = __builtin_fma(1.0 / 3, w3, __builtin_fma(1.0 / 6, w2, e3));   // (-e2 + 5 e3 + 2 e4)/6 = e3 + w3/3 + w2/6
    const double X = __builtin_fma(c1W1, A1 - A2, c3W3 * (A2 - A3));
    return __builtin_fma(rc, X, dphi2);
}
LSM_DEV double weno5_undivided(double e1, double e2, double e3, double e4, double e5, double eps_floor) {
    double P, Q;
    return weno5_undivided_pq<false>(e1, e2, e3, e4, e5, eps_floor, P, Q);
}

LSM_DEV double weno5_upwind(const double q[6], double /*hs*/, double inv_hs, double eps_floor) {
    return weno5_undivided(q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], eps_floor) * inv_hs;
}

// minmod(x,y) (src/levelsetterms.jl:184-187) without selects
// = median(x, y, 0), four min/max
LSM_DEV double minmod_fast(double x, double y) {
    return __builtin_fmax(__builtin_fmin(x, y), __builtin_fmin(__builtin_fmax(x, y), 0.0));
}

// FAST: A and B are returned UNDIVIDED (h·A, h·B); the caller applies 1/h² to the squares.
LSM_DEV void eno2_pair(double m2, double m1, double c, double p1, double p2, double /*h*/, double /*h2*/, double /*inv_h*/,
                       double& A, double& B) {
    const double dm = c - m1, dp = p1 - c;
    const double s0 = dp - dm;
    const double smm = dm - (m1 - m2);
    const double spp = (p2 - p1) - dp;
    // minmod(x, s0) = clamp of x to the interval between 0 and s0: the pair shares max(s0,0) and min(s0,0) (6 instead of 8 min/max)
    const double hi = __builtin_fmax(s0, 0.0), lo = __builtin_fmin(s0, 0.0);
    A = __builtin_fma(0.5, __builtin_fmax(__builtin_fmin(smm, hi), lo), dm);
    B = __builtin_fma(-0.5, __builtin_fmax(__builtin_fmin(spp, hi), lo), dp);
}
#endif

// Godunov selection shared by Eikonal (_compute_∇_norm, src/levelsetterms.jl:252-265) and, in FAST
// mode, NormalMotion: v>0 ? (positive(A)², negative(B)²) : (negative(A)², positive(B)²)
#if LSM_STRICT
LSM_DEV void godunov_sel(bool vpos, double A, double B, double& a2, double& b2) {
    double a = vpos ? positive(A) : negative(A);
    double b = vpos ? negative(B) : positive(B);
    a2 = a * a;
    b2 = b * b;
}
#else
// multiply the operands by sg = ±1 instead of selecting (negative(A)² == positive(-A)²; a sign-bit XOR costs the same
// issue slot and makes the compiler canonicalise the result before the max/min); A,B undivided, returns inv_h2·(a² + b²)
LSM_DEV double godunov_term(double sg, double A, double B, double inv_h2) {
    const double a = __builtin_fmax(sg * A, 0.0);
    const double b = __builtin_fmin(sg * B, 0.0);
    return __builtin_fma(a, a, b * b) * inv_h2;
}
// the same with the sign fixed.  max/min against 0 through the instruction itself: A and B reach this point from
// another basic block (the WENO variants), where the compiler no longer knows them to be canonical and would put a
// v_max_f64 x,x,x in front of every __builtin_fmax/fmin (one issue slot each, six per node).
LSM_DEV double vmax0(double x) { double r; asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x)); return r; }
LSM_DEV double vmin0(double x) { double r; asm("v_min_f64 %0, %1, 0" : "=v"(r) : "v"(x)); return r; }
LSM_DEV double godunov_pos(double A, double B, double inv_h2) {   // max(A,0)² + min(B,0)²
    const double a = vmax0(A);
    const double b = vmin0(B);
    return __builtin_fma(a, a, b * b) * inv_h2;
}
LSM_DEV double godunov_neg(double A, double B, double inv_h2) {   // min(A,0)² + max(B,0)²
    const double a = vmin0(A);
    const double b = vmax0(B);
    return __builtin_fma(a, a, b * b) * inv_h2;
}
// Σ_d over NDIM dimensions with one sign per wave; equal spacings (the rule): the undivided squares are summed in one
// FMA chain and scaled once (2 instructions fewer in 3-D than scaling per dimension)
template <int NDIM, bool POS>
LSM_DEV double godunov_sum(const double* A, const double* B, const double* inv_h2, bool uniform_h) {
    if (uniform_h) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < NDIM; ++d) {
            const double a = POS ? vmax0(A[d]) : vmin0(A[d]);
            const double b = POS ? vmin0(B[d]) : vmax0(B[d]);
            s = d == 0 ? a * a : __builtin_fma(a, a, s);
            s = __builtin_fma(b, b, s);
        }
        return s * inv_h2[0];
    }
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < NDIM; ++d) s += POS ? godunov_pos(A[d], B[d], inv_h2[d]) : godunov_neg(A[d], B[d], inv_h2[d]);
    return s;
}
#endif

}  // namespace LSM_NS
}  // namespace lsm
