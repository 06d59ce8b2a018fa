/*
 * lsm_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A literal restatement in C of the per-timestep grid-update path of LevelSetMethods.jl
 * (reference @ /root/reference, v0.2.0): same loop structure (term-outer / node-inner sweeps,
 * separate copy/combine sweeps), same operation order, true divisions, no FMA contraction
 * (build with -ffp-contract=off).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library — as the checker / the timed CPU baseline, never as a
 * fallback of the product path.
 *
 * Parity pinning: the reference ships no golden vectors and no Julia runtime
 * exists in the build container, so this oracle is pinned by the reference's own analytic tests
 * restated in tests/test_oracle_reference_tests.py (test/test-derivatives.jl, test-levelsetterms.jl,
 * test-timestepping.jl, test-levelsetequation.jl:26-119, test-meshfield.jl:44-125) and the
 * 792-step CFL known answer of docs/src/time-integrators.md:92-94.
 *
 * Indices are 0-based here (Julia's I-1); every formula cites the reference line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/lsm.h"

static int g_threads = 1; /* the reference's hot path is single-threaded (src/timestepping.jl:101-202) */
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Julia's min/max propagate NaN (Base.min(x::Float64,y::Float64)). */
static inline double jl_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (b < a ? b : a); }
static inline double jl_max(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (b > a ? b : a); }

/* A field as the oracle sees it: either the reference's dense Array (ghosts resolved on every
 * read through the _getindexbc recursion) or a padded array whose ghosts were materialised. */
typedef struct F {
    const LsmGrid* g;
    const LsmBc (*bc)[2];
    int has_bc;
    const double* v;
    int padded;
    LsmLayout lay;
    int64_t off[3];   /* global index of local interior (0,0,0) (slab) */
    int64_t nloc[3];  /* local interior extent */
} F;

/* meshsize(g, dim) — src/meshes.jl:110 */
static inline double meshsize(const LsmGrid* g, int d) { return (g->hc[d] - g->lc[d]) / (double)(g->n[d] - 1); }
static inline double min_meshsize(const LsmGrid* g) {
    double m = meshsize(g, 0);
    for (int d = 1; d < g->ndim; ++d) m = jl_min(m, meshsize(g, d));
    return m;
}
/* _getnode — src/meshes.jl:114-117: lc .+ (I .- 1) .* h  (0-based i == I-1) */
static inline double node_coord(const LsmGrid* g, int d, int64_t iglobal) {
    double h = meshsize(g, d);
    return g->lc[d] + (double)iglobal * h;
}

static inline int64_t dense_index(const LsmGrid* g, const int64_t I[3]) {
    return I[0] + g->n[0] * (I[1] + g->n[1] * I[2]);
}

/* _lagrange_extrap_weight — src/boundaryconditions.jl:90-97 */
static double lagrange_w(int j, int k, int P) {
    double w = 1.0;
    for (int m = 0; m <= P; ++m) {
        if (m == j) continue;
        w *= (double)(-k - m) / (double)(j - m);
    }
    return w;
}

/* _getindexbc(ϕ, I, Val(dim)) — src/meshfield.jl:248-260, with bc_stencil
 * (src/boundaryconditions.jl:132-153) and _wrap_index_periodic (:107-119) inlined. */
static double getindexbc(const F* f, const int64_t I[3], int dim /* number of dims still to check */) {
    if (dim == 0) return f->v[dense_index(f->g, I)];
    int d = dim - 1;
    int64_t n = f->g->n[d];
    if (I[d] >= 0 && I[d] < n) return getindexbc(f, I, dim - 1);
    int left = I[d] < 0;
    const LsmBc* bc = &f->bc[d][left ? 0 : 1];
    int64_t J[3] = {I[0], I[1], I[2]};
    double acc = 0.0;
    if (bc->kind == LSM_BC_PERIODIC) {
        /* i < first: last - (first - i); i > last: first + (i - last)  — period n-1 */
        J[d] = left ? (n - 1) - (0 - I[d]) : 0 + (I[d] - (n - 1));
        acc += 1.0 * getindexbc(f, J, dim - 1);
    } else if (bc->kind == LSM_BC_EXTRAPOLATION) {
        int k = (int)(left ? (0 - I[d]) : (I[d] - (n - 1)));
        int64_t b = left ? 0 : n - 1;
        int dir = left ? 1 : -1;
        int P = bc->degree;
        for (int j = 0; j <= P; ++j) {
            J[d] = b + dir * j;
            acc += lagrange_w(j, k, P) * getindexbc(f, J, dim - 1);
        }
    } else { /* LSM_BC_SYMMETRY: mirror about the boundary node */
        int64_t k = left ? (0 - I[d]) : (I[d] - (n - 1));
        int64_t b = left ? 0 : n - 1;
        int dir = left ? 1 : -1;
        J[d] = b + dir * k;
        acc += 1.0 * getindexbc(f, J, dim - 1);
    }
    return acc;
}

/* getindex(ϕ::MeshField, I) — src/meshfield.jl:213-217 */
static inline double fget(const F* f, const int64_t I[3]) {
    if (f->padded) return f->v[f->lay.origin + I[0] + I[1] * f->lay.stride[1] + I[2] * f->lay.stride[2]];
    const LsmGrid* g = f->g;
    if (I[0] >= 0 && I[0] < g->n[0] && I[1] >= 0 && I[1] < g->n[1] && I[2] >= 0 && I[2] < g->n[2])
        return f->v[dense_index(g, I)];
    return getindexbc(f, I, g->ndim);
}
static inline double fget_shift(const F* f, const int64_t I[3], int d, int nb) {
    int64_t J[3] = {I[0], I[1], I[2]};
    J[d] += nb;
    return fget(f, J);
}

/* ---- src/derivatives.jl:28-57 ---- */
static double D0(const F* f, const int64_t I[3], int d) {
    double h = meshsize(f->g, d);
    return (fget_shift(f, I, d, 1) - fget_shift(f, I, d, -1)) / (2 * h);
}
static double Dp(const F* f, const int64_t I[3], int d) {
    double h = meshsize(f->g, d);
    return (fget_shift(f, I, d, 1) - fget(f, I)) / h;
}
static double Dm(const F* f, const int64_t I[3], int d) {
    double h = meshsize(f->g, d);
    return (fget(f, I) - fget_shift(f, I, d, -1)) / h;
}

/* _weno5 — src/derivatives.jl:61-81 */
double orc_weno5_core(double v1, double v2, double v3, double v4, double v5) {
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

/* weno5⁻ / weno5⁺ — src/derivatives.jl:89-121 */
static double weno5m(const F* f, const int64_t I[3], int d) {
    int64_t Im[3] = {I[0], I[1], I[2]}, Imm[3] = {I[0], I[1], I[2]}, Ip[3] = {I[0], I[1], I[2]}, Ipp[3] = {I[0], I[1], I[2]};
    Im[d] -= 1; Imm[d] -= 2; Ip[d] += 1; Ipp[d] += 2;
    return orc_weno5_core(Dm(f, Imm, d), Dm(f, Im, d), Dm(f, I, d), Dm(f, Ip, d), Dm(f, Ipp, d));
}
static double weno5p(const F* f, const int64_t I[3], int d) {
    int64_t Im[3] = {I[0], I[1], I[2]}, Imm[3] = {I[0], I[1], I[2]}, Ip[3] = {I[0], I[1], I[2]}, Ipp[3] = {I[0], I[1], I[2]};
    Im[d] -= 1; Imm[d] -= 2; Ip[d] += 1; Ipp[d] += 2;
    return orc_weno5_core(Dp(f, Ipp, d), Dp(f, Ip, d), Dp(f, I, d), Dp(f, Im, d), Dp(f, Imm, d));
}

/* ---- src/derivatives.jl:129-175 ---- */
static double D20(const F* f, const int64_t I[3], int d) {
    double h = meshsize(f->g, d);
    return (fget_shift(f, I, d, 1) - 2 * fget(f, I) + fget_shift(f, I, d, -1)) / (h * h);
}
static double D2mixed(const F* f, const int64_t I[3], int d1, int d2) {
    double h = meshsize(f->g, d1);
    int64_t Ip[3] = {I[0], I[1], I[2]}, Im[3] = {I[0], I[1], I[2]};
    Ip[d1] += 1; Im[d1] -= 1;
    return (D0(f, Ip, d2) - D0(f, Im, d2)) / (2 * h);
}
static double D2pp(const F* f, const int64_t I[3], int d) {
    double h = meshsize(f->g, d);
    return (fget(f, I) - 2 * fget_shift(f, I, d, 1) + fget_shift(f, I, d, 2)) / (h * h);
}
static double D2mm(const F* f, const int64_t I[3], int d) {
    double h = meshsize(f->g, d);
    return (fget_shift(f, I, d, -2) - 2 * fget_shift(f, I, d, -1) + fget(f, I)) / (h * h);
}

/* positive / negative / limiter (minmod) — src/levelsetterms.jl:180-187 */
static inline double positive(double x) { return x > 0.0 ? x : 0.0; }
static inline double negative(double x) { return x < 0.0 ? x : 0.0; }
static inline double limiter(double x, double y) {
    if (!(x * y > 0.0)) return 0.0;
    return fabs(x) <= fabs(y) ? x : y;
}

/* time factor of a SEPARABLE coefficient */
static double time_factor(const LsmCoeff* c, double t) {
    return c->time_kind == LSM_TIME_COS ? cos(M_PI * t / c->time_param) : 1.0;
}

/* _eval_field — src/levelsetterms.jl:42-43 (see LsmCoeff in include/lsm.h for the catalogue) */
static void eval_coeff(const LsmCoeff* c, const F* f, const int64_t I[3], double t, int ncomp, double out[3]) {
    const LsmGrid* g = f->g;
    switch (c->kind) {
    case LSM_COEFF_CONST:
        for (int k = 0; k < ncomp; ++k) out[k] = c->value[k];
        break;
    case LSM_COEFF_ROTATION: {
        double x1 = node_coord(g, 0, I[0] + f->off[0]);
        double x2 = node_coord(g, 1, I[1] + f->off[1]);
        double w = c->value[0];
        out[0] = -(w * (x2 - c->value[2]));
        out[1] = w * (x1 - c->value[1]);
        out[2] = 0.0;
        break;
    }
    case LSM_COEFF_SEPARABLE: {
        double gt = time_factor(c, t);
        for (int k = 0; k < ncomp; ++k) {
            const double* T = c->sep[k];
            double p = T[I[0] + f->off[0]];
            if (g->ndim > 1) p = p * T[g->n[0] + I[1] + f->off[1]];
            if (g->ndim > 2) p = p * T[g->n[0] + g->n[1] + I[2] + f->off[2]];
            out[k] = p * gt;
        }
        break;
    }
    default: /* LSM_COEFF_FIELD: f[I], in-grid only */
        for (int k = 0; k < ncomp; ++k) {
            const double* a = (const double*)c->field[k];
            out[k] = f->padded ? a[f->lay.origin + I[0] + I[1] * f->lay.stride[1] + I[2] * f->lay.stride[2]]
                               : a[dense_index(g, I)];
        }
    }
}

/* second-order ENO one-sided derivatives shared by NormalMotion and Eikonal
 * (src/levelsetterms.jl:161-163, 255-257) */
static inline void eno2_pair(const F* f, const int64_t I[3], int d, double* A, double* B) {
    double h = meshsize(f->g, d);
    *A = Dm(f, I, d) + 0.5 * h * limiter(D2mm(f, I, d), D20(f, I, d));
    *B = Dp(f, I, d) - 0.5 * h * limiter(D2pp(f, I, d), D20(f, I, d));
}

/* _compute_∇_norm — src/levelsetterms.jl:252-265 */
static double grad_norm_upwind(double v, const F* f, const int64_t I[3]) {
    double mA = 0.0, mB = 0.0;
    for (int d = 0; d < f->g->ndim; ++d) {
        double A, B;
        eno2_pair(f, I, d, &A, &B);
        double a, b;
        if (v > 0) { a = positive(A); b = negative(B); }
        else       { a = negative(A); b = positive(B); }
        a = a * a; b = b * b;
        if (d == 0) { mA = a; mB = b; } else { mA = mA + a; mB = mB + b; }
    }
    return sqrt(mA + mB);
}

static inline double jl_sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); } /* sign(0)=0, sign(NaN)=NaN */

/* curvature — src/levelsetops.jl:197-244.  gᵀHg follows LinearAlgebra's dot(x, ::Symmetric, y)
 * (upper triangle, column by column); the reference leaves that order to its dependencies, so
 * this value is a tolerance-level (not bit-level) parity point. */
static double curvature(const F* f, const int64_t I[3]) {
    int N = f->g->ndim;
    double gr[3] = {0, 0, 0};
    for (int d = 0; d < N; ++d) gr[d] = D0(f, I, d);
    double nrmsq = gr[0] * gr[0];
    for (int d = 1; d < N; ++d) nrmsq = nrmsq + gr[d] * gr[d];
    if (nrmsq < 2.220446049250313e-16) return 0.0;
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int a = 0; a < N; ++a)
        for (int b = a; b < N; ++b) H[a][b] = (a == b) ? D20(f, I, a) : D2mixed(f, I, a, b); /* upper triangle */
    double lap = H[0][0];
    for (int d = 1; d < N; ++d) lap = lap + H[d][d];
    double r = 0.0;
    for (int j = 0; j < N; ++j) {
        r += gr[j] * (H[j][j] * gr[j]);
        for (int i = 0; i < j; ++i) r += gr[i] * (H[i][j] * gr[j]) + gr[j] * (H[i][j] * gr[i]);
    }
    return (lap * nrmsq - r) / pow(nrmsq, 1.5);
}

/* _compute_term for the four terms — src/levelsetterms.jl:73-82,111-121,156-170,234-248 */
static double compute_term(const LsmTerm* term, const F* f, const F* s0f, const int64_t I[3], double t) {
    int N = f->g->ndim;
    switch (term->kind) {
    case LSM_TERM_ADVECTION: {
        double u[3];
        eval_coeff(&term->coeff, f, I, t, N, u);
        double s = 0.0;
        for (int d = 0; d < N; ++d) {
            double v = u[d];
            double der;
            if (term->scheme == LSM_SCHEME_WENO5) der = v > 0 ? weno5m(f, I, d) : weno5p(f, I, d);
            else                                  der = v > 0 ? Dm(f, I, d) : Dp(f, I, d);
            double c = v * der;
            s = (d == 0) ? c : s + c;
        }
        return s;
    }
    case LSM_TERM_NORMAL_MOTION: {
        double vv[3];
        eval_coeff(&term->coeff, f, I, t, 1, vv);
        double v = vv[0];
        double gp = 0.0, gm = 0.0;
        for (int d = 0; d < N; ++d) {
            double neg, pos;
            eno2_pair(f, I, d, &neg, &pos);
            double a = positive(neg) * positive(neg) + negative(pos) * negative(pos);
            double b = negative(neg) * negative(neg) + positive(pos) * positive(pos);
            if (d == 0) { gp = a; gm = b; } else { gp = gp + a; gm = gm + b; }
        }
        return positive(v) * sqrt(gp) + negative(v) * sqrt(gm);
    }
    case LSM_TERM_CURVATURE: {
        double kappa = curvature(f, I);
        double bb[3];
        eval_coeff(&term->coeff, f, I, t, 1, bb);
        double p2 = 0.0;
        for (int d = 0; d < N; ++d) {
            double g0 = D0(f, I, d);
            p2 = (d == 0) ? g0 * g0 : p2 + g0 * g0;
        }
        return bb[0] * kappa * sqrt(p2);
    }
    default: { /* LSM_TERM_EIKONAL */
        double phiI = fget(f, I);
        if (term->s0 == NULL) {
            double nrm = grad_norm_upwind(jl_sign(phiI), f, I);
            double dx = min_meshsize(f->g);
            double denom = sqrt(phiI * phiI + (nrm * nrm) * (dx * dx));
            double S = denom == 0.0 ? 0.0 : phiI / denom;
            return S * (nrm - 1);
        } else {
            double s0 = fget(s0f, I);
            double nrm = grad_norm_upwind(jl_sign(s0), f, I);
            return s0 * (nrm - 1);
        }
    }
    }
}

/* per-node CFL — src/levelsetterms.jl:90-96,123-127,172-178,250 */
static double compute_cfl_node(const LsmTerm* term, const F* f, const int64_t I[3], double t) {
    int N = f->g->ndim;
    switch (term->kind) {
    case LSM_TERM_ADVECTION: {
        double u[3];
        eval_coeff(&term->coeff, f, I, t, N, u);
        double s = fabs(u[0]) / meshsize(f->g, 0);
        for (int d = 1; d < N; ++d) s = s + fabs(u[d]) / meshsize(f->g, d);
        return 1 / s;
    }
    case LSM_TERM_NORMAL_MOTION: {
        double v[3];
        eval_coeff(&term->coeff, f, I, t, 1, v);
        double s = fabs(v[0]) / meshsize(f->g, 0);
        for (int d = 1; d < N; ++d) s = s + fabs(v[0]) / meshsize(f->g, d);
        return 1 / s;
    }
    case LSM_TERM_CURVATURE: {
        double b[3];
        eval_coeff(&term->coeff, f, I, t, 1, b);
        double dx = min_meshsize(f->g);
        return (dx * dx) / (2 * fabs(b[0]));
    }
    default:
        return min_meshsize(f->g);
    }
}

static void make_field(F* f, const LsmGrid* g, const LsmBc (*bc)[2], int has_bc, const double* v) {
    memset(f, 0, sizeof(*f));
    f->g = g; f->bc = bc; f->has_bc = has_bc; f->v = v; f->padded = 0;
    for (int d = 0; d < 3; ++d) { f->nloc[d] = g->n[d]; f->off[d] = 0; }
}
static void make_padded_field(F* f, const LsmGrid* g, const LsmBc (*bc)[2], const LsmSlab* slab, const LsmLayout* lay,
                              const double* v) {
    memset(f, 0, sizeof(*f));
    f->g = g; f->bc = bc; f->has_bc = 1; f->v = v; f->padded = 1; f->lay = *lay;
    for (int d = 0; d < 3; ++d) { f->nloc[d] = lay->n[d]; f->off[d] = 0; }
    if (slab) f->off[g->ndim - 1] = slab->lo;
}

/* ------------------------------------------------------------------ exported: unit probes */

double orc_get(const LsmGrid* g, const LsmBc bc[3][2], int has_bc, const double* v, const int64_t I[3]) {
    F f; make_field(&f, g, bc, has_bc, v);
    return fget(&f, I);
}

/* which: 0 D⁰, 1 D⁺, 2 D⁻, 3 weno5⁻, 4 weno5⁺, 5 D2⁰, 6 D2(dims), 7 D2⁺⁺, 8 D2⁻⁻ */
double orc_deriv(const LsmGrid* g, const LsmBc bc[3][2], int has_bc, const double* v, int which, const int64_t I[3],
                 int dim, int dim2) {
    F f; make_field(&f, g, bc, has_bc, v);
    switch (which) {
    case 0: return D0(&f, I, dim);
    case 1: return Dp(&f, I, dim);
    case 2: return Dm(&f, I, dim);
    case 3: return weno5m(&f, I, dim);
    case 4: return weno5p(&f, I, dim);
    case 5: return D20(&f, I, dim);
    case 6: return D2mixed(&f, I, dim, dim2);
    case 7: return D2pp(&f, I, dim);
    default: return D2mm(&f, I, dim);
    }
}

double orc_term(const LsmGrid* g, const LsmBc bc[3][2], const double* v, const LsmTerm* term, const int64_t I[3], double t) {
    F f, s0; make_field(&f, g, bc, 1, v);
    make_field(&s0, g, bc, 1, (const double*)term->s0);
    return compute_term(term, &f, &s0, I, t);
}

/* EikonalReinitializationTerm(ϕ₀): S₀ = v / sqrt(v^2 + Δx^2) — src/levelsetterms.jl:217-221 */
void orc_eikonal_sign(const LsmGrid* g, const double* v, double* s0) {
    double dx = min_meshsize(g);
    int64_t C = g->n[0] * g->n[1] * g->n[2];
    for (int64_t i = 0; i < C; ++i) s0[i] = v[i] / sqrt(v[i] * v[i] + dx * dx);
}

/* ------------------------------------------------------------------ volume / perimeter
 * src/levelsetops.jl:27-33 (volume), :139-149 (perimeter), :171-183 (smooth_heaviside / smooth_delta).
 * Julia's sum() is pairwise (blocks of 1024, SIMD inside a block), so the last digits of these sums
 * are order-dependent in the reference itself; the oracle sums pairwise with the same block size,
 * sequentially inside a block.  Known answers: jldoctests at src/levelsetops.jl:14-25,126-137. */
static double smooth_heaviside(double x, double alpha) {
    if (x > alpha) return 1.0;
    if (x < -alpha) return 0.0;
    return 0.5 * (1.0 + x / alpha + 1.0 / M_PI * sin(M_PI * x / alpha));
}
static double smooth_delta(double x, double alpha) { return fabs(x) > alpha ? 0.0 : 0.5 / alpha * (1.0 + cos(M_PI * x / alpha)); }

static double measure_node(int mode, const F* f, int64_t q, double dmin) {
    const LsmGrid* g = f->g;
    int64_t I[3];
    I[0] = q % g->n[0]; I[1] = (q / g->n[0]) % g->n[1]; I[2] = q / (g->n[0] * g->n[1]);
    if (mode == 0) return smooth_heaviside(-f->v[q], dmin);
    double nrm2 = 0.0;
    for (int d = 0; d < g->ndim; ++d) {
        double gd = D0(f, I, d);
        nrm2 = d == 0 ? gd * gd : nrm2 + gd * gd;
    }
    return smooth_delta(fget(f, I), dmin) * sqrt(nrm2);
}
static double pairwise(int mode, const F* f, int64_t lo, int64_t hi, double dmin) {   /* [lo, hi) */
    if (hi - lo <= 1024) {
        double v = measure_node(mode, f, lo, dmin);
        for (int64_t q = lo + 1; q < hi; ++q) v += measure_node(mode, f, q, dmin);
        return v;
    }
    int64_t mid = lo + ((hi - lo - 1) >> 1) + 1;   /* Base.mapreduce_impl: imid = ifirst + (ilast - ifirst) >> 1 (inclusive) */
    return pairwise(mode, f, lo, mid, dmin) + pairwise(mode, f, mid, hi, dmin);
}
/* mode 0: volume, 1: perimeter (bc used for the centred gradient at the border) */
double orc_measure(int mode, const LsmGrid* g, const LsmBc bc[3][2], const double* v) {
    F f; make_field(&f, g, bc, 1, v);
    double vol = meshsize(g, 0);
    for (int d = 1; d < g->ndim; ++d) vol = vol * meshsize(g, d);
    return vol * pairwise(mode, &f, 0, g->n[0] * g->n[1] * g->n[2], min_meshsize(g));
}

/* ------------------------------------------------------------------ curvature / gradient / normal at every node
 * src/levelsetops.jl:197-226 (what: 0 curvature -> out0; 1 gradient, 2 normal -> out0..out[N-1]); dense arrays. */
void orc_geometry(int what, const LsmGrid* g, const LsmBc bc[3][2], const double* v, double* out0, double* out1, double* out2) {
    F f; make_field(&f, g, bc, 1, v);
    int N = g->ndim;
    double* out[3] = {out0, out1, out2};
    for (int64_t i2 = 0; i2 < g->n[2]; ++i2)
        for (int64_t i1 = 0; i1 < g->n[1]; ++i1)
            for (int64_t i0 = 0; i0 < g->n[0]; ++i0) {
                const int64_t I[3] = {i0, i1, i2};
                const int64_t q = i0 + g->n[0] * (i1 + g->n[1] * i2);
                if (what == 0) { out0[q] = curvature(&f, I); continue; }
                double gr[3] = {0, 0, 0}, s = 0.0;
                for (int d = 0; d < N; ++d) { gr[d] = D0(&f, I, d); s = d == 0 ? gr[d] * gr[d] : s + gr[d] * gr[d]; }
                const double nrm = sqrt(s);                       /* norm(::SVector) */
                for (int d = 0; d < N; ++d) out[d][q] = what == 1 ? gr[d] : gr[d] / nrm;
            }
}

/* ------------------------------------------------------------------ extend_along_normals!
 * src/velocityextension.jl:20-67 (sweeps), :78-94 (frozen mask), :96-116 (signed normal components).
 * frozen: NULL (band rule) or a dense 0/1 double array.  F is updated in place. */
void orc_extend_along_normals(const LsmGrid* g, const LsmBc bc[3][2], double* Fv, const double* phi, const double* frozen,
                              int nb_iters, double cfl, double interface_band, double min_norm) {
    int N = g->ndim;
    int64_t C = g->n[0] * g->n[1] * g->n[2];
    F fphi; make_field(&fphi, g, bc, 1, phi);
    double delta = min_meshsize(g);
    double tau = cfl * delta;
    unsigned char* fz = (unsigned char*)malloc((size_t)C);
    double* comp[3];
    for (int d = 0; d < 3; ++d) comp[d] = (double*)calloc((size_t)C, sizeof(double));
    double mn2 = min_norm * min_norm;
    for (int64_t q = 0; q < C; ++q) {
        int64_t I[3] = {q % g->n[0], (q / g->n[0]) % g->n[1], q / (g->n[0] * g->n[1])};
        fz[q] = frozen ? (frozen[q] != 0.0) : (fabs(phi[q]) <= interface_band * delta);
        double gr[3] = {0, 0, 0}, nrm2 = 0.0;
        for (int d = 0; d < N; ++d) { gr[d] = D0(&fphi, I, d); nrm2 = d == 0 ? gr[d] * gr[d] : nrm2 + gr[d] * gr[d]; }
        if (nrm2 <= mn2) continue;
        double invnorm = 1.0 / sqrt(nrm2);
        double S = phi[q] / sqrt(phi[q] * phi[q] + delta * delta);
        for (int d = 0; d < N; ++d) comp[d][q] = S * gr[d] * invnorm;
    }
    double* Fnew = (double*)malloc(sizeof(double) * (size_t)C);
    for (int it = 0; it < nb_iters; ++it) {
        F fF; make_field(&fF, g, bc, 1, Fv);
        for (int64_t q = 0; q < C; ++q) {
            int64_t I[3] = {q % g->n[0], (q / g->n[0]) % g->n[1], q / (g->n[0] * g->n[1])};
            if (fz[q]) { Fnew[q] = Fv[q]; continue; }
            double adv = 0.0;
            for (int d = 0; d < N; ++d) {
                double a = comp[d][q];
                double dF = a > 0 ? Dm(&fF, I, d) : Dp(&fF, I, d);
                adv += a * dF;
            }
            Fnew[q] = Fv[q] - tau * adv;
        }
        memcpy(Fv, Fnew, sizeof(double) * (size_t)C);
    }
    free(Fnew); free(fz);
    for (int d = 0; d < 3; ++d) free(comp[d]);
}

/* ------------------------------------------------------------------ CFL (src/levelsetterms.jl:22-38) */

static double cfl_sweep(const LsmTerm* terms, int nterms, const F* f, double t) {
    /* minimum(terms) do term: _compute_cfl(term, ϕ, t) ... — each term a full-grid sweep, dt=Inf start */
    double best = 0;
    for (int k = 0; k < nterms; ++k) {
        double dt = INFINITY;
        int64_t n0 = f->nloc[0], n1 = f->nloc[1], n2 = f->nloc[2];
        int64_t outer = f->g->ndim == 3 ? n2 : (f->g->ndim == 2 ? n1 : 1);
        double* partial = (double*)malloc(sizeof(double) * (size_t)outer);
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static)
        for (int64_t o = 0; o < outer; ++o) {
            double loc = INFINITY;
            int64_t I[3];
            if (f->g->ndim == 3) {
                I[2] = o;
                for (I[1] = 0; I[1] < n1; ++I[1])
                    for (I[0] = 0; I[0] < n0; ++I[0]) loc = jl_min(loc, compute_cfl_node(&terms[k], f, I, t));
            } else if (f->g->ndim == 2) {
                I[2] = 0; I[1] = o;
                for (I[0] = 0; I[0] < n0; ++I[0]) loc = jl_min(loc, compute_cfl_node(&terms[k], f, I, t));
            } else {
                I[2] = 0; I[1] = 0;
                for (I[0] = 0; I[0] < n0; ++I[0]) loc = jl_min(loc, compute_cfl_node(&terms[k], f, I, t));
            }
            partial[o] = loc;
        }
        for (int64_t o = 0; o < outer; ++o) dt = jl_min(dt, partial[o]);
        free(partial);
        best = (k == 0) ? dt : jl_min(best, dt);
    }
    return best;
}

/* raw minimum; the caller applies `Δt > 0 || throw` (src/levelsetterms.jl:26) */
double orc_compute_cfl(const LsmGrid* g, const LsmBc bc[3][2], const double* v, const LsmTerm* terms, int nterms, double t) {
    F f; make_field(&f, g, bc, 1, v);
    return cfl_sweep(terms, nterms, &f, t);
}

/* ------------------------------------------------------------------ sweeps of _advance! */

/* for I in active_nodeindices(ϕ): dst[I] -= c * _compute_term(term, src, I, t)  [and dst2[I] -= c2 * v] */
static void term_sweep(const LsmTerm* term, const F* src, double t, double c, double* dst, double c2, double* dst2) {
    const LsmGrid* g = src->g;
    F s0; make_field(&s0, g, src->bc, 1, (const double*)term->s0);
    int64_t n0 = g->n[0], n1 = g->n[1], n2 = g->n[2];
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static) collapse(2)
    for (int64_t i2 = 0; i2 < n2; ++i2)
        for (int64_t i1 = 0; i1 < n1; ++i1) {
            int64_t I[3] = {0, i1, i2};
            for (I[0] = 0; I[0] < n0; ++I[0]) {
                double v = compute_term(term, src, &s0, I, t);
                int64_t q = dense_index(g, I);
                dst[q] -= c * v;
                if (dst2) dst2[q] -= c2 * v;
            }
        }
}

static void copy_vals(double* dst, const double* src, int64_t C) { memcpy(dst, src, sizeof(double) * (size_t)C); }

/* integrator: 0 ForwardEuler (src/timestepping.jl:126-137), 1 RK2 (:141-164), 2 RK3 (:168-202) */
void orc_advance(int integrator, const LsmGrid* g, const LsmBc bc[3][2], double* phi, double* buf1, double* buf2,
                 const LsmTerm* terms, int nterms, double tc, double dt) {
    int64_t C = g->n[0] * g->n[1] * g->n[2];
    F fphi, fb1, fb2;
    make_field(&fphi, g, bc, 1, phi);
    make_field(&fb1, g, bc, 1, buf1);
    make_field(&fb2, g, bc, 1, buf2);
    if (integrator == 0) {
        copy_vals(buf1, phi, C);
        for (int k = 0; k < nterms; ++k) term_sweep(&terms[k], &fphi, tc, dt, buf1, 0, NULL);
        copy_vals(phi, buf1, C);
    } else if (integrator == 1) {
        double* pred = buf1; double* corr = buf2;
        copy_vals(pred, phi, C);
        copy_vals(corr, phi, C);
        for (int k = 0; k < nterms; ++k) term_sweep(&terms[k], &fphi, tc, dt, pred, 0.5 * dt, corr);
        for (int k = 0; k < nterms; ++k) term_sweep(&terms[k], &fb1, tc + dt, 0.5 * dt, corr, 0, NULL);
        copy_vals(phi, corr, C);
    } else {
        copy_vals(buf1, phi, C);
        for (int k = 0; k < nterms; ++k) term_sweep(&terms[k], &fphi, tc, dt, buf1, 0, NULL);
        copy_vals(buf2, phi, C);
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static)
        for (int64_t i = 0; i < C; ++i) buf2[i] = 0.75 * phi[i] + 0.25 * buf1[i];
        for (int k = 0; k < nterms; ++k) term_sweep(&terms[k], &fb1, tc + dt, 0.25 * dt, buf2, 0, NULL);
        copy_vals(buf1, phi, C);
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static)
        for (int64_t i = 0; i < C; ++i) buf1[i] = (phi[i] + 2 * buf2[i]) / 3;
        for (int k = 0; k < nterms; ++k) term_sweep(&terms[k], &fb2, tc + 0.5 * dt, (2.0 / 3) * dt, buf1, 0, NULL);
        copy_vals(phi, buf1, C);
    }
}

/* _integrate! without hooks — src/timestepping.jl:101-122.  Returns the number of steps taken,
 * or -(steps+1) when compute_cfl would throw (Δt not > 0); *t_out = final time. */
static double eps_of(double x) { /* Julia eps(x): distance to the next float */
    x = fabs(x);
    return nextafter(x, INFINITY) - x;
}
int64_t orc_integrate(int integrator, double cfl, const LsmGrid* g, const LsmBc bc[3][2], double* phi,
                      const LsmTerm* terms, int nterms, double t0, double tf, double dt_max, int64_t max_steps,
                      double* t_out, double* last_dt) {
    int64_t C = g->n[0] * g->n[1] * g->n[2];
    double* buf1 = (double*)malloc(sizeof(double) * (size_t)C);
    double* buf2 = (double*)malloc(sizeof(double) * (size_t)C);
    copy_vals(buf1, phi, C); copy_vals(buf2, phi, C);
    double tc = t0;
    int64_t steps = 0;
    while (tc <= tf - eps_of(tc)) {
        if (max_steps >= 0 && steps >= max_steps) break;
        double dtc = orc_compute_cfl(g, bc, phi, terms, nterms, tc);
        if (!(dtc > 0)) { free(buf1); free(buf2); *t_out = tc; return -(steps + 1); }
        double dt = jl_min(jl_min(dt_max, cfl * dtc), tf - tc);
        orc_advance(integrator, g, bc, phi, buf1, buf2, terms, nterms, tc, dt);
        tc += dt;
        if (last_dt) *last_dt = dt;
        ++steps;
    }
    if (!(max_steps >= 0 && steps >= max_steps && tc <= tf - eps_of(tc))) tc = tf; /* ls.t = tf */
    *t_out = tc;
    free(buf1); free(buf2);
    return steps;
}

/* ------------------------------------------------------------------ padded-layout variants
 * Used (a) to prove that filling ghosts dimension 1 -> N reproduces the _getindexbc recursion
 * bit for bit, and (b) as the per-rank compute of the world_size-2 gloo tests. */

void orc_layout(const LsmGrid* g, const LsmSlab* slab, LsmLayout* lay) {
    int64_t s = 1;
    lay->origin = 0;
    for (int d = 0; d < 3; ++d) {
        lay->n[d] = g->n[d];
        if (slab && d == g->ndim - 1) lay->n[d] = slab->n;
        lay->g[d] = d < g->ndim ? LSM_GHOST : 0;
        lay->stride[d] = s;
        lay->origin += lay->g[d] * s;
        s *= lay->n[d] + 2 * lay->g[d];
    }
    lay->total = s;
}

void orc_fill_ghosts_padded(const LsmGrid* g, const LsmBc bc[3][2], const LsmSlab* slab, const LsmLayout* lay, double* v,
                            int dim_mask) {
    (void)slab;
    for (int d = 0; d < g->ndim; ++d) {
        if (!((dim_mask >> d) & 1)) continue;
        int64_t lo[3], hi[3];
        for (int e = 0; e < 3; ++e) {
            if (e < d) { lo[e] = -lay->g[e]; hi[e] = lay->n[e] + lay->g[e]; }  /* lower dims: incl. their ghosts */
            else       { lo[e] = 0; hi[e] = lay->n[e]; }                       /* higher dims: interior */
        }
        hi[d] = 1; /* dim d itself: one pass per transverse position */
        int64_t n = lay->n[d];
        for (int side = 0; side < 2; ++side) {
            const LsmBc* b = &bc[d][side];
            if (b->kind == LSM_BC_NONE) continue;
            for (int k = 1; k <= lay->g[d]; ++k) {
                int64_t ig = side == 0 ? -k : n - 1 + k;
                int64_t I[3];
                for (I[2] = lo[2]; I[2] < hi[2]; ++I[2])
                    for (I[1] = lo[1]; I[1] < hi[1]; ++I[1])
                        for (I[0] = lo[0]; I[0] < hi[0]; ++I[0]) {
                            int64_t J[3] = {I[0], I[1], I[2]};
                            double acc = 0.0;
                            if (b->kind == LSM_BC_PERIODIC) {
                                J[d] = side == 0 ? (n - 1) - k : k;
                                acc += 1.0 * v[lay->origin + J[0] + J[1] * lay->stride[1] + J[2] * lay->stride[2]];
                            } else if (b->kind == LSM_BC_EXTRAPOLATION) {
                                int64_t bb = side == 0 ? 0 : n - 1;
                                int dir = side == 0 ? 1 : -1;
                                for (int j = 0; j <= b->degree; ++j) {
                                    J[d] = bb + dir * j;
                                    acc += lagrange_w(j, k, b->degree) *
                                           v[lay->origin + J[0] + J[1] * lay->stride[1] + J[2] * lay->stride[2]];
                                }
                            } else {
                                int64_t bb = side == 0 ? 0 : n - 1;
                                int dir = side == 0 ? 1 : -1;
                                J[d] = bb + dir * k;
                                acc += 1.0 * v[lay->origin + J[0] + J[1] * lay->stride[1] + J[2] * lay->stride[2]];
                            }
                            J[d] = ig;
                            v[lay->origin + J[0] + J[1] * lay->stride[1] + J[2] * lay->stride[2]] = acc;
                        }
            }
        }
    }
}

/* the lsm_stage contract (include/lsm.h) evaluated with the oracle's arithmetic on padded arrays */
void orc_stage_padded(const LsmGrid* g, const LsmBc bc[3][2], const LsmSlab* slab, const LsmLayout* lay,
                      const LsmTerm* terms, int nterms, const double* psi, const double* phin, double* out, double* out2,
                      int base_mode, double cdt, double cdt2, double t) {
    F f; make_padded_field(&f, g, bc, slab, lay, psi);
    F s0[LSM_MAX_TERMS];
    for (int k = 0; k < nterms; ++k) make_padded_field(&s0[k], g, bc, slab, lay, (const double*)terms[k].s0);
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static) collapse(2)
    for (int64_t i2 = 0; i2 < lay->n[2]; ++i2)
        for (int64_t i1 = 0; i1 < lay->n[1]; ++i1) {
            int64_t I[3] = {0, i1, i2};
            for (I[0] = 0; I[0] < lay->n[0]; ++I[0]) {
                int64_t q = lay->origin + I[0] + I[1] * lay->stride[1] + I[2] * lay->stride[2];
                double base;
                switch (base_mode) {
                case LSM_BASE_PSI: base = psi[q]; break;
                case LSM_BASE_RK3_S2: base = 0.75 * phin[q] + 0.25 * psi[q]; break;
                case LSM_BASE_RK3_S3: base = (phin[q] + 2 * psi[q]) / 3; break;
                default: base = phin[q];
                }
                double b2 = psi[q];
                for (int k = 0; k < nterms; ++k) {
                    double v = compute_term(&terms[k], &f, &s0[k], I, t);
                    base -= cdt * v;
                    b2 -= cdt2 * v;
                }
                out[q] = base;
                if (out2) out2[q] = b2;
            }
        }
}

double orc_cfl_padded(const LsmGrid* g, const LsmBc bc[3][2], const LsmSlab* slab, const LsmLayout* lay,
                      const LsmTerm* terms, int nterms, const double* phi, double t) {
    F f; make_padded_field(&f, g, bc, slab, lay, phi);
    return cfl_sweep(terms, nterms, &f, t);
}
