"""TEST-ONLY restatement (numpy + scipy's exact KD-tree, Python loops: small grids only) of the reference's
Newton closest-point reinitialisation:

    reinitialize!            src/reinitializer.jl:12-42
    NewtonSDF, sampling      src/sdf.jl:57-78,186-221
    closest point            src/sdf.jl:85-131,223-249
    piecewise interpolant    src/interpolation.jl:29-151,228-271 (Bernstein patches, src/bernstein.jl:53-94)

`getphi(J)` must return ϕ[J] for any integer index tuple the stencils reach (boundary conditions / band
extrapolation already resolved by the caller).  0-based indices.  Derivatives of a patch are the exact
derivatives of the Bernstein form (the reference differentiates the same polynomial with ForwardDiff)."""
import itertools
import math

import numpy as np
from scipy.spatial import cKDTree

MAX_SEEDS = 10


def interpolation_matrix(order):
    """_interpolation_matrix (src/interpolation.jl:52-63): (order+1) x (stencil_order+1)."""
    so = order if order % 2 == 1 else order + 1
    nc, nv = order + 1, so + 1
    nodes = [i / so for i in range(nv)]
    a, b = (so - 1) / (2 * so), (so + 1) / (2 * so)
    B = lambda i, k, x: math.comb(k, i) * x ** i * (1 - x) ** (k - i)
    V = np.array([[B(j, order, (nodes[i] - a) / (b - a)) for j in range(nc)] for i in range(nv)])
    return np.linalg.pinv(V)


def bernstein_1d(n, t):
    """values, first and second derivatives (w.r.t. t) of the degree-n Bernstein basis at t"""
    B = np.array([math.comb(n, i) * t ** i * (1 - t) ** (n - i) for i in range(n + 1)])

    def lower(m):   # degree-m basis, zero-padded at both ends
        return np.array([0.0] + [math.comb(m, i) * t ** i * (1 - t) ** (m - i) for i in range(m + 1)] + [0.0]) if m >= 0 else np.zeros(2)
    if n >= 1:
        L1 = lower(n - 1)
        dB = np.array([n * (L1[i] - L1[i + 1]) for i in range(n + 1)])
    else:
        dB = np.zeros(1)
    if n >= 2:
        L2 = np.concatenate([[0.0], lower(n - 2), [0.0]])
        d2B = np.array([n * (n - 1) * (L2[i] - 2 * L2[i + 1] + L2[i + 2]) for i in range(n + 1)])
    else:
        d2B = np.zeros(n + 1)
    return B, dB, d2B


class ReinitRef:
    def __init__(self, getphi, n, lc, hc, order=3, upsample=2, maxiters=20, xtol=None, ftol=None, cells=None):
        self.getphi, self.n, self.N = getphi, tuple(n), len(n)
        self.lc = np.array(lc, dtype=float)
        self.h = (np.array(hc, dtype=float) - self.lc) / (np.array(n) - 1)
        self.order, self.upsample, self.maxiters = order, upsample, maxiters
        eps = np.finfo(float).eps
        self.xtol = math.sqrt(eps) if xtol is None else xtol
        self.ftol = math.sqrt(eps) if ftol is None else ftol
        self.mat = interpolation_matrix(order)
        self.nv = self.mat.shape[1]
        self.off = -((self.nv - 1) - 1) // 2
        self._cache = {}
        cells = list(itertools.product(*[range(k - 1) for k in self.n])) if cells is None else list(cells)
        self.pts = self._sample(cells)
        self.tree = cKDTree(np.array(self.pts)) if self.pts else None

    # ---- interpolant
    def coeffs(self, I):
        c = self._cache.get(I)
        if c is None:
            vals = np.empty((self.nv,) * self.N)
            for J in itertools.product(range(self.nv), repeat=self.N):
                vals[J] = self.getphi(tuple(I[d] + self.off + J[d] for d in range(self.N)))
            c = vals
            for d in range(self.N):   # kron(mat, ..., mat): apply mat along every dimension
                c = np.moveaxis(np.tensordot(self.mat, c, axes=([1], [d])), 0, d)
            self._cache[I] = c
        return c

    def cell_of(self, x):   # compute_index (src/meshes.jl:155-167), clamped
        return tuple(int(min(max(math.floor((x[d] - self.lc[d]) / self.h[d]), 0), self.n[d] - 2)) for d in range(self.N))

    def node(self, I):
        return self.lc + np.array(I) * self.h

    def vgh(self, I, x):
        """value, gradient, hessian at x of the patch of cell I"""
        c = self.coeffs(I)
        t = (np.asarray(x) - self.node(I)) / self.h
        bas = [bernstein_1d(self.order, t[d]) for d in range(self.N)]

        def contract(which):
            r = c
            for d in range(self.N - 1, -1, -1):
                r = np.tensordot(r, bas[d][which[d]], axes=([d], [0]))
            return float(r)
        val = contract([0] * self.N)
        g = np.array([contract([1 if e == d else 0 for e in range(self.N)]) / self.h[d] for d in range(self.N)])
        H = np.empty((self.N, self.N))
        for a in range(self.N):
            for b in range(self.N):
                w = [0] * self.N
                if a == b:
                    w[a] = 2
                else:
                    w[a] = w[b] = 1
                H[a, b] = contract(w) / (self.h[a] * self.h[b])
        return val, g, H

    # ---- sampling (src/sdf.jl:186-221)
    def _project(self, x0, safeguard):
        x = np.array(x0, dtype=float)
        for _ in range(self.maxiters):
            val, g, _ = self.vgh(self.cell_of(x), x)
            if abs(val) < self.ftol:
                return x
            g2 = float(g @ g)
            if g2 == 0.0:
                break
            x = x - val * g / g2
            if np.linalg.norm(x - x0) > safeguard:
                break
        return None

    def _sample(self, cells):
        safeguard = float(self.h.max())
        pts = []
        for I in cells:
            c = self.coeffs(I)
            if c.min() * c.max() > 0:     # proven_empty(...; surface = true)
                continue
            lo = self.node(I)
            for xi in itertools.product(range(self.upsample + 1), repeat=self.N):
                xi = xi[::-1]             # Iterators.product: first range fastest (order is irrelevant to the point set)
                x = lo + ((lo + self.h) - lo) * np.array(xi) / self.upsample    # cell.lc .+ (cell.hc .- cell.lc) .* ξ ./ upsample (src/sdf.jl:167)
                pt = self._project(x, safeguard)
                if pt is None or self.cell_of(pt) != I:
                    continue
                pts.append(pt)
        return pts

    # ---- closest point (src/sdf.jl:223-249)
    def _closest_point(self, I, xq, x0, safeguard):
        N = self.N
        _, g0, _ = self.vgh(I, x0)
        g2 = float(g0 @ g0)
        lam = 0.0 if g2 == 0.0 else float((xq - x0) @ g0) / g2
        x = np.array(x0, dtype=float)
        best_x, best_res = x.copy(), math.inf
        reg = math.sqrt(np.finfo(float).eps)
        for _ in range(self.maxiters):
            px, gp, Hp = self.vgh(I, x)
            res = np.concatenate([x - xq + lam * gp, [px]])
            rn = float(np.linalg.norm(res))
            if rn < best_res:
                best_res, best_x = rn, x.copy()
            if abs(px) < self.ftol and rn < self.xtol:
                return x, True
            K = np.zeros((N + 1, N + 1))
            K[:N, :N] = np.eye(N) + lam * Hp
            K[:N, N] = gp
            K[N, :N] = gp
            d = -np.linalg.solve(K + reg * np.eye(N + 1), res)
            dx, dl = d[:N], d[N]
            nd = float(np.linalg.norm(dx))
            alpha = min(1.0, safeguard / nd) if nd > 0 else 1.0
            x, lam = x + alpha * dx, lam + alpha * dl
            if np.linalg.norm(x - x0) > safeguard:
                return best_x, False
        return best_x, False

    def closest_point(self, xq):
        safeguard = 1.5 * float(self.h.max())
        _, idx = self.tree.query(xq)
        seed = self.pts[idx]
        cp, ok = self._closest_point(self.cell_of(seed), xq, seed, safeguard)
        if ok:
            return cp, True
        k = min(MAX_SEEDS, len(self.pts))
        _, idxs = self.tree.query(xq, k=k)
        best_cp, best_d = cp, float(np.linalg.norm(xq - cp))
        for j in np.atleast_1d(idxs):
            if j == idx:
                continue
            cp, ok = self._closest_point(self.cell_of(self.pts[j]), xq, self.pts[j], safeguard)
            if ok:
                return cp, True
            d = float(np.linalg.norm(xq - cp))
            if d < best_d:
                best_cp, best_d = cp, d
        return best_cp, False

    def reinitialize(self, nodes):
        """{I: sign(ϕ[I]) * |x_I - cp|} for the given node indices; second value: number of non-converged nodes"""
        out, nfail = {}, 0
        for I in nodes:
            x = self.node(I)
            cp, ok = self.closest_point(x)
            nfail += 0 if ok else 1
            v = self.getphi(I)
            out[I] = float(np.sign(v)) * float(np.linalg.norm(x - cp))
        return out, nfail
