"""The reference's own interpolation tests (test/test-interpolation.jl) restated on tests/_reinit_ref.py — the literal
restatement of the piecewise Bernstein interpolant (src/interpolation.jl, src/bernstein.jl) that the device
`reinitialize!` is checked against.  They pin that restatement: polynomial reproduction, gradient and Hessian of the
patches, the least-squares (even order) fit, the convex-hull emptiness test, and the O(h^{k+1}) convergence."""
import math

import numpy as np
import pytest

from _reinit_ref import ReinitRef


def _field(orc, f, n, lc, hc, bcspec, order):
    nd = len(n)
    g = orc.Grid(lc, hc, n)
    vals = g.sample(f)
    bc = orc.make_bc(bcspec, nd)
    getphi = lambda J: orc.get(g, bc, vals, J)          # ϕ[J] with ghost resolution, as the stencils read it
    return ReinitRef(getphi, n, lc, hc, order=order, cells=[])


def _eval(r, x):
    return r.vgh(r.cell_of(x), np.asarray(x, dtype=float))


def test_mesh_interpolation_2d(orc):
    """test/test-interpolation.jl:36-52,78-117: cubic patches reproduce a quadratic with its gradient and Hessian"""
    f = lambda x, y: x ** 2 + 2 * y ** 2 - 0.5
    r = _field(orc, f, (21, 21), (-1.0, -1.0), (1.0, 1.0), ("extrapolation", 2), 3)
    x = (0.15, -0.25)
    val, g, H = _eval(r, x)
    assert abs(val - f(*x)) < 1e-12
    assert np.abs(g - np.array([2 * x[0], 4 * x[1]])).max() < 1e-12
    assert np.abs(H - np.array([[2.0, 0.0], [0.0, 4.0]])).max() < 1e-10


def test_least_squares_approximation_k2(orc):
    """test/test-interpolation.jl:54-64: order 2 fits a stencil one node larger in the least-squares sense"""
    f = lambda x, y: x ** 2 + 2 * y ** 2 - 0.5
    r = _field(orc, f, (21, 21), (-1.0, -1.0), (1.0, 1.0), ("extrapolation", 2), 2)
    assert r.nv == 4 and r.mat.shape == (3, 4)
    x = (0.15, -0.25)
    val, g, _ = _eval(r, x)
    assert abs(val - f(*x)) < 1e-12
    assert np.abs(g - np.array([2 * 0.15, 4 * (-0.25)])).max() < 1e-12


def test_mesh_interpolation_3d(orc):
    """test/test-interpolation.jl:66-80"""
    f = lambda x, y, z: x ** 2 + y ** 2 + z ** 2 - 0.5
    r = _field(orc, f, (11, 11, 11), (-1.0,) * 3, (1.0,) * 3, ("extrapolation", 2), 3)
    x = (0.1, -0.2, 0.3)
    val, g, _ = _eval(r, x)
    assert abs(val - f(*x)) < 1e-12
    assert np.abs(g - 2 * np.array(x)).max() < 1e-12


def test_convex_hull_and_proven_empty(orc):
    """test/test-interpolation.jl:82-105: extrema of the Bernstein coefficients bound the patch (0-based cells)"""
    r = _field(orc, lambda x, y: x + 0 * y, (20, 20), (-1.0, -1.0), (1.0, 1.0), ("extrapolation", 2), 3)
    c = r.coeffs((0, 0))
    assert c.max() < 0 and c.min() * c.max() > 0            # no surface in the cell (fully inside)
    c = r.coeffs((18, 18))                                   # the last cell (the reference indexes node 20,20 -> clamped patch)
    assert c.min() > 0
    c = r.coeffs((9, 3))                                     # the cell the interface x = 0 crosses
    assert c.min() < 0 < c.max()


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_interpolation_h_convergence(orc, k):
    """test/test-interpolation.jl:196-228: L∞ error of the order-k interpolant is O(h^{k+1}) (orders ≥ k + 0.5) on
    N = 20, 40, 80 (the reference also runs N = 160; the orders are already resolved), ExtrapolationBC(k)."""
    f = lambda x, y: np.sin(np.pi * x) * np.cos(np.pi * y)
    pts = [(x, y) for x in np.linspace(-0.95, 0.95, 20) for y in np.linspace(-0.95, 0.95, 20)]
    Ns = [20, 40, 80]
    errs = []
    for N in Ns:
        r = _field(orc, f, (N, N), (-1.0, -1.0), (1.0, 1.0), ("extrapolation", k), k)
        errs.append(max(abs(_eval(r, p)[0] - f(*p)) for p in pts))
    orders = [math.log(errs[i] / errs[i + 1]) / math.log(Ns[i + 1] / Ns[i]) for i in range(len(Ns) - 1)]
    assert all(o >= k + 0.5 for o in orders), (errs, orders)
