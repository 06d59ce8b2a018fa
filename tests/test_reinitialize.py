"""reinitialize!.  CPU part: the restatement tests/_reinit_ref.py against the reference's own
tests (test/test-reinitializer.jl:70-132).  GPU part: lsm_reinitialize against the restatement, and the reference's
tests through the host API."""
import itertools
import math
import warnings

import numpy as np
import pytest


def _dense_getter(vals, P):
    """ϕ[J] for any J under ExtrapolationBC(P) on every face (_getindexbc, src/meshfield.jl:248-260)"""
    n, N = vals.shape, vals.ndim

    def w(j, k):
        r = 1.0
        for m in range(P + 1):
            if m != j:
                r *= (-k - m) / (j - m)
        return r

    def get(I, dim=None):
        dim = N if dim is None else dim
        if dim == 0:
            return float(vals[I])
        d = dim - 1
        if 0 <= I[d] < n[d]:
            return get(I, dim - 1)
        left = I[d] < 0
        k = -I[d] if left else I[d] - (n[d] - 1)
        b, s = (0, 1) if left else (n[d] - 1, -1)
        return sum(w(j, k) * get(I[:d] + (b + s * j,) + I[d + 1:], dim - 1) for j in range(P + 1))
    return get


def _field(n, f):
    ax = [np.linspace(-1.0, 1.0, k) for k in n]
    X = np.meshgrid(*ax, indexing="ij")
    return np.asfortranarray(f(X))


# ----------------------------------------------------------------------------- restatement vs the reference's tests

def test_ref_2d_circle_reaches_the_solver_tolerance():
    """test/test-reinitializer.jl:71-87: 100² grid, ϕ = x²+y²-0.25, max error < 2 sqrt(eps)."""
    from _reinit_ref import ReinitRef
    n = (100, 100)
    phi = _field(n, lambda X: X[0] ** 2 + X[1] ** 2 - 0.25)
    R = ReinitRef(_dense_getter(phi, 3), n, (-1, -1), (1, 1))
    assert len(R.pts) > 500
    nodes = [I for k, I in enumerate(itertools.product(range(100), range(100))) if k % 5 == 0]
    out, nfail = R.reinitialize(nodes)
    assert nfail == 0
    err = max(abs(v - (math.hypot(*R.node(I)) - 0.5)) for I, v in out.items())
    assert err < 2 * math.sqrt(np.finfo(float).eps)


def test_ref_3d_sphere_and_h_convergence():
    """test/test-reinitializer.jl:89-100 (coarser: 17³) and :103-132 (orders 2 and 3, N = 20, 40)."""
    from _reinit_ref import ReinitRef
    n = (17, 17, 17)
    phi = _field(n, lambda X: X[0] ** 2 + X[1] ** 2 + X[2] ** 2 - 0.45 ** 2)
    R = ReinitRef(_dense_getter(phi, 3), n, (-1,) * 3, (1,) * 3)
    nodes = [I for k, I in enumerate(itertools.product(*[range(17)] * 3)) if k % 37 == 0]
    out, _ = R.reinitialize(nodes)
    assert max(abs(v - (np.linalg.norm(R.node(I)) - 0.45)) for I, v in out.items()) < 5e-3
    for k in (2, 3):
        errs = []
        for N in (20, 40):
            phi = _field((N, N), lambda X: np.hypot(X[0], X[1]) - 0.5)
            R = ReinitRef(_dense_getter(phi, k), (N, N), (-1, -1), (1, 1), order=k, upsample=10, xtol=1e-14, ftol=1e-14)
            nodes = [I for q, I in enumerate(itertools.product(range(N), range(N))) if q % 3 == 0]
            out, _ = R.reinitialize(nodes)
            errs.append(max(abs(v - (math.hypot(*R.node(I)) - 0.5)) for I, v in out.items()))
        assert math.log(errs[0] / errs[1]) / math.log(2) >= k + 0.5


# ----------------------------------------------------------------------------- device vs restatement / reference tests

@pytest.fixture(scope="module")
def lsm():
    import lsm_amd
    return lsm_amd


def _device_field(lsm, phi, grid, bc, band_layers=None):
    ic = lsm.MeshField(phi, grid)
    if band_layers is not None:
        ic = lsm.NarrowBandMeshField(ic, nlayers=band_layers)
    return lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=ic, bc=bc)


@pytest.mark.gpu
@pytest.mark.parametrize("n,order,upsample", [((40, 36), 3, 2), ((33, 30), 2, 3), ((30, 41), 5, 2), ((14, 12, 13), 3, 2)])
def test_gpu_dense_matches_restatement(lsm, n, order, upsample):
    """Same samples-to-seed-to-closest-point pipeline with tight tolerances: the two implementations must agree to
    round-off amplification (1e-10), not just to the solver tolerance."""
    from _reinit_ref import ReinitRef
    nd = len(n)
    ctr = (0.13, -0.08, 0.05)[:nd]
    phi = _field(n, lambda X: sum((X[d] - ctr[d]) ** 2 for d in range(nd)) - 0.5 ** 2)
    grid = lsm.CartesianGrid((-1.0,) * nd, (1.0,) * nd, n)
    eq = _device_field(lsm, phi, grid, lsm.ExtrapolationBC(3))
    st = eq.current_state()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lsm.reinitialize_(st, order=order, upsample=upsample, xtol=1e-13, ftol=1e-13)
    got = st.values()
    R = ReinitRef(_dense_getter(phi, 3), n, (-1.0,) * nd, (1.0,) * nd, order=order, upsample=upsample, xtol=1e-13, ftol=1e-13)
    nodes = [I for k, I in enumerate(itertools.product(*[range(k) for k in n])) if k % (3 if nd == 2 else 11) == 0]
    want, nfail = R.reinitialize(nodes)
    assert nfail == 0
    assert max(abs(got[I] - v) for I, v in want.items()) < 1e-10


@pytest.mark.gpu
def test_gpu_reference_tests_dense(lsm):
    """test/test-reinitializer.jl:71-100 through the host API: 2-D 100² (error < 2 sqrt(eps), volume preserved) and
    3-D 31³ with upsample 4 (error < 5e-3)."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (100, 100))
    eq = _device_field(lsm, _field((100, 100), lambda X: X[0] ** 2 + X[1] ** 2 - 0.25), grid, lsm.ExtrapolationBC(3))
    assert abs(lsm.volume(eq) - math.pi / 4) < 1e-2
    lsm.reinitialize_(eq)
    X = np.meshgrid(*grid.coords(), indexing="ij")
    assert np.abs(eq.current_state().values() - (np.hypot(X[0], X[1]) - 0.5)).max() < 2 * math.sqrt(np.finfo(float).eps)
    assert abs(lsm.volume(eq) - math.pi / 4) < 1e-2
    g3 = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (31, 31, 31))
    e3 = _device_field(lsm, _field((31, 31, 31), lambda X: X[0] ** 2 + X[1] ** 2 + X[2] ** 2 - 0.45 ** 2), g3, lsm.ExtrapolationBC(3))
    lsm.reinitialize_(e3, upsample=4)
    X = np.meshgrid(*g3.coords(), indexing="ij")
    assert np.abs(e3.current_state().values() - (np.sqrt(X[0] ** 2 + X[1] ** 2 + X[2] ** 2) - 0.45)).max() < 5e-3


@pytest.mark.gpu
def test_gpu_h_convergence(lsm):
    """test/test-reinitializer.jl:103-132: order k interpolation gives O(h^(k+1)) signed distances (k = 2, 3, 4)."""
    for k in (2, 3, 4):
        errs = []
        for N in (20, 40, 80):
            grid = lsm.CartesianGrid((-1, -1), (1, 1), (N, N))
            eq = _device_field(lsm, _field((N, N), lambda X: np.hypot(X[0], X[1]) - 0.5), grid, lsm.ExtrapolationBC(k))
            lsm.reinitialize_(eq, order=k, upsample=10, xtol=1e-14, ftol=1e-14)
            X = np.meshgrid(*grid.coords(), indexing="ij")
            errs.append(np.abs(eq.current_state().values() - (np.hypot(X[0], X[1]) - 0.5)).max())
        orders = [math.log(errs[i] / errs[i + 1]) / math.log(2) for i in range(2)]
        assert all(o >= k + 0.5 for o in orders), (k, errs, orders)


@pytest.mark.gpu
def test_gpu_narrow_band_reinitialize(lsm):
    """Band fields: only band nodes are rewritten, from samples of the active cells; values agree with the dense
    reinitialisation of the same field where both are defined, and the band stays a signed distance."""
    n = (60, 56)
    grid = lsm.CartesianGrid((-1, -1), (1, 1), n)
    phi = _field(n, lambda X: (X[0] - 0.1) ** 2 + X[1] ** 2 - 0.3)          # not a distance function
    dense = _device_field(lsm, phi, grid, lsm.ExtrapolationBC(2))
    band = _device_field(lsm, phi, grid, lsm.ExtrapolationBC(2), band_layers=4)
    lsm.reinitialize_(dense, xtol=1e-13, ftol=1e-13)
    st = band.current_state()
    before = st.values().copy()
    lsm.reinitialize_(st, xtol=1e-13, ftol=1e-13)
    m = st.active_mask()
    v, w = st.values(), dense.current_state().values()
    assert m.sum() > 300 and np.array_equal(np.isnan(v), ~m)
    assert np.abs(v[m] - w[m]).max() < 1e-9
    X = np.meshgrid(*grid.coords(), indexing="ij")
    exact = np.hypot(X[0] - 0.1, X[1]) - math.sqrt(0.3)
    assert np.abs(v[m] - exact[m]).max() < 1e-4
    assert np.abs(before[m] - exact[m]).max() > 1e-2
