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
@pytest.mark.parametrize("n,order,upsample", [((40, 36), 3, 2), ((33, 30), 2, 3), ((30, 41), 5, 2), ((14, 12, 13), 3, 2), ((90,), 3, 2), ((61,), 4, 3)])
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
    nodes = [I for k, I in enumerate(itertools.product(*[range(k) for k in n])) if k % {1: 1, 2: 3, 3: 11}[nd] == 0]
    want, nfail = R.reinitialize(nodes)
    assert nfail == 0
    assert max(abs(got[I] - v) for I, v in want.items()) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("n,band", [((48, 44), None), ((22, 20, 21), None), ((40, 40, 40), 3)])
def test_nodes_whose_first_solve_fails_take_the_second_pass(lsm, n, band):
    """src/sdf.jl:113-131: the solve starts from the nearest sample; only where it does not converge are further near samples
    collected and tried in order.  An unreachable xtol makes EVERY first solve "fail" (its iterate is at the closest point all the
    same): every node then goes through the second pass — the shell search's five nearest samples, tried in order, the best
    iterate kept — and must end where the ordinary call ends; the warning counts every evaluated node."""
    nd = len(n)
    ctr = (0.07, -0.04, 0.05)[:nd]
    phi = _field(n, lambda X: sum((X[d] - ctr[d]) ** 2 for d in range(nd)) - 0.5 ** 2)
    grid = lsm.CartesianGrid((-1.0,) * nd, (1.0,) * nd, n)
    a = _device_field(lsm, phi, grid, lsm.ExtrapolationBC(3), band_layers=band).current_state()
    b = _device_field(lsm, phi, grid, lsm.ExtrapolationBC(3), band_layers=band).current_state()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lsm.reinitialize_(a)
    with pytest.warns(UserWarning, match="did not converge for") as rec:
        lsm.reinitialize_(b, xtol=1e-300)
    nodes = b.active_count() if band else int(np.prod(n))
    assert f"did not converge for {nodes} / {nodes} points" in str(rec[0].message)
    m = a.active_mask() if band else np.ones(n, bool)
    assert np.abs(a.values()[m] - b.values()[m]).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(120, 120), (56, 56, 56)])
def test_a_band_that_outgrows_the_previous_calls_buffers(lsm, n):
    """A call is launched without waiting for its own counts: node list, candidate cells and samples go into buffers sized from the
    previous call on the handle, and a call whose band no longer fits repeats itself with larger ones — ϕ untouched in between.  One
    handle reinitialises a small circle, then (copy!) a band three times as long, then the small one again; every result is that
    of a fresh handle, and a second call on the unchanged band (buffers now fit: one round) repeats the first."""
    nd = len(n)
    grid = lsm.CartesianGrid((-1.0,) * nd, (1.0,) * nd, n)

    def band(r):
        return _device_field(lsm, _field(n, lambda X: np.sqrt(sum(x ** 2 for x in X)) * 1.3 - 1.3 * r), grid, lsm.ExtrapolationBC(2), band_layers=3)

    def fresh(r):
        st = band(r).current_state()
        lsm.reinitialize_(st)
        return st.values(), st.active_mask()

    small, big = fresh(0.22), fresh(0.75)
    eq = band(0.22)
    st = eq.current_state()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for r, (want, m) in ((0.22, small), (0.75, big), (0.75, big), (0.22, small)):
            st.copy_(band(r).current_state())
            lsm.reinitialize_(st)
            assert (st.active_mask() == m).all()
            assert np.abs(st.values()[m] - want[m]).max() < 1e-12, r
            lsm.reinitialize_(st)          # ϕ is a distance function now: again, from the buffers that fit
            assert np.abs(st.values()[m] - want[m]).max() < 1e-4     # (not idempotent: the interpolant of the new values differs at O(h^4))


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


# ------------------------------------------------------------------ NewtonSDF objects (point queries)

@pytest.mark.gpu
def test_newton_sdf_2d_circle_reference_test():
    """test/test-reinitializer.jl:13-36: spot checks inside / on / outside the interface and the sampled-grid error."""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (50, 50))
    r = 0.5
    exact = lambda x: np.hypot(x[..., 0], x[..., 1]) - r
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - r, grid), bc=lsm.ExtrapolationBC(2))
    sdf = lsm.NewtonSDF(eq.current_state(), upsample=4)
    assert abs(sdf([0.0, 0.0]) + r) < 2e-5
    assert abs(sdf([r, 0.0])) < 2e-5
    assert abs(sdf([1.0, 0.0]) - (1 - r)) < 2e-5
    X = np.stack(np.meshgrid(*grid.coords(), indexing="ij"), axis=-1).reshape(-1, 2, order="F")[::10]
    assert np.abs(sdf(X) - exact(X)).max() < 1e-5
    # get_sample_points (:56-63): the samples lie on the interface of the interpolant
    pts = sdf.get_sample_points()
    itp = lsm.InterpolatedField(eq.current_state(), 3)
    assert len(pts) > 0 and np.abs(itp(pts)).max() < 1e-6


@pytest.mark.gpu
def test_newton_sdf_3d_sphere_reference_test():
    """test/test-reinitializer.jl:38-54"""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0,) * 3, (1.0,) * 3, (25, 25, 25))
    r = 0.45
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - r, grid),
                              bc=lsm.ExtrapolationBC(2))
    sdf = lsm.NewtonSDF(eq.current_state(), upsample=3)
    assert abs(sdf([r, 0.0, 0.0])) < 1e-4
    assert abs(sdf([0.0, 0.0, 0.0]) + r) < 1e-4
    X = np.stack(np.meshgrid(*grid.coords(), indexing="ij"), axis=-1).reshape(-1, 3, order="F")[::20]
    assert np.abs(sdf(X) - (np.sqrt((X ** 2).sum(axis=1)) - r)).max() < 5e-3


@pytest.mark.gpu
def test_newton_sdf_from_a_narrow_band_and_sign_far_outside():
    """test/test-narrow-band.jl:105-145: an SDF built from a band agrees with the one of the full field near the
    interface, and its sign far outside the band (domain corners) comes from the closest point's normal."""
    import lsm_amd as lsm
    r = 0.5
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (50, 50))
    f = lambda x: np.hypot(x[0], x[1]) - r
    band = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.NarrowBandMeshField(lsm.MeshField(f, grid), nlayers=5), bc=lsm.ExtrapolationBC(2))
    full = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(f, grid), bc=lsm.ExtrapolationBC(2))
    sdf, sdf_full = lsm.NewtonSDF(band.current_state(), upsample=4), lsm.NewtonSDF(full.current_state(), upsample=4)
    assert abs(sdf([r, 0.0])) < 2e-5 and abs(sdf([0.0, 0.0]) + r) < 2e-5
    for x in ([0.5, 0.0], [0.3, 0.0], [0.6, 0.0]):
        assert abs(sdf(x) - sdf_full(x)) < 1e-5
    grid2 = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (80, 80))
    band2 = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.NarrowBandMeshField(lsm.MeshField(f, grid2), nlayers=5), bc=lsm.ExtrapolationBC(2))
    sdf2 = lsm.NewtonSDF(band2.current_state(), upsample=4)
    for x in ([1.8, 1.8], [-1.8, 1.8], [1.8, -1.8], [-1.8, -1.8]):
        v = sdf2(x)
        assert v > 0 and abs(v - (np.hypot(*x) - r)) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("ndim", [2, 3])
def test_newton_sdf_points_match_the_restatement(orc, ndim):
    """Closest points and signed distances at random points (near, far, outside the grid) against the literal restatement
    (exact KD-tree seed + the same Newton–Lagrange solve), to 1e-9."""
    import lsm_amd as lsm
    from _reinit_ref import ReinitRef
    n = (33, 29, 25)[:ndim]
    lc, hc = (-1.0,) * ndim, (1.0,) * ndim
    og = orc.Grid(lc, hc, n)
    X = np.meshgrid(*og.coords(), indexing="ij")
    vals = np.asfortranarray(np.sqrt(sum((x - 0.05 * (d + 1)) ** 2 for d, x in enumerate(X))) - 0.5 + 0.08 * np.sin(3 * X[0]) * np.cos(2 * X[-1]))
    obc = orc.make_bc(("extrapolation", 2), ndim)
    ref = ReinitRef(lambda J: orc.get(og, obc, vals, J), n, lc, hc, order=3, upsample=2, maxiters=10)
    lg = lsm.CartesianGrid(lc, hc, n)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(vals, lg), bc=lsm.ExtrapolationBC(2))
    sdf = lsm.NewtonSDF(eq.current_state(), order=3, upsample=2, maxiters=10)
    assert sdf.nsamples == len(ref.pts)
    rng = np.random.default_rng(9)
    pts = -1.0 + 2.0 * rng.random((40, ndim))
    pts[:4] *= 1.2                                  # some outside the grid
    cp, nfail = sdf.closest_point(pts)
    d = sdf(pts)
    assert nfail == 0
    for k, x in enumerate(pts):
        rcp, ok = ref.closest_point(x)
        assert ok
        assert np.abs(cp[k] - rcp).max() <= 1e-9, (k, cp[k], rcp)
        I = ref.cell_of(rcp)
        _, g, _ = ref.vgh(I, rcp)
        want = np.sign(np.dot(x - rcp, g)) * np.linalg.norm(x - rcp)
        assert abs(d[k] - want) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("n,order,upsample", [((33, 29), 3, 2), ((31, 30), 2, 3), ((26, 22), 1, 1), ((17, 15, 16), 3, 2), ((19, 18, 17), 2, 3),
                                              ((15, 16, 14), 3, 1)])
def test_interface_samples_are_the_restatements_set(orc, n, order, upsample):
    """get_sample_points (src/sdf.jl:56-63,186-221): the device projects every distinct start point once (a point on a face, edge or
    corner is shared by up to 2^N cells) and the cell it lands in keeps it; the restatement projects it once per sharing cell, as
    the reference does.  Same points, none twice, none missing — for upsample 1 (corners only: every start point is shared),
    2 and 3, odd and even orders, 2-D and 3-D."""
    import lsm_amd as lsm
    from _reinit_ref import ReinitRef
    nd = len(n)
    lc, hc = (-1.0,) * nd, (1.0,) * nd
    og = orc.Grid(lc, hc, n)
    X = np.meshgrid(*og.coords(), indexing="ij")
    vals = np.asfortranarray(np.sqrt(sum((x - 0.04 * (d + 1)) ** 2 for d, x in enumerate(X))) - 0.55 + 0.06 * np.sin(4 * X[0]) * np.cos(3 * X[-1]))
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(vals, lsm.CartesianGrid(lc, hc, n)), bc=lsm.ExtrapolationBC(2))
    sdf = lsm.NewtonSDF(eq.current_state(), order=order, upsample=upsample, maxiters=12)
    got = np.asarray(sdf.get_sample_points())
    obc = orc.make_bc(("extrapolation", 2), nd)
    ref = ReinitRef(lambda J: orc.get(og, obc, vals, J), n, lc, hc, order=order, upsample=upsample, maxiters=12)
    want = np.asarray(ref.pts)
    assert got.shape == want.shape, (got.shape, want.shape)
    from scipy.spatial import cKDTree
    dist, idx = cKDTree(want).query(got)
    assert dist.max() < 1e-11                       # every device sample is a sample of the restatement ...
    assert len(np.unique(idx)) == len(want)         # ... and each of those exactly once


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(150, 141), (75, 70, 72)])
def test_far_queries_and_far_nodes_cross_the_block_levels(orc, n):
    """Far from the interface the nearest sample is found over occupied blocks (8^N cells) and super-blocks (8^N blocks) instead of
    shells of cells; these grids have two and more super-blocks per dimension.  NewtonSDF at points far inside, far outside and
    beyond the grid, and reinitialize! of the dense field at its far corners, against the restatement (exact KD-tree seed)."""
    import lsm_amd as lsm
    from _reinit_ref import ReinitRef
    nd = len(n)
    lc, hc = (-1.0,) * nd, (1.0,) * nd
    og = orc.Grid(lc, hc, n)
    X = np.meshgrid(*og.coords(), indexing="ij")
    ctr = (0.45, -0.4, 0.35)[:nd]
    vals = np.asfortranarray(np.sqrt(sum((x - c) ** 2 for x, c in zip(X, ctr))) - 0.3)
    obc = orc.make_bc(("extrapolation", 2), nd)
    # the restatement samples the cells around the interface only (every other cell is proven empty anyway)
    near = np.argwhere(np.abs(vals) < 3 * (2.0 / (min(n) - 1)))
    cells = sorted({tuple(int(min(max(i + o, 0), n[d] - 2)) for d, (i, o) in enumerate(zip(I, off)))
                    for I in near for off in itertools.product((-1, 0), repeat=nd)})
    ref = ReinitRef(lambda J: orc.get(og, obc, vals, J), n, lc, hc, order=3, upsample=2, maxiters=12, cells=cells)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(vals, lsm.CartesianGrid(lc, hc, n)), bc=lsm.ExtrapolationBC(2))
    sdf = lsm.NewtonSDF(eq.current_state(), order=3, upsample=2, maxiters=12)
    assert sdf.nsamples == len(ref.pts)
    rng = np.random.default_rng(4)
    pts = np.concatenate([-1.0 + 2.0 * rng.random((24, nd)), -1.6 + 3.2 * rng.random((8, nd)), np.array([ctr]), np.array([lc]), np.array([hc])])
    cp, nfail = sdf.closest_point(pts)
    assert nfail == 0
    for k, x in enumerate(pts):
        rcp, ok = ref.closest_point(x)
        assert ok and np.abs(cp[k] - rcp).max() <= 1e-9, (k, x, cp[k], rcp)
    st = eq.current_state()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lsm.reinitialize_(st, order=3, upsample=2, maxiters=12)
    got = st.values()
    corners = list(itertools.product(*[(0, k // 3, k - 1) for k in n]))
    want, nf = ref.reinitialize(corners)
    assert nf == 0
    assert max(abs(got[I] - v) for I, v in want.items()) < 1e-9


@pytest.mark.gpu
def test_hausdorff_distance_of_two_circles():
    """src/sdf.jl:129-150: concentric circles of radii 0.5 and 0.6 are 0.1 apart; a shifted one by its shift."""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (81, 81))

    def make(cx, r):
        eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.hypot(x[0] - cx, x[1]) - r, grid), bc=lsm.ExtrapolationBC(2))
        return lsm.NewtonSDF(eq.current_state(), upsample=3)
    a, b, c = make(0.0, 0.5), make(0.0, 0.6), make(0.07, 0.5)
    assert abs(lsm.hausdorff_distance(a, b) - 0.1) < 1e-4
    assert abs(lsm.hausdorff_distance(a, c) - 0.07) < 1e-3
    assert lsm.hausdorff_distance(a, a) < 1e-7
