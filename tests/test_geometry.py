"""curvature / gradient / normal at a node (src/levelsetops.jl:197-226) as fields: the oracle restatement against
analytic values, the device kernel (lsm_geometry) against the oracle, and the reference's curvature-driven
velocity-extension cycles (test/test-velocityextension.jl:106-207) through the host API with the speed update
running on the device."""
import numpy as np
import pytest


def _sphere(orc, n, ndim, r0=0.5):
    g = orc.Grid((-1.0,) * ndim, (1.0,) * ndim, (n,) * ndim)
    X = np.meshgrid(*g.coords(), indexing="ij")
    r = np.sqrt(sum(x * x for x in X))
    return g, np.asfortranarray(r - r0), X, r


@pytest.mark.parametrize("ndim", [2, 3])
def test_oracle_curvature_and_normal_of_a_sphere(orc, ndim):
    """κ = (N-1)/r and n = x/r, second-order accurate away from the centre."""
    errs = []
    for n in (41, 81):
        g, phi, X, r = _sphere(orc, n, ndim)
        bc = orc.make_bc("linear", ndim)
        m = np.abs(r - 0.5) < 0.1
        k = orc.geometry(g, bc, phi, "curvature")
        errs.append(np.abs(k[m] - (ndim - 1) / r[m]).max())
        nr = orc.geometry(g, bc, phi, "normal")
        for d in range(ndim):
            assert np.abs(nr[d][m] - X[d][m] / r[m]).max() < 2e-3
    assert errs[1] < errs[0] / 3.0 and errs[1] < 6e-3 * (ndim - 1)


def test_oracle_gradient_is_exact_for_linear_fields_and_curvature_vanishes_with_the_gradient(orc):
    g = orc.Grid((0.0, 0.0, 0.0), (1.0, 2.0, 3.0), (9, 11, 13))
    phi = g.sample(lambda x, y, z: 2 * x - 3 * y + 0.5 * z + 1)
    bc = orc.make_bc("linear", 3)
    gr = orc.geometry(g, bc, phi, "gradient")
    for d, v in enumerate((2.0, -3.0, 0.5)):
        assert np.abs(gr[d] - v).max() < 1e-12
    flat = np.asfortranarray(np.full((9, 11, 13), 0.3))
    assert np.array_equal(orc.geometry(g, bc, flat, "curvature"), np.zeros((9, 11, 13)))   # src/levelsetops.jl:201


def _device_state(lsm, og, phi, bc, dtype=None):
    lg = lsm.CartesianGrid(og.lc, og.hc, og.n)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(phi, lg, dtype=dtype), bc=bc)
    return eq, eq.current_state()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,bcname", [((61,), "periodic"), ((45, 37), "linear"), ((23, 19, 21), "neumann"), ((23, 19, 21), "periodic")])
def test_gpu_geometry_matches_oracle(orc, shape, bcname):
    import lsm_amd as lsm
    nd = len(shape)
    og = orc.Grid((-1.0,) * nd, tuple(1.0 + 0.1 * d for d in range(nd)), shape)
    X = np.meshgrid(*og.coords(), indexing="ij")
    phi = np.asfortranarray(np.sqrt(sum((x - 0.1 * (d + 1)) ** 2 for d, x in enumerate(X))) - 0.55 + 0.05 * np.sin(3 * X[0]))
    lbc = {"periodic": lsm.PeriodicBC(), "linear": lsm.LinearExtrapolationBC(), "neumann": lsm.NeumannBC()}[bcname]
    obc = orc.make_bc(bcname, nd)
    _, st = _device_state(lsm, og, phi, lbc)
    want = orc.geometry(og, obc, phi, "curvature")
    got = lsm.curvature_field(st).values()
    assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()       # pow(q, 1.5): device libm vs host libm
    got = lsm.curvature_field(st, scale=-2.0).values()
    assert np.abs(got + 2.0 * want).max() <= 2e-13 * np.abs(want).max()
    wg = orc.geometry(og, obc, phi, "gradient")
    for d, f in enumerate(lsm.gradient_field(st)):
        assert np.array_equal(f.values(), wg[d])
    wn = orc.geometry(og, obc, phi, "normal")
    for d, f in enumerate(lsm.normal_field(st)):
        assert np.abs(f.values() - wn[d]).max() <= 4e-16
    I = tuple(n // 3 for n in shape)
    assert abs(lsm.curvature(st, I) - want[I]) <= 1e-13 * abs(want[I]) + 1e-15
    assert np.array_equal(lsm.gradient(st, I), np.array([w[I] for w in wg]))


@pytest.mark.gpu
def test_gpu_curvature_seed_band_and_frozen_mask(orc):
    """The seed loop of test/test-velocityextension.jl:118-131 in one launch: -κ on |ϕ| <= 1.5Δ, 0 elsewhere, and
    the frozen mask of exactly those nodes; float32 storage widens exactly."""
    import lsm_amd as lsm
    og, phi, X, r = _sphere(orc, 48, 3, 0.45)
    d = min(og.meshsize())
    for dtype in (None, np.float32):
        p = phi if dtype is None else np.asfortranarray(phi.astype(np.float32).astype(np.float64))
        _, st = _device_state(lsm, og, phi, lsm.PeriodicBC(), dtype=dtype)
        want = orc.geometry(og, orc.make_bc("periodic", 3), p, "curvature")
        fz = lsm.SideField(st.backend, st.mesh)
        out = lsm.curvature_field(st, scale=-1.0, band=1.5 * d, fill=0.0, frozen_out=fz)
        on = np.abs(p) <= 1.5 * d
        got, gfz = out.values(), fz.values()
        assert np.array_equal(gfz, on.astype(np.float64))
        assert np.array_equal(got[~on], np.zeros((~on).sum()))
        assert np.abs(got[on] + want[on]).max() <= 1e-13 * np.abs(want[on]).max()
        assert on.sum() > 2000


@pytest.mark.gpu
def test_gpu_geometry_argument_checks():
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (33, 33))
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - 0.5, grid), bc=lsm.NeumannBC())
    st = eq.current_state()
    with pytest.raises(ValueError):
        lsm.curvature_field(np.zeros((33, 33)))
    with pytest.raises(lsm.LsmError):
        st.backend.geometry(7, st.buf, [st.backend.alloc_side()])
    with pytest.raises(lsm.LsmError):
        st.backend.geometry(lsm._lib.GEOM_NORMAL, st.buf, [st.backend.alloc_side()])     # needs one output per dimension
    with pytest.raises(lsm.LsmError):
        st.backend.geometry(lsm._lib.GEOM_CURVATURE, st.buf, [st.buf])                    # output aliases ϕ


def _run_curvature_extension_cycle(lsm, phi_ic, grid, nsteps, dt_motion, dt_reinit, ext_iters, seed_band=1.5):
    """_run_curvature_extension_cycle! (test/test-velocityextension.jl:106-152); the speed update runs on the device:
    lsm_geometry seeds -κ on the interface band and marks it frozen, lsm_extend_along_normals extends it."""
    delta = min(grid.meshsize())

    def update_speed(coeff, phi_state, t):
        b = phi_state.backend
        speed = lsm.SideField(b, phi_state.mesh, coeff.fields[0])
        frozen = lsm.SideField(b, phi_state.mesh)
        lsm.curvature_field(phi_state, scale=-1.0, band=seed_band * delta, fill=0.0, out=speed, frozen_out=frozen)
        F = lsm.ROCMeshField(b, phi_state.mesh, phi_state.bcs, buf=coeff.fields[0])
        lsm.extend_along_normals_(F, phi_state, frozen=frozen, cfl=0.3, nb_iters=ext_iters)

    speed = lsm.MeshField(np.zeros(grid.n), grid)
    eq_motion = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(speed, update_speed),), ic=phi_ic, bc=lsm.PeriodicBC(),
                                     integrator=lsm.ForwardEuler(cfl=0.35))
    eq_reinit = lsm.LevelSetEquation(terms=(lsm.EikonalReinitializationTerm(),), ic=phi_ic, bc=lsm.PeriodicBC(),
                                     integrator=lsm.ForwardEuler(cfl=0.45))
    for _ in range(nsteps):
        lsm.integrate_(eq_motion, eq_motion.current_time() + dt_motion, dt_motion)
        lsm.integrate_(eq_reinit, eq_reinit.current_time() + dt_reinit, dt_reinit)
    return eq_motion.current_state()


def _interface_radius_stats(grid, vals, band=1.5):
    """test/test-velocityextension.jl:154-170"""
    delta = min(grid.meshsize())
    X = np.meshgrid(*grid.coords(), indexing="ij")
    r = np.sqrt(sum(x * x for x in X))
    radii = r[np.abs(vals) <= band * delta]
    mean = radii.sum() / radii.size
    return mean, np.sqrt(((radii - mean) ** 2).sum() / radii.size), radii.size


@pytest.mark.gpu
def test_classical_circular_reconstruction_2d():
    """test/test-velocityextension.jl:172-189"""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-0.5, -0.5), (0.5, 0.5), (128, 128))
    R0 = 0.45
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - R0, grid)
    delta = min(grid.meshsize())
    out = _run_curvature_extension_cycle(lsm, phi, grid, nsteps=3, dt_motion=1.2e-3, dt_reinit=delta, ext_iters=30)
    rmean, rstd, npts = _interface_radius_stats(grid, out.values())
    assert npts > 300
    assert rmean < R0
    assert rstd / rmean < 0.05


@pytest.mark.gpu
def test_classical_spherical_reconstruction_3d():
    """test/test-velocityextension.jl:191-207"""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), (48, 48, 48))
    R0 = 0.45
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - R0, grid)
    delta = min(grid.meshsize())
    out = _run_curvature_extension_cycle(lsm, phi, grid, nsteps=2, dt_motion=7.0e-4, dt_reinit=0.15 * delta, ext_iters=22)
    rmean, rstd, npts = _interface_radius_stats(grid, out.values())
    assert npts > 2000
    assert rmean < R0
    assert rstd / rmean < 0.09


@pytest.mark.gpu
def test_crystal_normal_extension_signs():
    """test/test-velocityextension.jl:209-287: -κ seeded on the 1.5Δ band of a six-fold crystal and extended along the
    normals is negative at the tips and positive at the kinks, and one short NormalMotion step with it reduces the
    shape anisotropy."""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (161, 161))
    R, deformation, nfacets = 0.6, 0.45, 6
    phi = lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - R * (1 + deformation * np.cos(nfacets * np.arctan2(x[1], x[0]))), grid)
    delta = min(grid.meshsize())
    eq0 = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=lsm.PeriodicBC())
    st = eq0.current_state()
    frozen = lsm.SideField(st.backend, st.mesh)
    v = lsm.curvature_field(st, scale=-1.0, band=1.5 * delta, fill=0.0, frozen_out=frozen)
    F = lsm.ROCMeshField(st.backend, st.mesh, st.bcs, buf=v.buf)
    lsm.extend_along_normals_(F, st, frozen=frozen, cfl=0.3, nb_iters=45)
    vals = v.values()
    h = grid.meshsize()

    def closest(x):
        return tuple(int(np.clip(round((x[d] - grid.lc[d]) / h[d]), 0, grid.n[d] - 1)) for d in range(2))
    tips, kinks = [], []
    for k in range(nfacets):
        th = 2 * np.pi * k / nfacets
        r = R * (1 + deformation * np.cos(nfacets * th))
        tips.append(vals[closest((r * np.cos(th), r * np.sin(th)))])
        th = (2 * k + 1) * np.pi / nfacets
        r = R * (1 + deformation * np.cos(nfacets * th))
        kinks.append(vals[closest((r * np.cos(th), r * np.sin(th)))])
    assert np.mean(tips) < 0
    assert np.mean(kinks) > 0

    def radius_cv(a):
        X = np.meshgrid(*grid.coords(), indexing="ij")
        rs = np.hypot(X[0], X[1])[np.abs(a) <= 1.5 * delta]
        return rs.std() / rs.mean()
    cv0 = radius_cv(phi.vals)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(lsm.MeshField(vals, grid)),), ic=phi, bc=lsm.PeriodicBC(),
                              integrator=lsm.ForwardEuler(cfl=0.3))
    lsm.integrate_(eq, 2.5e-3, 2.5e-3)
    assert radius_cv(eq.current_state().values()) < cv0


@pytest.mark.gpu
def test_interpolated_field_on_device_reference_tests(orc):
    """test/test-interpolation.jl:36-80,107-137 through the host API on the device: cubic patches reproduce quadratics
    with gradient and Hessian (2-D, 3-D), the least-squares order 2 does too; argument checks."""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (21, 21))
    f = lambda x: x[0] ** 2 + 2 * x[1] ** 2 - 0.5
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(f, grid), bc=lsm.ExtrapolationBC(2))
    x = np.array([0.15, -0.25])
    for order in (3, 2):
        itp = lsm.InterpolatedField(eq.current_state(), order)
        assert abs(itp(x) - f(x)) < 1e-12
        assert np.abs(itp.gradient(x) - np.array([2 * x[0], 4 * x[1]])).max() < 1e-12
    itp = lsm.InterpolatedField(eq.current_state(), 3)
    assert np.abs(itp.hessian(x) - np.array([[2.0, 0.0], [0.0, 4.0]])).max() < 1e-10
    v, g = itp.value_and_gradient(x)
    assert abs(v - f(x)) < 1e-12 and np.abs(g - np.array([0.3, -1.0])).max() < 1e-12
    v, g, H = itp.value_gradient_hessian(x)
    assert abs(v - f(x)) < 1e-12 and np.abs(H - np.array([[2.0, 0.0], [0.0, 4.0]])).max() < 1e-10
    g3 = lsm.CartesianGrid((-1.0,) * 3, (1.0,) * 3, (11, 11, 11))
    f3 = lambda x: x[0] ** 2 + x[1] ** 2 + x[2] ** 2 - 0.5
    e3 = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(f3, g3), bc=lsm.ExtrapolationBC(2))
    x3 = np.array([0.1, -0.2, 0.3])
    it3 = lsm.InterpolatedField(e3.current_state(), 3)
    assert abs(it3(x3) - f3(x3)) < 1e-12 and np.abs(it3.gradient(x3) - 2 * x3).max() < 1e-12
    with pytest.raises(ValueError):
        lsm.InterpolatedField(e3.current_state(), 6)
    with pytest.raises(ValueError):
        it3(np.array([0.1, 0.2]))


@pytest.mark.gpu
@pytest.mark.parametrize("ndim,order", [(1, 1), (2, 2), (2, 3), (3, 3), (2, 5), (3, 4)])
def test_interpolated_field_matches_the_restatement(orc, ndim, order):
    """Device patch evaluation (the one reinitialize! uses) against tests/_reinit_ref.py at random points, incl. points in
    boundary cells (stencils reach the ghost layers) and outside the grid (clamped cell): value, gradient, Hessian."""
    import lsm_amd as lsm
    from _reinit_ref import ReinitRef
    n = (17, 15, 13)[:ndim]
    lc, hc = (-1.0, -0.8, -0.6)[:ndim], (1.0, 1.2, 0.9)[:ndim]
    og = orc.Grid(lc, hc, n)
    X = np.meshgrid(*og.coords(), indexing="ij")
    vals = np.asfortranarray(np.sqrt(sum((x - 0.1 * (d + 1)) ** 2 for d, x in enumerate(X)) + 0.05) - 0.5 + 0.1 * np.sin(2 * X[0]))
    bcs = ("extrapolation", 2)
    obc = orc.make_bc(bcs, ndim)
    ref = ReinitRef(lambda J: orc.get(og, obc, vals, J), n, lc, hc, order=order, cells=[])
    lg = lsm.CartesianGrid(lc, hc, n)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(vals, lg), bc=lsm.ExtrapolationBC(2))
    itp = lsm.InterpolatedField(eq.current_state(), order)
    rng = np.random.default_rng(5)
    pts = np.array(lc) + rng.random((60, ndim)) * (np.array(hc) - np.array(lc))
    pts[:5] = np.array(lc) + 1e-3                      # first cell
    pts[5:10] = np.array(hc) - 1e-3                    # last cell
    pts[10:12] = np.array(hc) + 0.05                   # outside: the clamped cell's patch, extrapolated
    v, g, H = itp.value_gradient_hessian(pts)
    for k, x in enumerate(pts):
        rv, rg, rH = ref.vgh(ref.cell_of(x), x)
        assert abs(v[k] - rv) <= 1e-12 * max(1.0, abs(rv)), (k, v[k], rv)
        assert np.abs(g[k] - rg).max() <= 1e-11 * max(1.0, np.abs(rg).max())
        assert np.abs(H[k] - rH).max() <= 1e-9 * max(1.0, np.abs(rH).max())


@pytest.mark.gpu
def test_geometry_queries_on_a_band_field(orc):
    """docs/src/geometry-queries.md: curvature / gradient / normal work on a NarrowBandMeshField as on a MeshField.  At
    the band nodes whose whole stencil lies in the band they equal the dense values; elsewhere on the band they use the
    band's extrapolated neighbours (checked against the dict-based restatement of ϕ[I]); off the band: fill."""
    import lsm_amd as lsm
    from _nb_ref import NBRef
    n = (41, 37)
    og = orc.Grid((-1.0, -1.0), (1.0, 1.0), n)
    X = np.meshgrid(*og.coords(), indexing="ij")
    vals = np.asfortranarray(np.hypot(X[0] - 0.1, X[1] + 0.05) - 0.55)
    lg = lsm.CartesianGrid(og.lc, og.hc, n)
    bc = lsm.LinearExtrapolationBC()
    dense = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(vals, lg), bc=bc).current_state()
    band = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.NarrowBandMeshField(lsm.MeshField(vals, lg), nlayers=3), bc=bc).current_state()
    m = band.active_mask()
    kd, kb = lsm.curvature_field(dense).values(), lsm.curvature_field(band, fill=-7.0).values()
    assert np.array_equal(kb[~m], np.full((~m).sum(), -7.0))
    inner = m.copy()                                             # band nodes whose 3x3 neighbourhood lies in the band
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            inner &= np.roll(np.roll(m, di, 0), dj, 1)
    inner[[0, -1], :] = False
    inner[:, [0, -1]] = False
    assert inner.sum() > 100 and np.array_equal(kb[inner], kd[inner])
    # the band's edge nodes: the centred differences read extrapolated values, as the reference's nb[I] does
    ref = NBRef(vals, 3)
    h = og.meshsize()
    gb = [g.values() for g in lsm.gradient_field(band)]
    edge = [tuple(int(i) for i in I) for I in np.argwhere(m & ~inner) if 0 < I[0] < n[0] - 1 and 0 < I[1] < n[1] - 1][:40]
    assert len(edge) == 40
    for I in edge:
        for d in range(2):
            e = tuple(1 if k == d else 0 for k in range(2))
            want = (ref.get((I[0] + e[0], I[1] + e[1])) - ref.get((I[0] - e[0], I[1] - e[1]))) / (2 * h[d])
            assert gb[d][I] == want, (I, d, gb[d][I], want)
