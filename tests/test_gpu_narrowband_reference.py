"""The testsets of test/test-narrow-band.jl, test/test-levelsetequation.jl and test/test-velocityextension.jl that
tests/test_gpu_narrowband.py does not already restate, through the host API on the device (0-based indices):
derivatives on a band equal those on the full grid, copy!, index sets, sign-preserving corner extrapolation in 2-D and 3-D,
a band around a non-SDF spiral, the 3-D band, h-convergence of the advected curve, reinitialisation driven from a prehook,
a velocity refreshed on the band by the term's update hook, the NormalMotion update hook."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsm():
    import lsm_amd
    return lsm_amd


def _band_eq(lsm, phi, nlayers, bc, terms=None, integrator=None):
    return lsm.LevelSetEquation(terms=terms or (lsm.NormalMotionTerm(0.0),), ic=lsm.NarrowBandMeshField(phi, nlayers=nlayers), bc=bc,
                                integrator=integrator)


def _sign(x):
    return int(x > 0) - int(x < 0)


def test_derivatives_on_the_band_match_the_full_grid(lsm):
    """test/test-narrow-band.jl:57-89.  D⁻/D⁺ (Upwind), weno5⁻/weno5⁺ (WENO5) through one ForwardEuler step with a unit
    velocity along ±e_dim — (ϕ − ϕ_new)/Δt = u·∂ϕ — and D⁰ through `gradient`, at the band node closest to the interface."""
    grid = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (100, 100))
    phi = lsm.MeshField(lambda x: x[0] ** 2 + x[1] ** 2 - 1, grid)
    bc = lsm.ExtrapolationBC(2)
    probe = _band_eq(lsm, phi, 5, bc).current_state()
    m = probe.active_mask()
    best = tuple(int(i) for i in np.argwhere(m)[np.argmin(np.abs(phi.vals[m]))])
    dt = 1e-3
    for scheme in (lsm.Upwind(), lsm.WENO5()):
        for dim in range(2):
            for sgn in (1.0, -1.0):
                u = tuple(sgn if d == dim else 0.0 for d in range(2))
                got = []
                for ic in (lsm.NarrowBandMeshField(phi, nlayers=5), phi):
                    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(u, scheme),), ic=ic, bc=bc, integrator=lsm.ForwardEuler())
                    eq._advance(0.0, dt)
                    got.append((phi.vals[best] - eq.current_state()[best]) / dt)
                assert got[0] == pytest.approx(got[1], rel=1e-9, abs=1e-9), (scheme, dim, sgn)
    full = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=bc).current_state()
    assert np.allclose(lsm.gradient(probe, best), lsm.gradient(full, best), rtol=1e-12)          # D⁰, both dimensions
    assert lsm.curvature(probe, best) == pytest.approx(lsm.curvature(full, best), rel=1e-10)    # D2⁰ and the mixed D2


def test_interpolation_on_a_band_matches_the_full_grid(lsm):
    """test/test-narrow-band.jl:91-103: a wide band, the interpolation stencils of the query points stay in it."""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (50, 50))
    phi = lsm.MeshField(lambda x: x[0] ** 2 + 2 * x[1] ** 2 - 0.5, grid)
    full = lsm.InterpolatedField(lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=lsm.ExtrapolationBC(2)).current_state(), 3)
    nb = lsm.InterpolatedField(_band_eq(lsm, phi, 8, lsm.ExtrapolationBC(2)).current_state(), 3)
    for x in ((0.5, 0.0), (0.0, 0.5), (0.3, 0.3)):
        assert nb(np.array(x)) == pytest.approx(full(np.array(x)), rel=1e-12, abs=1e-14)


def test_copy_into_a_band_takes_values_and_active_set(lsm):
    """test/test-narrow-band.jl:178-190"""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (30, 30))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    nb1 = _band_eq(lsm, phi, 2, lsm.LinearExtrapolationBC()).current_state()
    nb2 = _band_eq(lsm, phi, 4, lsm.LinearExtrapolationBC()).current_state()
    assert nb2.active_count() > nb1.active_count()
    nb2.copy_(nb1)
    m1 = nb1.active_mask()
    assert np.array_equal(nb2.active_mask(), m1)
    assert np.array_equal(nb2.values()[m1], nb1.values()[m1])
    I = tuple(int(i) for i in np.argwhere(m1)[0])
    J = (I[0] - 1, I[1])
    assert nb2[J] == nb1[J]


def test_active_and_full_index_sets(lsm):
    """test/test-narrow-band.jl:192-205"""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (30, 30))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    nb = _band_eq(lsm, phi, 3, lsm.LinearExtrapolationBC()).current_state()
    nodes, cells = set(lsm.nodeindices(nb.mesh)), set(lsm.cellindices(nb.mesh))
    act = lsm.active_nodeindices(nb)
    assert set(act) <= nodes and len(act) < len(nodes)
    assert set(lsm.active_cellindices(nb)) <= cells


def test_halo_node_extrapolates_finitely(lsm):
    """test/test-narrow-band.jl:243-258 (the far-node error of :259-260 is in tests/test_gpu_narrowband.py)"""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (50, 50))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    nb = _band_eq(lsm, phi, 3, lsm.LinearExtrapolationBC()).current_state()
    m = nb.active_mask()
    grown = np.zeros_like(m)
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            grown[max(di, 0):50 + min(di, 0), max(dj, 0):50 + min(dj, 0)] |= m[max(-di, 0):50 + min(-di, 0), max(-dj, 0):50 + min(-dj, 0)]
    halo = np.argwhere(grown & ~m)
    assert len(halo) > 0
    assert math.isfinite(nb[tuple(int(i) for i in halo[0])])


@pytest.mark.parametrize("n", [(40, 40), (20, 20, 20)])
def test_corner_extrapolation_preserves_the_sign(lsm, n):
    """test/test-narrow-band.jl:263-284 (2-D) and :294-314 (3-D): a thin band, both neighbours of a band node may lie
    outside it; the extrapolated value never has the wrong sign."""
    N = len(n)
    grid = lsm.CartesianGrid((-2.0,) * N, (2.0,) * N, n)
    phi = lsm.MeshField(lambda x: np.sqrt(sum(c * c for c in x)) - 0.5, grid)
    nb = _band_eq(lsm, phi, 2, lsm.ExtrapolationBC(2)).current_state()
    full = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=lsm.ExtrapolationBC(2)).current_state()
    nb.prepare(nb.buf)                       # the whole halo at once (the scalar ϕ[I] of the reference, for every I below)
    full.backend.fill_ghosts(full.buf)
    lay = nb.backend.lay
    pad = lambda f: f.buf.cpu().numpy()[:int(lay.total)]
    a, b = pad(nb), pad(full)
    off = lambda I: int(lay.origin) + sum(int(I[d]) * int(lay.stride[d]) for d in range(N))
    checked = 0
    for I in lsm.active_nodeindices(nb):
        for d in range(N):
            for s in (-1, 1):
                J = I[:d] + (I[d] + s,) + I[d + 1:]
                assert _sign(a[off(J)]) == _sign(b[off(J)]), (I, J)
                checked += 1
    assert checked > 100


def _spiral(x, y):
    """test/test-narrow-band.jl:317-327: a spiral band, NOT a signed distance"""
    d, r0, th0, al = 1, 0.5, -math.pi / 3, math.pi / 100.0
    R = np.array([[math.cos(al), -math.sin(al)], [math.sin(al), math.cos(al)]])
    M = R @ np.array([[1 / 0.06 ** 2, 0], [0, 1 / (4 * math.pi ** 2)]]) @ R.T
    r, th = np.sqrt(x * x + y * y), np.arctan2(y, x)
    res = np.full(np.broadcast(x, y).shape, 1.0e30)
    for i in range(5):
        th1 = th + (2 * i - 4) * math.pi
        v0, v1 = r - r0, th1 - th0
        res = np.minimum(res, np.sqrt(M[0, 0] * v0 * v0 + (M[0, 1] + M[1, 0]) * v0 * v1 + M[1, 1] * v1 * v1) - d)
    return res


def test_band_from_non_sdf_input(lsm):
    """test/test-narrow-band.jl:316-333"""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (100, 100))
    phi = lsm.MeshField(lambda x: _spiral(x[0], x[1]), grid)
    nb = _band_eq(lsm, phi, 6, lsm.ExtrapolationBC(2)).current_state()
    assert nb.active_count() > 3000


def test_3d_band(lsm):
    """test/test-narrow-band.jl:335-353"""
    grid = lsm.CartesianGrid((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), (25, 25, 25))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.45, grid)
    nb = _band_eq(lsm, phi, 3, lsm.ExtrapolationBC(2)).current_state()
    full = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=lsm.ExtrapolationBC(2)).current_state()
    assert nb.mesh.ndim == 3 and 0 < nb.active_count() < 25 ** 3
    m = nb.active_mask()
    best = tuple(int(i) for i in np.argwhere(m)[np.argmin(np.abs(phi.vals[m]))])
    assert np.allclose(lsm.gradient(nb, best), lsm.gradient(full, best), rtol=1e-12)           # D⁰ in the three dimensions
    sdf = lsm.NewtonSDF(nb, upsample=3)
    assert abs(sdf(np.array([0.45, 0.0, 0.0]))) < 1.0e-3


def test_band_h_convergence_of_the_advected_curve(lsm):
    """test/test-narrow-band.jl:355-400: a circle advected at (1, 0) to t = 0.5 with RK3, reinitialised after every step;
    the error at the band nodes within 3h of the true interface converges at order ≥ 2.5."""
    r, tf, Ns, errs = 0.5, 0.5, (30, 60, 120), []
    for N in Ns:
        grid = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (N, N))
        h = min(grid.meshsize())
        phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - r, grid)
        eq = _band_eq(lsm, phi, 5, lsm.ExtrapolationBC(2), terms=(lsm.AdvectionTerm((1.0, 0.0)),), integrator=lsm.RK3())
        lsm.integrate_(eq, tf, posthook=lambda e: lsm.reinitialize_(e.current_state()))
        st = eq.current_state()
        xs = grid.coords()
        exact = np.hypot(xs[0][:, None] - tf, xs[1][None, :]) - r
        near = st.active_mask() & (np.abs(exact) < 3 * h)
        errs.append(np.abs(st.values() - exact)[near].max())
    orders = [math.log(errs[i] / errs[i + 1]) / math.log(Ns[i + 1] / Ns[i]) for i in range(2)]
    assert all(o >= 2.5 for o in orders), (errs, orders)


def test_reinitialisation_from_a_prehook_with_rk2(lsm):
    """test/test-levelsetequation.jl:121-131: periodic advection with reinitialize! at the start of every step (RK2, the
    default integrator) runs through; here also: the field stays a distance function."""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (33, 33))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((1.0, 0.0)),), ic=phi, bc=lsm.PeriodicBC())
    calls = []
    lsm.integrate_(eq, 0.2, prehook=lambda e: calls.append(lsm.reinitialize_(e.current_state())))
    assert isinstance(eq, lsm.LevelSetEquation) and eq.current_time() == 0.2 and len(calls) > 0
    xs = grid.coords()
    exact = np.hypot(xs[0][:, None] - 0.2, xs[1][None, :]) - 0.5
    near = np.abs(exact) < 0.2
    assert np.abs(eq.current_state().values() - exact)[near].max() < 0.02


def test_velocity_refreshed_on_the_band_by_the_update_hook(lsm):
    """test/test-levelsetequation.jl:223-247: the velocity is known only on the state's band and is refilled from the
    stage field's active set by the term's update hook before every stage, then read through the WENO5 stencil.  The
    device coefficient is a dense array whose off-band entries are never meaningful: the hook zeroes them."""
    grid = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (60, 60))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    bc = lsm.ExtrapolationBC(2)
    X, Y = np.meshgrid(*grid.coords(), indexing="ij")
    seen, box = [], {}

    def refill(coeff, psi, t):                           # psi: the stage field; its active set is the state's
        m = box["eq"].current_state().active_mask()
        seen.append(int(m.sum()))
        coeff.set_values(np.stack([np.where(m, -Y, 0.0), np.where(m, X, 0.0)]))

    ic = lsm.NarrowBandMeshField(phi, nlayers=5)
    eq_nb = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.MeshField(np.zeros((2, 60, 60)), grid), lsm.WENO5(), refill),), ic=ic, bc=bc)
    box["eq"] = eq_nb
    eq_full = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation()),), ic=phi, bc=bc)
    lsm.integrate_(eq_full, 0.3, prehook=lambda e: lsm.reinitialize_(e.current_state()))
    lsm.integrate_(eq_nb, 0.3, posthook=lambda e: lsm.reinitialize_(e.current_state()))
    assert len(seen) > 3 and all(0 < k < 3600 for k in seen)
    st, full = eq_nb.current_state(), eq_full.current_state().values()
    gamma = 5 * min(grid.meshsize())
    v = st.values()
    sel = st.active_mask() & (np.abs(np.nan_to_num(v, nan=1e9)) < gamma / 2)
    assert np.abs(v - full)[sel].max() < 0.05


def test_normal_motion_update_hook(lsm):
    """test/test-velocityextension.jl:4-17: update_term! hands the hook the speed field, the state and the time."""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (21, 21))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    v = lsm.MeshField(np.zeros((21, 21)), grid)

    def hook(speed, psi, t):                             # speed: the term's coefficient (device arrays behind set_values)
        speed.set_values(np.full((21, 21), 2 * t))

    term = lsm.NormalMotionTerm(v, hook)
    eq = lsm.LevelSetEquation(terms=(term,), ic=phi, bc=lsm.PeriodicBC())
    eq._update_terms(eq.current_state(), 0.3)
    assert np.all(eq.backend.download_side(term.coeff.fields[0]) == 0.6)
