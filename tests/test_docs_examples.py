"""The hidden `@assert`s of the reference's documentation examples (docs/src/*.md, "# hide" lines): analytic
relations the reference's own doc build checks on every run, restated on the oracle (CPU) and through the host API on
the device (GPU).  The one that reaches the WENO5 arithmetic of the hot path is the integrator comparison
(docs/src/time-integrators.md:49-70): AdvectionTerm's default scheme is WENO5, and the relations between the area
lost by ForwardEuler, RK2 and RK3 over one revolution hold only while the spatial error stays below the temporal one."""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lsm():
    import lsm_amd
    return lsm_amd


# ----------------------------------------------------------------------------- docs/src/time-integrators.md:49-70
def _dumbbell(sample):
    disk = lambda c: sample(lambda x, y: np.hypot(x - c[0], y - c[1]) - 0.25)
    bar = sample(lambda x, y: np.maximum(np.abs(x) - 0.5, np.abs(y) - 0.1))
    return np.minimum(np.minimum(disk((-0.5, 0.0)), disk((0.5, 0.0))), bar)      # disk ∪ disk ∪ bar


def _check_area_errors(errs):
    assert errs["ForwardEuler"] > 4 * errs["RK2"], errs            # :68
    assert errs["ForwardEuler"] > 4 * errs["RK3"], errs            # :69
    assert abs(errs["RK2"] - errs["RK3"]) < 0.1 * errs["RK2"], errs   # :70


def test_integrator_comparison_on_the_oracle(orc):
    grid = orc.Grid((-1, -1), (1, 1), (64, 64))
    phi0 = np.asfortranarray(_dumbbell(grid.sample))
    bc = orc.make_bc("neumann", 2)
    v0 = orc.volume(grid, phi0)
    errs = {}
    for name, integ in (("ForwardEuler", orc.FE), ("RK2", orc.RK2), ("RK3", orc.RK3)):
        phi = phi0.copy(order="F")
        steps, t, _ = orc.integrate(integ, grid, bc, phi, [orc.advection(orc.rotation())], 2 * math.pi)   # WENO5: the default scheme
        assert steps == 792 and t == 2 * math.pi                   # docs/src/time-integrators.md:92-94
        errs[name] = abs(orc.volume(grid, phi) - v0) / v0
    _check_area_errors(errs)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_integrator_comparison_on_the_device(lsm, mode):
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (64, 64))
    phi0 = lsm.MeshField(_dumbbell(lambda f: lsm.MeshField(lambda x: f(x[0], x[1]), grid).vals), grid)
    errs = {}
    for name, integ in (("ForwardEuler", lsm.ForwardEuler()), ("RK2", lsm.RK2()), ("RK3", lsm.RK3())):
        eq = lsm.LevelSetEquation(terms=lsm.AdvectionTerm(lsm.RigidRotation()), ic=phi0, bc=lsm.NeumannBC(), integrator=integ, mode=mode)
        v0 = lsm.volume(eq)
        lsm.integrate_(eq, 2 * math.pi)
        errs[name] = abs(lsm.volume(eq) - v0) / v0
    _check_area_errors(errs)


# ----------------------------------------------------------------------------- docs/src/levelset-equation.md:80-96
@pytest.mark.gpu
def test_integrate_lands_on_tf_with_and_without_a_step_cap(lsm):
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (50, 50))
    phi = lsm.MeshField(lambda x: x[0] ** 2 + x[1] ** 2 - 0.5 ** 2, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.MeshField(lambda x: (1.0 + 0 * x[0], 0.0 * x[0]), grid)),), ic=phi, bc=lsm.NeumannBC())
    lsm.integrate_(eq, 0.5)
    assert eq.current_time() == 0.5                                # :83
    steps = []
    lsm.integrate_(eq, 1.0, 0.01, posthook=lambda e: steps.append(e.current_time()))
    assert eq.current_time() == 1.0                                # :95
    assert max(np.diff([0.5] + steps)) <= 0.01 * (1 + 1e-12)       # "no internal step exceeding Δt = 0.01"


# ----------------------------------------------------------------------------- docs/src/grids.md:85-89, geometry.md:27-32
def test_narrow_band_sizes_of_the_docs_on_the_restatement():
    from _nb_ref import NBRef
    def active(n, f):
        X = np.meshgrid(*[np.linspace(-1, 1, k) for k in n], indexing="ij")
        return len(NBRef(f(X[0], X[1]), 3).d), n[0] * n[1]
    a, tot = active((32, 32), lambda x, y: np.hypot(x, y) - 0.5)
    assert 0 < a < tot                                             # geometry.md:31
    a, tot = active((64, 64), lambda x, y: x ** 2 + y ** 2 - 0.5 ** 2)   # grids.md:24,63: the page's grid and field
    assert 300 < a < 1000                                          # grids.md:88


@pytest.mark.gpu
def test_narrow_band_sizes_of_the_docs_on_the_device(lsm):
    def active(n, f):
        grid = lsm.CartesianGrid((-1, -1), (1, 1), n)
        eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.NarrowBandMeshField(lsm.MeshField(f, grid), nlayers=3), bc=lsm.NeumannBC())
        return eq.current_state().active_count(), n[0] * n[1]
    a, tot = active((32, 32), lambda x: np.hypot(x[0], x[1]) - 0.5)
    assert 0 < a < tot                                             # geometry.md:31
    a, tot = active((64, 64), lambda x: x[0] ** 2 + x[1] ** 2 - 0.5 ** 2)   # grids.md:24,63: the page's grid and field
    assert 300 < a < 1000                                          # grids.md:88


# ----------------------------------------------------------------------------- docs/src/geometry-queries.md:40-46
def test_volume_error_shrinks_under_refinement(orc):
    def err(n):
        g = orc.Grid((-1, -1), (1, 1), (n, n))
        return abs(orc.volume(g, g.sample(lambda x, y: np.hypot(x, y) - 0.5)) - math.pi * 0.25)
    assert err(128) < err(32)                                      # :44


@pytest.mark.gpu
def test_volume_error_shrinks_under_refinement_on_the_device(lsm):
    def err(n):
        grid = lsm.CartesianGrid((-1, -1), (1, 1), (n, n))
        eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - 0.5, grid), bc=lsm.NeumannBC())
        return abs(lsm.volume(eq) - math.pi * 0.25)
    assert err(128) < err(32)


# ----------------------------------------------------------------------------- docs/src/signed-distance.md:35-40,66-70,101-106
@pytest.mark.gpu
def test_reinitialize_recovers_the_circle_distance(lsm):
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (64, 64))
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: x[0] ** 2 + x[1] ** 2 - 0.5 ** 2, grid),
                              bc=lsm.LinearExtrapolationBC())
    phi = eq.current_state()
    lsm.reinitialize_(phi)
    xs = grid.coords()
    exact = np.hypot(xs[0][:, None], xs[1][None, :]) - 0.5
    assert np.abs(phi.values() - exact).max() < 1e-6               # :69
    sdf = lsm.NewtonSDF(phi, upsample=2)
    assert abs(sdf(np.array([0.0, 0.0])) + 0.5) < 1e-3             # :105


# ----------------------------------------------------------------------------- docs/src/velocity-extension.md:40-52,76-82
def _speed_near_the_circle(coords, delta):
    X, Y = np.meshgrid(*coords, indexing="ij")
    return np.where(np.abs(np.hypot(X, Y) - 0.5) <= 1.5 * delta, np.sin(np.arctan2(Y, X)), 0.0)


@pytest.mark.gpu
def test_extension_carries_the_interface_speed_along_the_normal(lsm):
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (64, 64))
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - 0.5, grid),
                              bc=lsm.LinearExtrapolationBC())
    phi = eq.current_state()
    delta = min(grid.meshsize())
    F = lsm.ROCMeshField.from_host(eq.backend, lsm.MeshField(_speed_near_the_circle(grid.coords(), delta), grid), bcs=phi.bcs)
    lsm.extend_along_normals_(F, phi, nb_iters=90)
    Fi = lsm.InterpolatedField(F, 1)
    vals = [Fi(np.array([r * math.cos(math.pi / 4), r * math.sin(math.pi / 4)])) for r in (0.3, 0.5, 0.8)]
    assert max(vals) - min(vals) < 2e-2                            # :80
    assert all(abs(v - math.sin(math.pi / 4)) < 2e-2 for v in vals)   # :81


def test_extension_carries_the_interface_speed_on_the_oracle(orc):
    """the same example on the oracle, probed at the nodes nearest the three radii (the oracle has no interpolant)"""
    grid = orc.Grid((-1, -1), (1, 1), (64, 64))
    phi = grid.sample(lambda x, y: np.hypot(x, y) - 0.5)
    delta = min(grid.meshsize())
    F = np.asfortranarray(_speed_near_the_circle(grid.coords(), delta))
    orc.extend_along_normals(grid, orc.make_bc("linear", 2), F, phi, nb_iters=90)
    xs = grid.coords()[0]
    for r in (0.3, 0.5, 0.8):
        i = int(np.argmin(np.abs(xs - r * math.cos(math.pi / 4))))
        assert abs(F[i, i] - math.sin(math.atan2(xs[i], xs[i]))) < 3e-2
