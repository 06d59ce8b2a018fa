"""extend_along_normals! (next row of SURVEY.md §8f, rank 3): the reference's own tests
(test/test-velocityextension.jl:18-105) restated on the oracle (CPU) and, on the GPU, parity of
lsm_extend_along_normals with the oracle plus the same tests through the host API."""
import numpy as np
import pytest


def _plane_case(orc):
    grid = orc.Grid((-1.0, -1.0), (1.0, 1.0), (81, 61))
    phi = grid.sample(lambda x, y: x + 0 * y)
    d = min(grid.meshsize())
    y = grid.coords()[1]
    frozen = np.abs(phi) <= d
    F = np.asfortranarray(np.where(frozen, np.sin(np.pi * y)[None, :], 0.0))
    Fref = np.repeat(np.sin(np.pi * y)[None, :], 81, axis=0)
    return grid, phi, F, frozen, Fref


def test_oracle_extend_along_normals_plane(orc):
    """test/test-velocityextension.jl:18-42"""
    grid, phi, F, frozen, Fref = _plane_case(orc)
    seed = F.copy()
    orc.extend_along_normals(grid, orc.make_bc("linear", 2), F, phi, nb_iters=150, frozen=frozen, cfl=0.45)
    assert np.abs(F - Fref).max() < 0.08
    assert np.array_equal(F[frozen], seed[frozen])


def _circle_case(orc):
    grid = orc.Grid((-1.0, -1.0), (1.0, 1.0), (121, 121))
    R = 0.55
    phi = grid.sample(lambda x, y: np.sqrt(x * x + y * y) - R)
    d = min(grid.meshsize())
    frozen = np.abs(phi) <= 1.1 * d
    X, Y = np.meshgrid(*grid.coords(), indexing="ij")
    r = np.sqrt(X * X + Y * Y)
    v = np.asfortranarray(np.where(frozen, Y / np.maximum(r, np.finfo(float).eps), 0.0))
    return grid, phi, v, frozen, d


def _n_dot_grad(orc, grid, bc, phi, v, frozen, d):
    """mean |n·∇v| over the 5Δ band (test/test-velocityextension.jl:68-82)"""
    tot, cnt = 0.0, 0
    for i in range(grid.n[0]):
        for j in range(grid.n[1]):
            if abs(phi[i, j]) <= 5.0 * d and not frozen[i, j]:
                gx = orc.deriv(grid, bc, phi, "D0", (i, j), 0)
                gy = orc.deriv(grid, bc, phi, "D0", (i, j), 1)
                nrm = np.hypot(gx, gy)
                if not np.isfinite(1 / nrm):
                    continue
                vx = orc.deriv(grid, bc, v, "D0", (i, j), 0)
                vy = orc.deriv(grid, bc, v, "D0", (i, j), 1)
                tot += abs(gx / nrm * vx + gy / nrm * vy)
                cnt += 1
    return tot / cnt, cnt


def test_oracle_extend_along_normals_circle_periodic(orc):
    """test/test-velocityextension.jl:44-84"""
    grid, phi, v, frozen, d = _circle_case(orc)
    bc = orc.make_bc("periodic", 2)
    seed = v.copy()
    orc.extend_along_normals(grid, bc, v, phi, nb_iters=100, frozen=frozen, cfl=0.45)
    assert np.array_equal(v[frozen], seed[frozen])
    mean, cnt = _n_dot_grad(orc, grid, bc, phi, v, frozen, d)
    assert cnt > 100 and mean < 0.12


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_gpu_extend_along_normals_matches_oracle(orc, mode):
    import lsm_amd as lsm
    cases = []
    grid, phi, F, frozen, _ = _plane_case(orc)
    cases.append(("plane", grid, "linear", lsm.LinearExtrapolationBC(), phi, F, frozen, 60))
    grid, phi, v, frozen, _ = _circle_case(orc)
    cases.append(("circle", grid, "periodic", lsm.PeriodicBC(), phi, v, frozen, 40))
    g3 = orc.Grid((-1, -1, -1), (1, 1, 1), (30, 28, 26))
    p3 = g3.sample(lambda x, y, z: np.sqrt(x * x + y * y + z * z) - 0.5)
    f3 = g3.sample(lambda x, y, z: np.sin(2 * x) * np.cos(3 * y) + z)
    cases.append(("sphere3d-bandrule", g3, "neumann", lsm.NeumannBC(), p3, f3, None, 25))
    for name, og, obc, lbc, phi, F, frozen, iters in cases:
        want = F.copy(order="F")
        orc.extend_along_normals(og, orc.make_bc(obc, og.ndim), want, phi, nb_iters=iters, frozen=frozen, cfl=0.45)
        lg = lsm.CartesianGrid(og.lc, og.hc, og.n)
        eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(phi, lg), bc=lbc, mode=mode)
        Fd = eq.current_state().copy()
        Fd.copy_(lsm.MeshField(F, lg))
        out = lsm.extend_along_normals_(Fd, eq.current_state(), nb_iters=iters, frozen=frozen, cfl=0.45)
        assert out is Fd
        got = Fd.values()
        if mode == "strict":
            assert np.array_equal(got, want), (name, np.abs(got - want).max())
        else:
            assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), name


@pytest.mark.gpu
def test_gpu_extend_along_normals_argument_checks():
    """test/test-velocityextension.jl:86-105"""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (41, 41))
    phi = lsm.MeshField(lambda x: x[0] + x[1], grid)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=lsm.LinearExtrapolationBC())
    F = eq.current_state().copy()
    assert lsm.extend_along_normals_(F, eq.current_state(), nb_iters=5) is F
    with pytest.raises(ValueError):
        lsm.extend_along_normals_(F, eq.current_state(), frozen=np.zeros((40, 41), dtype=bool))
    with pytest.raises(ValueError):
        lsm.extend_along_normals_(F, eq.current_state(), cfl=0.0)
    with pytest.raises(ValueError):
        lsm.extend_along_normals_(np.zeros((41, 41)), eq.current_state())
