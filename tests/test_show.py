"""The reference's text/plain `show` (test/test-show.jl) and the LevelSetEquation doctest
(src/levelsetequation.jl:33-57: spacing h = (0.04082, 0.04082), min = -0.2492, max = 1.75) through the host mirror.

CPU: grids, boundary conditions, host fields, integrators, and the equation tree over the TEST-ONLY oracle backend
(extrema from the oracle).  GPU: the same equation tree with the extrema computed by lsm_extrema on the device."""
import numpy as np
import pytest

import lsm_amd as lsm

DOCTEST = """LevelSetEquation
  ├─ equation: ϕₜ + 𝐮 ⋅ ∇ ϕ = 0
  ├─ time:     0.0
  ├─ integrator: RK2 (2nd order TVD Runge-Kutta, Heun's method)
  │  └─ cfl: 0.5
  ├─ state: MeshField on CartesianGrid in ℝ²
  │  ├─ domain:  [-1.0, 1.0] × [-1.0, 1.0]
  │  ├─ nodes:   50 × 50
  │  ├─ spacing: h = (0.04082, 0.04082)
  │  ├─ bc:     Neumann (all)
  │  ├─ valtype: Float64
  │  └─ values:  min = -0.2492,  max = 1.75
  ╰─"""


def _doctest_equation(**kw):
    """src/levelsetequation.jl:33-40, verbatim set-up."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (50, 50))
    phi = lsm.MeshField(lambda x: x[0] ** 2 + x[1] ** 2 - 0.5 ** 2, grid)
    u = lsm.MeshField(lambda x: (1.0 + 0 * x[0], 0.0 * x[0]), grid)
    return lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(u),), ic=phi, bc=lsm.NeumannBC(), **kw)


def test_grid_and_bc_show():
    s = lsm.show(lsm.CartesianGrid((0, 0), (1, 1), (10, 4)))          # test/test-show.jl:8-15
    assert s.startswith("CartesianGrid in ℝ²")
    assert "├─ domain:  [0.0, 1.0] × [0.0, 1.0]" in s and "├─ nodes:   10 × 4" in s
    assert "└─ spacing: h = (0.1111, 0.3333)" in s
    assert repr(lsm.PeriodicBC()) == "Periodic" and repr(lsm.NeumannBC()) == "Neumann"       # :25-31
    assert repr(lsm.LinearExtrapolationBC()) == "Linear extrapolation"
    assert repr(lsm.ExtrapolationBC(4)) == "Degree 4 extrapolation" and repr(lsm.SymmetryBC()) == "Symmetry"


def test_grid_basic_ops():
    """test/test-meshes.jl:5-13, and the index helpers of the export list (src/LevelSetMethods.jl:45-55) in 0-based form."""
    nx, ny = 100, 50
    a, b = (-1, 0), (1, 3)
    grid = lsm.CartesianGrid(a, b, (nx, ny))
    assert grid.size() == (nx, ny) and len(lsm.nodeindices(grid)) == nx * ny == len(grid)
    assert lsm.getnode(grid, 0, 0) == (-1.0, 0.0) and lsm.getnode(grid, nx - 1, ny - 1) == (1.0, 3.0)
    idx = lsm.nodeindices(lsm.CartesianGrid((0, 0), (1, 1), (3, 2)))
    assert idx == [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)]               # first index fastest, as CartesianIndices
    cells = lsm.cellindices(grid)
    assert len(cells) == (nx - 1) * (ny - 1) and cells[-1] == (nx - 2, ny - 2)    # src/meshes.jl:147
    lo, hi = lsm.getcell(grid, nx - 2, ny - 2)                                   # :183-197: node I to node I+1
    assert np.allclose(hi, (1.0, 3.0)) and np.allclose(np.subtract(hi, lo), grid.meshsize())
    with pytest.raises(ValueError):
        lsm.getcell(grid, nx - 1, 0)
    with pytest.raises(ValueError):
        lsm.getnode(grid, nx, 0)
    assert grid.compute_index((-5.0, 0.01)) == (0, 0) and grid.compute_index((0.999, 9.0)) == (nx - 2, ny - 2)   # :155-169, clamped
    assert np.array_equal(grid.grid1d(0), np.linspace(-1, 1, nx)) and len(grid.grid1d()) == 2
    phi = lsm.MeshField(lambda x: x[0] + x[1], grid)
    assert len(lsm.active_nodeindices(phi)) == nx * ny and lsm.update_band_(phi) is phi   # dense: all nodes, no-op (src/meshfield.jl:134,553)


def test_cartesian_cell_show():
    """test/test-show.jl:17-23"""
    s = lsm.show(lsm.getcell(lsm.CartesianGrid((0, 0), (1, 1), (10, 4)), 0, 0))
    assert s == "CartesianCell in ℝ²\n  ├─ lower corner: (0.0, 0.0)\n  └─ upper corner: (0.1111, 0.3333)"


def test_meshfield_construction_bc_and_copy():
    """test/test-meshfield.jl:7-42"""
    grid = lsm.CartesianGrid((-1.0, -1.0), (1.0, 1.0), (10, 5))
    f = lambda x: x[0] ** 2 + x[1] ** 2 - 0.5
    phi = lsm.MeshField(f, grid)
    assert phi.mesh is grid and isinstance(phi, lsm.MeshField) and not phi.has_boundary_conditions()
    assert phi.ndim == 2 and phi.values().shape == (10, 5)
    assert phi[2, 1] == pytest.approx(f(lsm.getnode(grid, 2, 1))) and phi.meshsize() == grid.meshsize()
    g1 = lsm.CartesianGrid((0.0,), (1.0,), (5,))
    p1 = lsm.MeshField(lambda x: x[0], g1)
    p1bc = p1.with_bc(((lsm.NeumannBC(), lsm.NeumannBC()),))
    assert not p1.has_boundary_conditions() and p1bc.has_boundary_conditions()
    assert p1bc.values() is p1.values()                                  # the underlying data is aliased
    assert p1bc[-2] == p1[0] and p1bc[6] == p1[4]
    g2 = lsm.CartesianGrid((0.0, 0.0), (1.0, 1.0), (5, 5))
    a, b = lsm.MeshField(lambda x: x[0] + x[1], g2), lsm.MeshField(lambda x: 0.0 * x[0], g2)
    b.copy_(a)
    assert np.array_equal(b.values(), a.values()) and b.values() is not a.values()   # copied, not aliased


def test_grid_from_meshsize_doctest():
    """src/meshes.jl:57-67: CartesianGrid((0, 0), (1, 1); meshsize = 0.3) has 5 × 5 nodes, h = 0.25 (cell count rounded up)."""
    g = lsm.CartesianGrid((0, 0), (1, 1), meshsize=0.3)
    assert lsm.show(g) == "CartesianGrid in ℝ²\n  ├─ domain:  [0.0, 1.0] × [0.0, 1.0]\n  ├─ nodes:   5 × 5\n  └─ spacing: h = (0.25, 0.25)"
    assert lsm.CartesianGrid((0, 0), (1, 2), meshsize=(0.5, 0.3)).n == (3, 8)
    for bad in (dict(meshsize=0.0), dict(meshsize=(0.1,)), dict(n=(3, 3), meshsize=0.1), dict()):
        with pytest.raises(ValueError):
            lsm.CartesianGrid((0, 0), (1, 1), **bad)
    # test/test-meshes.jl:15-41, "meshsize constructor"
    g = lsm.CartesianGrid((-1, -1), (1, 1), meshsize=0.5)                 # exact divisor: the spacing hits the request
    assert g.n == (5, 5) and np.allclose(g.meshsize(), (0.5, 0.5))
    assert tuple(g.getnode((0, 0))) == (-1.0, -1.0) and tuple(g.getnode((4, 4))) == (1.0, 1.0)   # corners unchanged
    g = lsm.CartesianGrid((0, 0), (1, 1), meshsize=0.3)                   # ceil rule: never coarser than requested
    assert g.n == (5, 5) and all(h <= 0.3 for h in g.meshsize()) and tuple(g.getnode((4, 4))) == (1.0, 1.0)
    g = lsm.CartesianGrid((0, 0), (2, 1), meshsize=(0.4, 0.3))            # per dimension
    assert g.n == (6, 5) and all(h <= m for h, m in zip(g.meshsize(), (0.4, 0.3)))
    for lc, hc, ms in (((0, 0), (1, 1), -0.1), ((0, 0), (1, 1), (0.1,)), ((1, 1), (0, 0), 0.1)):
        with pytest.raises(ValueError):
            lsm.CartesianGrid(lc, hc, meshsize=ms)


def test_meshfield_show():
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (5, 5))
    s = lsm.show(lsm.MeshField(lambda x: x[0] ** 2 + x[1] ** 2 - 0.5 ** 2, grid))           # :36-46
    assert s.startswith("MeshField on CartesianGrid in ℝ²")
    assert "├─ domain:  [-1.0, 1.0] × [-1.0, 1.0]" in s and "├─ nodes:   5 × 5" in s
    assert "├─ spacing: h = (0.5, 0.5)" in s and "bc:" not in s
    assert "├─ valtype: Float64" in s and "└─ values:  min = -0.25,  max = 1.75" in s
    assert s == ("MeshField on CartesianGrid in ℝ²\n  ├─ domain:  [-1.0, 1.0] × [-1.0, 1.0]\n  ├─ nodes:   5 × 5\n  ├─ spacing: h = (0.5, 0.5)\n"
                 "  ├─ valtype: Float64\n  └─ values:  min = -0.25,  max = 1.75")            # the doctest of src/meshfield.jl:192-206, whole
    s = lsm.show(lsm.MeshField(lambda x: (x[0], x[1]), grid))                              # :56-63
    assert "└─ valtype: SVector{2, Float64}" in s and "values" not in s and "bc:" not in s


def test_integrator_show():
    assert lsm.show(lsm.ForwardEuler()) == "ForwardEuler (1st order explicit)\n  └─ cfl: 0.5"       # :76-82
    assert lsm.show(lsm.RK2()) == "RK2 (2nd order TVD Runge-Kutta, Heun's method)\n  └─ cfl: 0.5"
    assert lsm.show(lsm.RK3()) == "RK3 (3rd order TVD Runge-Kutta)\n  └─ cfl: 0.5"
    assert lsm.show(lsm.ForwardEuler(cfl=0.3)) == "ForwardEuler (1st order explicit)\n  └─ cfl: 0.3"


def test_julia_float_printing():
    from lsm_amd.api import _jl_float, _sig4
    assert [_jl_float(x) for x in (1.0, 0.5, 1e-5, 1e-4, 123456.0, 1234567.0, 1e20, -0.0)] == \
        ["1.0", "0.5", "1.0e-5", "0.0001", "123456.0", "1.234567e6", "1.0e20", "-0.0"]
    assert [_sig4(x) for x in (2 / 49, -0.2492, 1.75, 12345.678, 1 / 9)] == ["0.04082", "-0.2492", "1.75", "12350.0", "0.1111"]


def test_equation_doctest_on_the_oracle_backend():
    from _oracle_backend import OracleBackend
    eq = _doctest_equation(backend_factory=lambda g, b, s: OracleBackend(g, b, s))
    assert lsm.show(eq) == DOCTEST
    assert repr(eq) == "LevelSetEquation(ϕₜ + 𝐮 ⋅ ∇ ϕ = 0, t=0.0)"                               # test/test-show.jl:102
    assert "├─ bc:     Neumann (all)" in lsm.show(eq.current_state())                          # :48-53


@pytest.mark.gpu
def test_equation_doctest_on_the_device():
    """The doctest's extrema and spacing with min/max reduced by lsm_extrema."""
    eq = _doctest_equation()
    assert lsm.show(eq) == DOCTEST
    lo, hi = eq.current_state().extrema()
    ref = eq.current_state().values()
    assert lo == ref.min() and hi == ref.max()
    eq3 = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(1.0), lsm.EikonalReinitializationTerm()), ic=lsm.MeshField(
        lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (12, 12, 12))),
        bc=(lsm.PeriodicBC(), (lsm.NeumannBC(), lsm.ExtrapolationBC(2)), lsm.SymmetryBC()), integrator=lsm.RK3())
    s = lsm.show(eq3)
    assert "├─ equation: ϕₜ + v|∇ϕ| + sign(ϕ) (|∇ϕ| - 1) = 0" in s and "├─ state: MeshField on CartesianGrid in ℝ³" in s
    assert "│  ├─ bc:     x: Periodic, y: Neumann ↔ Degree 2 extrapolation, z: Symmetry" in s and s.endswith("╰─")


@pytest.mark.gpu
def test_narrowband_show_on_the_device():
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (20, 20))                                    # test/test-show.jl:66-74
    phi = lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(1.0),), ic=lsm.NarrowBandMeshField(phi, nlayers=3), bc=lsm.NeumannBC())
    s = lsm.show(eq.current_state())
    assert s.startswith("NarrowBandMeshField on CartesianGrid in ℝ²")
    assert "├─ active:" in s and "(3-layer halo)" in s and "└─ values:" in s
