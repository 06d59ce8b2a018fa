"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them bit for bit (CPU), and the HIP path must match them (GPU): bit for bit in STRICT
mode where the reference arithmetic is bit-defined, within the stated tolerance in FAST mode."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _vt(orc, grid, flat):
    n = grid.n
    out, o = [], 0
    for c in range(3):
        comp = []
        for d in range(3):
            comp.append(flat[o:o + n[d]])
            o += n[d]
        out.append(comp)
    return out


def test_oracle_reproduces_weno_and_ghost_goldens(orc):
    z = np.load(os.path.join(G, "weno5_core.npz"))
    got = np.array([orc.weno5_core(*row) for row in z["v"]])
    assert np.array_equal(got, z["out"])
    z = np.load(os.path.join(G, "ghosts_3d.npz"))
    grid = orc.Grid((0, 0, 0), (1, 1, 1), z["phi"].shape)
    lay = orc.layout(grid)
    specs = {"periodic": "periodic", "neumann": "neumann", "extrap2": ("extrapolation", 2), "symmetry": "symmetry",
             "mixed": [("neumann", ("extrapolation", 3)), "periodic", ("symmetry", "linear")]}
    for name, spec in specs.items():
        p = orc.fill_ghosts_padded(grid, orc.make_bc(spec, 3), lay, orc.to_padded(lay, 3, np.asfortranarray(z["phi"])))
        assert np.array_equal(p, z[name]), name


def test_oracle_reproduces_integration_goldens(orc):
    z = np.load(os.path.join(G, "headline_3d.npz"))
    grid = orc.Grid((0, 0, 0), (1, 1, 1), tuple(z["n"]))
    bc = orc.make_bc("neumann", 3)
    terms = [orc.advection(orc.separable(_vt(orc, grid, z["tables"]), orc.TIME_COS, 3.0)), orc.eikonal()]
    phi = np.asfortranarray(z["phi0"].copy())
    t = 0.0
    for dt_want in z["dts"]:
        dt = 0.5 * orc.compute_cfl(grid, bc, phi, terms, t)
        assert dt == dt_want
        orc.advance(orc.RK3, grid, bc, phi, terms, t, dt)
        t += dt
    assert np.array_equal(phi, z["phi3"])
    z = np.load(os.path.join(G, "mcf_3d.npz"))
    grid = orc.Grid((-1, -1, -1), (1, 1, 1), z["phi0"].shape)
    bc = orc.make_bc(("extrapolation", 2), 3)
    terms = [orc.normal_motion(orc.const(0.1)), orc.curvature(orc.const(-0.1))]
    phi = np.asfortranarray(z["phi0"].copy())
    t = 0.0
    for _ in range(3):
        dt = 0.5 * orc.compute_cfl(grid, bc, phi, terms, t)
        orc.advance(orc.RK3, grid, bc, phi, terms, t, dt)
        t += dt
    assert np.array_equal(phi, z["phi3"]) and t == float(z["t"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_hip_matches_headline_golden(mode):
    import lsm_amd as lsm
    z = np.load(os.path.join(G, "headline_3d.npz"))
    n = tuple(int(k) for k in z["n"])
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), n)
    ic = lsm.MeshField(z["phi0"], grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.vortex_deformation(grid), lsm.WENO5()), lsm.EikonalReinitializationTerm()),
                              ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3(), mode=mode)
    t = 0.0
    for dt_want in z["dts"]:
        dt = 0.5 * eq.compute_cfl(t)
        assert dt == dt_want          # Δt bitwise in both modes
        eq._advance(t, dt)
        t += dt
    got = eq.current_state().values()
    if mode == "strict":
        assert np.array_equal(got, z["phi3"])
    else:
        assert np.abs(got - z["phi3"]).max() <= 3 * 3e-13 * np.abs(z["phi3"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_hip_matches_zalesak_and_mcf_goldens(mode):
    import lsm_amd as lsm
    z = np.load(os.path.join(G, "zalesak_2d.npz"))
    grid = lsm.CartesianGrid((-1.5, -1.5), (1.5, 1.5), z["phi0"].shape)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()),), ic=lsm.MeshField(z["phi0"], grid),
                              bc=lsm.NeumannBC(), integrator=lsm.RK3(), mode=mode)
    t = 0.0
    for _ in range(4):
        dt = 0.5 * eq.compute_cfl(t)
        eq._advance(t, dt)
        t += dt
    adv = eq.current_state().values()
    # periodic PDE reinit: a second equation built from the current state (frozen sign), handed across with copy!
    cur = lsm.MeshField(adv, grid)
    re = lsm.LevelSetEquation(terms=(lsm.EikonalReinitializationTerm(cur),), ic=cur, bc=lsm.NeumannBC(), integrator=lsm.RK3(), mode=mode)
    for _ in range(2):
        re._advance(0.0, 0.5 * re.compute_cfl(0.0))
    out = re.current_state().values()
    tol = 0 if mode == "strict" else 1e-12
    assert np.abs(adv - z["adv4"]).max() <= tol * np.abs(z["adv4"]).max()
    assert np.abs(out - z["reinit2"]).max() <= tol * np.abs(z["reinit2"]).max()
    z = np.load(os.path.join(G, "mcf_3d.npz"))
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), z["phi0"].shape)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.1), lsm.CurvatureTerm(-0.1)), ic=lsm.MeshField(z["phi0"], grid),
                              bc=lsm.ExtrapolationBC(2), integrator=lsm.RK3(), mode=mode)
    t = 0.0
    for _ in range(3):
        dt = 0.5 * eq.compute_cfl(t)
        eq._advance(t, dt)
        t += dt
    assert t == float(z["t"])
    assert np.abs(eq.current_state().values() - z["phi3"]).max() <= 1e-12 * np.abs(z["phi3"]).max()   # curvature: pow/dot order
