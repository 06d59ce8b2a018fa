"""BASELINE config 1: 2-D 256² circle SDF ‖x − (0.8, 0)‖ − 0.5 on (−2,−2)–(2,2), rigid rotation
u = (−x₂, x₁), first-order Upwind + ForwardEuler at cfl 0.5, NeumannBC, one revolution (tf = 2π) — the set-up of
test/test-levelsetequation.jl:197-198 at the config's resolution.

The step count is analytic: the CFL minimum 1/((|x₂| + |x₁|)/h) is attained at the corners, Δt_cfl = h/4 = 1/255, so
Δt = 0.5/255 and the loop of src/timestepping.jl:104-116 takes ⌈2π·510⌉ = 3205 steps, the last one cut to land on tf.

CPU: the oracle (the config is "CPU plumbing").  GPU twin: the same run through integrate! → lsm_advance_fe."""
import math

import numpy as np
import pytest

N = 256
STEPS = math.ceil(2 * math.pi * 510)      # 3205


def _phi0(x, y):
    return np.hypot(x - 0.8, y) - 0.5


def _oracle_run(orc, threads=8):
    grid = orc.Grid((-2.0, -2.0), (2.0, 2.0), (N, N))
    phi0 = grid.sample(_phi0)
    phi = phi0.copy(order="F")
    orc.set_threads(min(threads, orc.max_threads()))
    try:
        steps, t, _ = orc.integrate(orc.FE, grid, orc.make_bc("neumann", 2), phi, [orc.advection(orc.rotation(), orc.SCHEME_UPWIND)],
                                    2 * math.pi)
    finally:
        orc.set_threads(1)
    return phi0, phi, steps, t


def test_config1_on_the_oracle(orc):
    assert STEPS == 3205
    grid = orc.Grid((-2.0, -2.0), (2.0, 2.0), (N, N))
    phi0 = grid.sample(_phi0)
    terms = [orc.advection(orc.rotation(), orc.SCHEME_UPWIND)]
    assert orc.compute_cfl(grid, orc.make_bc("neumann", 2), phi0, terms) == 1.0 / (2.0 / grid.meshsize(0) + 2.0 / grid.meshsize(1))
    phi0, phi, steps, t = _oracle_run(orc)
    assert steps == STEPS and t == 2 * math.pi                       # integrate! lands exactly on tf
    near = np.abs(phi0) < 0.1
    assert np.abs(phi - phi0)[near].max() < 0.15                     # the circle came back (first-order smearing: 0.106)
    inside0, inside = (phi0 < 0).sum(), (phi < 0).sum()
    assert abs(inside - inside0) / inside0 < 0.3                     # first-order upwind loses 24 % of the area in one revolution
    phi1 = _oracle_run(orc, threads=1)[1]
    assert np.array_equal(phi, phi1)                                 # thread count does not change a bit


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_config1_on_the_device(orc, mode):
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (N, N))
    ic = lsm.MeshField(lambda x: _phi0(x[0], x[1]), grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.Upwind()),), ic=ic, bc=lsm.NeumannBC(),
                              integrator=lsm.ForwardEuler(), mode=mode)
    steps = []
    lsm.integrate_(eq, 2 * math.pi, posthook=lambda e: steps.append(e.current_time()))
    assert len(steps) == STEPS and eq.current_time() == 2 * math.pi
    phi0, want, _, _ = _oracle_run(orc)
    assert np.array_equal(ic.vals, phi0)
    got = eq.current_state().values()
    if mode == "strict":
        assert np.array_equal(got, want)                             # 3205 ForwardEuler steps, bit for bit
    else:
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
