#!/usr/bin/env python
"""Generates tests/golden/*.npz with the CPU oracle (oracle/lsm_oracle.c).

The reference (Julia) cannot run in the build container and ships no golden vectors of its own
(SURVEY.md §8c), so these fixtures freeze the ORACLE's outputs — after the oracle passed the
reference's analytic tests (tests/test_oracle_reference_tests.py) — to (a) detect any later drift
of the oracle and (b) give the GPU tests inputs/outputs that do not depend on the oracle library
being rebuilt identically.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402


def case_weno_core():
    rng = np.random.default_rng(100)
    v = rng.standard_normal((64, 5)) * 10.0 ** rng.integers(-6, 6, size=(64, 1))
    v[0] = 0.0
    v[1] = [1, 1, 1, 1, 1]
    v[2] = [0, 0, 1, 1, 1]          # a kink
    out = np.array([orc.weno5_core(*row) for row in v])
    np.savez(os.path.join(HERE, "weno5_core.npz"), v=v, out=out)


def case_ghosts():
    rng = np.random.default_rng(101)
    shape = (7, 6, 5)
    grid = orc.Grid((0, 0, 0), (1, 1, 1), shape)
    phi = np.asfortranarray(rng.standard_normal(shape))
    specs = {"periodic": "periodic", "neumann": "neumann", "extrap2": ("extrapolation", 2), "symmetry": "symmetry",
             "mixed": [("neumann", ("extrapolation", 3)), "periodic", ("symmetry", "linear")]}
    out = {"phi": phi}
    lay = orc.layout(grid)
    for name, spec in specs.items():
        p = orc.fill_ghosts_padded(grid, orc.make_bc(spec, 3), lay, orc.to_padded(lay, 3, phi))
        out[name] = p
    np.savez(os.path.join(HERE, "ghosts_3d.npz"), **out)


def _vortex_tables(grid):
    x, y, z = grid.coords()
    s2 = lambda a: np.sin(np.pi * a) * np.sin(np.pi * a)
    s = lambda a: np.sin(2 * np.pi * a)
    return [[2 * s2(x), s(y), s(z)], [-s(x), s2(y), s(z)], [-s(x), s(y), s2(z)]]


def case_headline_3d():
    """Config 4 in miniature: vortex-deformation WENO5 advection + Eikonal reinit, RK3, NeumannBC."""
    n = (20, 18, 16)
    grid = orc.Grid((0, 0, 0), (1, 1, 1), n)
    bc = orc.make_bc("neumann", 3)
    phi0 = grid.sample(lambda X, Y, Z: np.sqrt((X - 0.35) ** 2 + (Y - 0.35) ** 2 + (Z - 0.35) ** 2) - 0.15)
    tables = _vortex_tables(grid)
    terms = [orc.advection(orc.separable(tables, orc.TIME_COS, 3.0)), orc.eikonal()]
    phi = phi0.copy(order="F")
    dts, t = [], 0.0
    for _ in range(3):
        dt = 0.5 * orc.compute_cfl(grid, bc, phi, terms, t)
        orc.advance(orc.RK3, grid, bc, phi, terms, t, dt)
        t += dt
        dts.append(dt)
    np.savez(os.path.join(HERE, "headline_3d.npz"), phi0=phi0, phi3=phi, dts=np.array(dts),
             tables=np.concatenate([np.concatenate(c) for c in tables]), n=np.array(n))


def case_zalesak_2d():
    """Config 2 in miniature: Zalesak disk, rigid rotation, WENO5 + RK3, NeumannBC; then 2 RK3
    steps of frozen-sign Eikonal reinitialisation (docs/src/example-zalesak.md:21-26)."""
    n = (48, 48)
    grid = orc.Grid((-1.5, -1.5), (1.5, 1.5), n)
    bc = orc.make_bc("neumann", 2)
    disk = grid.sample(lambda X, Y: np.hypot(X + 0.75, Y) - 0.5)
    rec = grid.sample(lambda X, Y: np.maximum(np.abs(X + 0.75) - 0.1, np.abs(Y + 0.25) - 0.5))
    phi0 = np.asfortranarray(np.maximum(disk, -rec))
    phi = phi0.copy(order="F")
    t = 0.0
    for _ in range(4):
        dt = 0.5 * orc.compute_cfl(grid, bc, phi, [orc.advection(orc.rotation())], t)
        orc.advance(orc.RK3, grid, bc, phi, [orc.advection(orc.rotation())], t, dt)
        t += dt
    adv = phi.copy(order="F")
    s0 = orc.eikonal_sign(grid, phi)
    for _ in range(2):
        dt = 0.5 * orc.compute_cfl(grid, bc, phi, [orc.eikonal(s0)], 0.0)
        orc.advance(orc.RK3, grid, bc, phi, [orc.eikonal(s0)], 0.0, dt)
    np.savez(os.path.join(HERE, "zalesak_2d.npz"), phi0=phi0, adv4=adv, reinit2=phi)


def case_mcf_3d():
    """Config 3 in miniature: sphere under NormalMotion(0.1) + Curvature(-0.1), RK3, ExtrapolationBC(2)."""
    n = (18, 18, 18)
    grid = orc.Grid((-1, -1, -1), (1, 1, 1), n)
    bc = orc.make_bc(("extrapolation", 2), 3)
    phi0 = grid.sample(lambda X, Y, Z: np.sqrt(X * X + Y * Y + Z * Z) - 0.5)
    terms = [orc.normal_motion(orc.const(0.1)), orc.curvature(orc.const(-0.1))]
    phi = phi0.copy(order="F")
    t = 0.0
    for _ in range(3):
        dt = 0.5 * orc.compute_cfl(grid, bc, phi, terms, t)
        orc.advance(orc.RK3, grid, bc, phi, terms, t, dt)
        t += dt
    np.savez(os.path.join(HERE, "mcf_3d.npz"), phi0=phi0, phi3=phi, t=np.array(t))


if __name__ == "__main__":
    case_weno_core()
    case_ghosts()
    case_headline_3d()
    case_zalesak_2d()
    case_mcf_3d()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
