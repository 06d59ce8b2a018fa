#!/usr/bin/env python
"""Narrow band vs dense step time (config-5-like: sphere, rigid-rotation WENO5 advection + curvature, RK3)."""
import json
import sys
import time

import numpy as np
import torch

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import lsm_amd as lsm


def run(n, band, nlayers=3, steps=10):
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    f = lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5
    terms = (lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01))
    if band:
        ic = lsm.NarrowBandMeshField(lsm.MeshField(lsm.LazyMeshField(f, grid).local_values(None), grid), nlayers=nlayers)
    else:
        ic = lsm.LazyMeshField(f, grid)
    eq = lsm.LevelSetEquation(terms=terms, ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3())
    tc = 0.0

    def one(tc):
        eq._update_terms(eq.state, tc)
        dt = eq.integrator.cfl * eq.compute_cfl(tc)
        eq._advance(tc, dt)
        eq.update_band()
        return tc + dt
    for _ in range(2):
        tc = one(tc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tc = one(tc)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"n": n, "band": band, "ms_per_step": round(ms, 3)}
    if band:
        c = eq.state.active_count()
        out.update(active_nodes=c, active_fraction=round(c / n ** 3, 4), active_tiles=int(eq.state.tiles.sum().item()),
                   tiles=int(eq.state.tiles.numel()), Mcells_s_active=round(c / ms / 1e3, 1), Mcells_s_grid=round(n ** 3 / ms / 1e3, 1))
    else:
        out.update(Mcells_s_grid=round(n ** 3 / ms / 1e3, 1))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    res = ([run(n, False)] if which in ("both", "dense") else []) + ([run(n, True)] if which in ("both", "band") else [])
    print(json.dumps(res, indent=1))
