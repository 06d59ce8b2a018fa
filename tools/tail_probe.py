#!/usr/bin/env python
"""Launch overhead of the headline stage kernel: stage time for 512 x 512 x nz grids (nz = 64 ... 1024) — the intercept of the
fit T = a·nz + b is what a launch costs beyond its share of the work (ramp-up + drain of the last workgroups).  GPU box.
LSM_STAGE_MC selects the march-chunk length."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsm_amd as lsm
from configs import timed


def run(nz, nxy=512, steps=8):
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, nz / nxy), (nxy, nxy, nz))
    ic = lsm.LazyMeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.35 * nz / nxy) ** 2) - 0.15, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.vortex_deformation(grid), lsm.WENO5()), lsm.EikonalReinitializationTerm()),
                              ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3())
    ms, kms, nl = timed(eq, steps)
    return ms, kms


if __name__ == "__main__":
    nzs = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]
    rows = []
    for nz in nzs:
        ms, kms = run(nz)
        rows.append((nz, kms, ms))
    x = np.array([r[0] for r in rows], float)
    y = np.array([r[1] for r in rows], float)
    a, b = np.polyfit(x, y, 1)
    print(json.dumps({"mc": os.environ.get("LSM_STAGE_MC", "default"), "tail": os.environ.get("LSM_STAGE_TAIL", "default"), "rows": [{"nz": r[0], "stage_ms": round(r[1], 4), "step_ms": round(r[2], 4)} for r in rows],
                      "fit_ms_per_plane": a, "fit_intercept_ms": b}))
