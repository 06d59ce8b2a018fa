#!/usr/bin/env python
"""Config 5 on one device: where the HOST spends a band step (GPU box) — wall time of each call of the step loop, and the step."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsm_amd as lsm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
vals = lsm.LazyMeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid).local_values(None).astype(np.float32)
eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01)),
                          ic=lsm.NarrowBandMeshField(lsm.MeshField(vals, grid, dtype=np.float32), nlayers=3), bc=lsm.NeumannBC(), integrator=lsm.RK3())
del vals
acc = {"update_terms": 0.0, "compute_cfl": 0.0, "advance (enqueue)": 0.0, "update_band (enqueue + status wait)": 0.0}
pc = time.perf_counter


def one(tc, rec):
    t0 = pc(); eq._update_terms(eq.state, tc)
    t1 = pc(); step = eq.integrator.cfl * eq.compute_cfl(tc)
    t2 = pc(); eq._advance(tc, step)
    t3 = pc(); eq.update_band()
    t4 = pc()
    if rec:
        for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += d
    return tc + step


tc = 0.0
for _ in range(60):
    tc = one(tc, False)
torch.cuda.synchronize()
t0 = pc()
for _ in range(steps):
    tc = one(tc, True)
torch.cuda.synchronize()
el = pc() - t0
print(json.dumps({"ms_per_step": round(el / steps * 1e3, 4), "host_us_per_step": {k: round(v / steps * 1e6, 1) for k, v in acc.items()}}))
