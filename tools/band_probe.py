#!/usr/bin/env python
"""Config 5 on one device: tile / halo statistics of the band and ms per step (GPU box).  `LSM_BAND_MC` selects the brick depth."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsm_amd as lsm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = np.float64 if os.environ.get("LSM_PROBE_F64") else np.float32
grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
f = lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5
vals = lsm.LazyMeshField(f, grid).local_values(None).astype(dt)
eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01)),
                          ic=lsm.NarrowBandMeshField(lsm.MeshField(vals, grid, dtype=dt), nlayers=3), bc=lsm.NeumannBC(), integrator=lsm.RK3())
del vals
st = eq.state
mc = st.MC
t = st.tiles.cpu().numpy().astype(bool)
nbx, nby, nbm = (n + 31) // 32, (n + 7) // 8, (n + mc - 1) // mc
T = t.reshape((nbm, nby, nbx))
W = np.zeros_like(T)
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            W |= np.roll(T, (dz, dy, dx), axis=(0, 1, 2))
# runs of active bricks along the march axis
runs = int((T & ~np.concatenate([np.zeros_like(T[:1]), T[:-1]], axis=0)).sum())
info = {"mc": mc, "tiles": int(T.size), "active": int(T.sum()), "work": int(W.sum()), "march_runs": runs,
        "halo_entries": int(st._hcount.item()), "active_nodes": st.active_count()}


def one(tc):
    eq._update_terms(eq.state, tc)
    step = eq.integrator.cfl * eq.compute_cfl(tc)
    eq._advance(tc, step)
    eq.update_band()
    return tc + step


tc = 0.0
for _ in range(2):
    tc = one(tc)
torch.cuda.synchronize()
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.1:
    for _ in range(4):
        tc = one(tc)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tc = one(tc)
torch.cuda.synchronize()
info["ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
print(json.dumps(info))
