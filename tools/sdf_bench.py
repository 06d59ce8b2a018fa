#!/usr/bin/env python
"""NewtonSDF queries far from the interface: signed distance at every node of a coarser grid covering (and exceeding) the domain."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lsm_amd as lsm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid), bc=lsm.ExtrapolationBC(2))
sdf = lsm.NewtonSDF(eq.current_state())
ax = np.linspace(-1.3, 1.3, m)
X = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1).reshape(-1, 3)
sdf(X[:64])
torch.cuda.synchronize()
t0 = time.perf_counter()
d = sdf(X)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
print(json.dumps({"grid": n, "queries": len(X), "ms": round(ms, 2), "Mqueries_s": round(len(X) / ms / 1e3, 3), "max_err": float(np.abs(d - (np.linalg.norm(X, axis=1) - 0.5)).max())}))
