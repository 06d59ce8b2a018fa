#!/usr/bin/env python
"""Timing of extend_along_normals! on small and large grids (a launch-bound loop on small ones: 2 launches per sweep)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lsm_amd as lsm


def run(n, ndim, iters=50, reps=5):
    grid = lsm.CartesianGrid((-1.0,) * ndim, (1.0,) * ndim, (n,) * ndim)
    phi = lsm.MeshField(lambda x: np.sqrt(sum(c * c for c in x)) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=phi, bc=lsm.NeumannBC())
    st = eq.current_state()
    F = st.copy()
    lsm.extend_along_normals_(F, st, nb_iters=iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lsm.extend_along_normals_(F, st, nb_iters=iters)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    return {"grid": f"{n}^{ndim}", "nb_iters": iters, "ms": round(ms, 3), "us_per_sweep": round(ms / iters * 1e3, 2)}


if __name__ == "__main__":
    print(json.dumps([run(128, 2), run(512, 2), run(48, 3), run(256, 3)], indent=1))
