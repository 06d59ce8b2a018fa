#!/usr/bin/env python
"""Timing of reinitialize! on the device: a band field (sphere, nlayers 3) and a small dense field."""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lsm_amd as lsm


def run(n, band, reps=8):
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    f = lambda x: (x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.25          # not a distance function
    vals = lsm.LazyMeshField(f, grid).local_values(None)
    ic = lsm.MeshField(vals, grid)
    if band:
        ic = lsm.NarrowBandMeshField(ic, nlayers=3)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=ic, bc=lsm.ExtrapolationBC(2))
    st = eq.current_state()
    keep = st.buf.clone()
    ts = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(reps):
            st.buf.copy_(keep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lsm.reinitialize_(st)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    v = st.values()
    X = np.meshgrid(*grid.coords(), indexing="ij", sparse=True)
    exact = np.sqrt(X[0] ** 2 + X[1] ** 2 + X[2] ** 2) - 0.5
    m = st.active_mask() if band else np.ones(v.shape, bool)
    nodes = int(m.sum())
    return {"n": n, "band": band, "active_nodes": nodes, "ms": round(min(ts), 2), "Mnodes_s": round(nodes / min(ts) / 1e3, 2),
            "max_err": float(np.abs(v[m] - exact[m]).max())}


if __name__ == "__main__":
    out = [run(int(sys.argv[1]) if len(sys.argv) > 1 else 256, True), run(int(sys.argv[2]) if len(sys.argv) > 2 else 96, False)]
    print(json.dumps(out, indent=1))
