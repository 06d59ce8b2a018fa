#!/usr/bin/env python
"""Latency of reinitialize! on small band fields (the sizes of the reference's own narrow-band tests): ms per call."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lsm_amd as lsm

out = []
for n in ((100, 100), (200, 200), (50, 50, 50)):
    nd = len(n)
    grid = lsm.CartesianGrid((-1,) * nd, (1,) * nd, n)
    ic = lsm.NarrowBandMeshField(lsm.MeshField(lambda x: sum(c ** 2 for c in x) - 0.25, grid), nlayers=3)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=ic, bc=lsm.ExtrapolationBC(2))
    st = eq.current_state()
    keep = st.buf.clone()
    ts = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(30):
            st.buf.copy_(keep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lsm.reinitialize_(st)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    out.append({"n": list(n), "band_nodes": st.active_count(), "ms_min": round(min(ts), 3), "ms_median": round(float(np.median(ts)), 3)})
print(json.dumps(out))
