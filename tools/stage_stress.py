#!/usr/bin/env python
"""Randomised stress of the fused stage kernel against the CPU oracle (GPU box): random shapes (partial tiles, tiny
dimensions), boundary conditions, term lists, base modes and coefficient kinds, FAST and STRICT arithmetic.
usage: tools/stage_stress.py [ncases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import _hip
from oracle import oracle as orc
from test_gpu_parity import _rand_field, _run_stage

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(ncases):
    nd = int(rng.integers(1, 4))
    shape = tuple(int(rng.integers(5, (300, 120, 70)[nd - 1])) for _ in range(nd))
    bc = [("neumann", ("extrapolation", 3)), "periodic", ("symmetry", "linear"), ("extrapolation", 2), "neumann"]
    bcspec = [bc[int(rng.integers(0, len(bc)))] for _ in range(nd)]
    mode = "fast" if rng.random() < 0.7 else "strict"
    c = _hip.Case(shape, bcspec, mode=mode)
    phi = _rand_field(shape, int(rng.integers(0, 1000)), smooth=rng.random() < 0.7)
    coords = c.grid.coords()
    kinds = []
    r = rng.random()
    if r < 0.25:
        adv = ("const", tuple(float(v) for v in rng.standard_normal(nd)))
    elif r < 0.5 and nd >= 2:
        adv = ("rot", float(rng.standard_normal()), 0.1, -0.2)
    elif r < 0.8:
        adv = ("sep", [[rng.standard_normal(len(x)) for x in coords] for _ in range(nd)], ("cos", 3.0) if rng.random() < 0.5 else None)
    else:
        adv = ("field", [np.asfortranarray(rng.standard_normal(shape)) for _ in range(nd)])
    pool = [("adv", adv, "weno5" if rng.random() < 0.7 else "upwind"), ("nm", ("const", (float(rng.standard_normal()),))),
            ("curv", ("const", (-abs(float(rng.standard_normal())) * 0.1,))), ("eik", None if rng.random() < 0.6 else phi)]
    k = int(rng.integers(1, 4))
    specs = [pool[i] for i in rng.permutation(4)[:k]]
    base_mode = int(rng.integers(0, 4))
    got, want, _, _ = _run_stage(c, orc, specs, phi, base_mode, t=float(rng.random()))
    scale = max(np.abs(want).max(), 1e-300)
    err = np.abs(got - want).max() / scale
    has_curv = any(s[0] == "curv" for s in specs)
    ok = err <= 1e-13 if (mode == "fast" or has_curv) else np.array_equal(got, want)
    worst = max(worst, err)
    if not ok:
        print("FAIL", case, shape, bcspec, mode, [s[0] for s in specs], adv[0], base_mode, err)
        sys.exit(1)
print(f"{ncases} random stage cases ok, worst relative deviation {worst:.2e}")
