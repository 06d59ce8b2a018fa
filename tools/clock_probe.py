#!/usr/bin/env python
"""In-kernel clock of the fused stage kernel (GPU box): runs the headline bench loop on the DIAGNOSTIC build
(levelsetmethods.jl_amd/variants/libhiplsm_stamp.so, `make -C levelsetmethods.jl_amd/csrc stamp`), whose stage kernels
stamp s_memtime / s_memrealtime around their plane loop, for >= 2 s of back-to-back steps, and prints the median over
the workgroups of the last launches: clock = Δs_memtime ÷ Δs_memrealtime × 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
usage: python tools/clock_probe.py [n] [seconds]   -> one JSON line"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSM_AMD_LIB"] = os.path.join(ROOT, "levelsetmethods.jl_amd", "variants", "libhiplsm_stamp.so")
import numpy as np
import torch

import lsm_amd as lsm
from bench import build_equation, one_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
eq, grid, vel = build_equation(lsm, (n, n, n), None, 0, "fast")
lib = eq.backend.lib
lib.lsm_debug_stamp.restype = C.c_int
lib.lsm_debug_stamp.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
tc = 0.0
for _ in range(3):
    tc = one_step(eq, tc)
assert lib.lsm_debug_stamp(eq.backend.h, 1, None, None) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        tc = one_step(eq, tc)
    steps += 20
torch.cuda.synchronize()
el = time.perf_counter() - t0
clk, us = C.c_double(), C.c_double()
assert lib.lsm_debug_stamp(eq.backend.h, 0, C.byref(clk), C.byref(us)) == 0
print(json.dumps({"grid": n, "steps": steps, "ms_per_step_stamped_build": el / steps * 1e3, "in_kernel_clock_ghz": round(clk.value, 4),
                  "plane_loop_us_median": round(us.value, 2), "note": "diagnostic build (-DLSM_STAMP); stamps go to a buffer nothing else reads"}))
