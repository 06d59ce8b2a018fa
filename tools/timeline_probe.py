#!/usr/bin/env python
"""Timeline of ONE launch of the headline stage kernel from the diagnostic build's absolute s_memrealtime stamps (100 MHz):
how many workgroups are inside their plane loop over time, how long a plane loop lasts in the first / middle / last
round, how long the drain is.  GPU box; `make -C levelsetmethods.jl_amd/csrc stamp` first.  usage: timeline_probe.py [n]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSM_AMD_LIB"] = os.path.join(ROOT, "levelsetmethods.jl_amd", "variants", "libhiplsm_stamp.so")
import numpy as np
import torch

import lsm_amd as lsm
from bench import build_equation, one_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eq, grid, vel = build_equation(lsm, (n, n, n), None, 0, "fast")
lib = eq.backend.lib
lib.lsm_debug_stamp.restype = C.c_int
lib.lsm_debug_stamp.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.lsm_debug_stamp_raw.restype = C.c_int
lib.lsm_debug_stamp_raw.argtypes = [C.c_void_p, C.c_void_p]
tc = 0.0
for _ in range(40):            # warm: clocks settle
    tc = one_step(eq, tc)
assert lib.lsm_debug_stamp(eq.backend.h, 1, None, None) == 0
tc = one_step(eq, tc)
raw = np.zeros(4 * 16384, dtype=np.uint64)
assert lib.lsm_debug_stamp_raw(eq.backend.h, raw.ctypes.data) == 0
raw = raw.reshape(16384, 4)
ok = raw[:, 3] > 0
# the stamps of the LAST launch only: spare tail workgroups that find no ticket leave without stamping, so their slots still hold
# an earlier launch's times
ok &= raw[:, 1].astype(np.int64) > int(raw[ok, 3].max()) - 140000        # 1.4 ms in 10 ns ticks
be, st, en = raw[ok, 1].astype(np.int64), raw[ok, 2].astype(np.int64), raw[ok, 3].astype(np.int64)
blk = np.arange(16384)[ok]
t0 = be.min()
be, st, en = (be - t0) * 0.01, (st - t0) * 0.01, (en - t0) * 0.01            # µs
dur = en - st
order = np.argsort(st)
total = en.max()
# concurrency in 10 µs bins
edges = np.arange(0, total + 10, 10.0)
conc = np.array([((st <= e) & (en > e)).sum() for e in edges])
full = conc.max()
drain_start = edges[np.where(conc >= 0.95 * full)[0][-1]]
out = {
    "grid": n, "workgroups": int(ok.sum()), "launch_span_us": round(float(total), 1), "max_concurrent": int(full),
    "loop_us_first_1280": round(float(dur[order[:1280]].mean()), 1), "loop_us_middle": round(float(dur[order[2560:5120]].mean()), 1),
    "loop_us_last_1280": round(float(dur[order[-1280:]].mean()), 1),
    "prologue_us (kernel entry to loop start) first_1280 / rest": [round(float((st - be)[order[:1280]].mean()), 2), round(float((st - be)[order[1280:]].mean()), 2)],
    "slot_turnaround_us (a loop end to the next kernel entry, matched in time order)": round(float(np.median(np.sort(be)[1280:] - np.sort(en)[:len(en) - 1280])), 2),
    "resident_share": {"in_loop": round(float(dur.sum() / (1280 * total)), 4), "prologue": round(float((st - be).sum() / (1280 * total)), 4)},
    "drain_us (from the last moment with >= 95 % of the peak concurrency to the end)": round(float(total - drain_start), 1),
    "lost_workgroup_us_in_drain": round(float(((full - conc[edges >= drain_start]) * 10.0).sum()), 0),
    "concurrency_every_50us": [int(c) for c in conc[::5]],
    "finish_spread_us_of_last_1280": round(float(en[np.argsort(en)[-1280:]].min()), 1),
    "loop_us_by_start_order (deciles)": [round(float(d.mean()), 1) for d in np.array_split(dur[order], 10)],
    "clock_ghz_by_start_order (deciles)": [round(float(c.mean()), 3) for c in np.array_split((raw[ok, 0].astype(np.float64) / (dur * 100.0) * 0.1)[order], 10)],
    "per_xcd_end_us": [round(float(en[blk % 8 == x].max()), 1) for x in range(8)],
}
print(json.dumps(out))
