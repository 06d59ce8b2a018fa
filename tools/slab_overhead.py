#!/usr/bin/env python
"""Cost of the slab step on ONE device (GPU box).  Grid = two ranks' share of `bench.py --gpus N` (2n x 2n x n/2 nodes):
  (a) one whole-grid handle stepped by lsm_advance_rk3;
  (b) the same grid as TWO slab handles of this process (LSM_COMM_LOCAL, one thread per rank) stepped by lsm_advance_rk3
      with the library's slab step: interface planes first, plane exchange (device-to-device copies here, xGMI on a real
      node) overlapped with the interior, Δt all-reduce — both ranks share the one GPU, so the time is the sum of their
      work plus everything the decomposition adds;
  (c) the same with the overlap switched off (stage -> ghost fill -> exchange).
(b)/(a) - 1 is the overhead of the decomposition itself, transfers excluded."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import lsm_amd as lsm


def run_whole(n, steps):
    eq, _, _ = bench.build_equation(lsm, n, None, 0, "fast")
    tc = 0.0
    for _ in range(12):          # out of the device's idle power state (≈50 ms of load)
        tc = bench.one_step(eq, tc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tc = bench.one_step(eq, tc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run_slabs(n, steps, overlap, world=2):
    g = lsm.LocalGroup(world)
    bar = threading.Barrier(world)
    out, errs = [0.0] * world, []

    def rank(r):
        try:
            eq, _, _ = bench.build_equation(lsm, n, g.rank(r), 0, "fast")
            eq.backend.comm_set_overlap(overlap)
            tc = 0.0
            for _ in range(12):
                tc = bench.one_step(eq, tc)
            torch.cuda.synchronize()
            bar.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                tc = bench.one_step(eq, tc)
            torch.cuda.synchronize()
            bar.wait()
            out[r] = (time.perf_counter() - t0) / steps * 1e3
        except BaseException as e:   # noqa: BLE001
            errs.append(repr(e))
            bar.abort()
            g._barrier.abort()

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise RuntimeError(errs)
    return max(out)


if __name__ == "__main__":
    base = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = 6
    n = (2 * base, 2 * base, base // 2)
    a = run_whole(n, steps)
    # two rank threads share the device: how their launches interleave varies from run to run (7.5–8.1 ms seen for the same build) —
    # the median of three
    bs = sorted(run_slabs(n, steps, True) for _ in range(3))
    cs = sorted(run_slabs(n, steps, False) for _ in range(3))
    b, c = bs[1], cs[1]
    print(json.dumps({"grid": n, "whole_grid_ms": round(a, 3), "two_local_slabs_overlap_ms": round(b, 3), "two_local_slabs_plain_ms": round(c, 3),
                      "decomposition_overhead": round(b / a - 1, 4), "overlap_runs_ms": [round(x, 3) for x in bs], "plain_runs_ms": [round(x, 3) for x in cs],
                      "note": "one device: the two ranks share it, transfers are device-to-device copies; medians of three runs"}))
