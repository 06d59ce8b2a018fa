#!/usr/bin/env python
"""Cost of the slab driver on ONE device (no neighbours, so no transfer): the per-rank workload of
`bench.py --gpus N` (2n x 2n x n/4 nodes) stepped (a) by lsm_advance_rk3, (b) by the stage-by-stage slab driver,
(c) by the slab driver with the boundary-first split used to overlap the halo exchange."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import bench
import lsm_amd as lsm


def run(n, comm, force, steps=6):
    eq, _, _ = bench.build_equation(lsm, n, comm, 0, "fast")
    eq._force_overlap = force
    tc = 0.0
    for _ in range(2):
        tc = bench.one_step(eq, tc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tc = bench.one_step(eq, tc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if __name__ == "__main__":
    base = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n = (2 * base, 2 * base, base // 4)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    out = {"grid": n, "advance_ms": round(run(n, None, False), 3), "slab_driver_ms": round(run(n, dist.group.WORLD, False), 3),
           "slab_driver_split_ms": round(run(n, dist.group.WORLD, True), 3)}
    dist.destroy_process_group()
    print(json.dumps(out))
