#!/usr/bin/env python
"""profiles/rN/pmc_per_dispatch.json from the rocprofv3 passes of tools/profile_r2.sh:
tools/make_profile_summary.py <profile-dir> <out.json>

Per-dispatch averages of the dominant kernel (the fused WENO5 + Eikonal stage), the figures bench.py derives its
`roofline.traffic` and `roofline_compute` from, and the key that ties them to a build: the sha256 of the kernel sources
(`lsm_amd._lib.source_hash()`; the GPU box has no .git) plus the workload (grid, GPUs, mode)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsm_amd

src, dst = sys.argv[1], sys.argv[2]
KEY = "stage_kernel<3, 2, 0, 0, 2"
counters = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> value
for f in sorted(glob.glob(src + "/pmc[0-9]*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if KEY in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    names = set(n for d in per.values() for n in d)
    for n in names:
        vals = [d[n] for d in per.values() if n in d]
        counters["avg"][n] = sum(vals) / len(vals)
        counters["n"][n] = len(vals)
c = dict(counters["avg"])
dur_ns = None
for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KEY in r["Name"]:
            dur_ns = float(r["AverageNs"])
n = 512
wave_planes = n ** 3 / 64.0
out = {
    "csrc_sha256": lsm_amd._lib.source_hash(), "grid": [n, n, n], "n_gpus": 1, "mode": "fast",
    "kernel": "lsm::fast_math::stage_kernel<3, 2, 0, 0, 2, 32, 8, 64, double, 2, false>",
    "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (one rocprofv3 pass per counter group)",
    "dispatches_per_pass": int(counters["n"].get("SQ_INSTS_VALU", 0)),
    "counters_per_dispatch": {k: c[k] for k in sorted(c)},
    "avg_duration_ns_kernel_trace": dur_ns,
}
if "SQ_INSTS_VALU" in c:
    out["valu_per_wave_plane"] = c["SQ_INSTS_VALU"] / wave_planes
    # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU issue summed over the SIMDs: ×4 = cycles a SIMD's vector pipe was held
    out["valu_busy_cycles_per_wave_plane"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / wave_planes
if "GRBM_GUI_ACTIVE" in c and dur_ns:
    out["clock_ghz_grbm"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / dur_ns, 4)
    out["valu_utilisation_profiled"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_traffic_bytes_per_launch"] = round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)
    out["traffic_note"] = "2*FETCH_SIZE + WRITE_SIZE, KiB -> B: FETCH_SIZE counts half of the 8-byte-per-lane reads on gfx950 (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact"
    out["algorithmic_bytes_per_launch"] = n ** 3 * 64.0 / 3.0
try:
    cp = json.load(open(os.path.join(src, "clock_probe.json")))
    out["in_kernel_clock_ghz"] = cp["in_kernel_clock_ghz"]
    out["in_kernel_clock_note"] = "s_memtime/s_memrealtime around the plane loop, diagnostic build, median over workgroups after 2.5 s of steps (tools/clock_probe.py)"
except Exception as e:   # noqa: BLE001
    out["in_kernel_clock_ghz"] = None
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
