#!/usr/bin/env python
"""Copies what is judged from a tools/profile_round.sh output directory (gpurun_out/..., scratch) into profiles/rN/ (tracked):
tools/collect_profiles.py <profile-dir> <profiles/rN>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def stats(sub, name):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, name))


stats("trace", "kernel_stats_bench_steps3.csv")
stats("trace_c2", "kernel_stats_config2.csv")
stats("trace_c3", "kernel_stats_config3.csv")
stats("trace_c5", "kernel_stats_config5.csv")
stats("trace_reinit", "kernel_stats_reinit.csv")
for f in ("pmc_per_dispatch.json", "bench_default_run.json", "bench_config2.json", "bench_config3.json", "bench_config5.json",
          "clock_probe.json", "slab_overhead.json", "reinit_bench.json", "terms.json", "timeline_probe.json", "timeline_probe_no_tail.json",
          "tail_probe.jsonl", "pmc_terms.json", "bench_local8.json", "bench_config5r.json", "reinit_bench_512.json", "reinit_timeline_256.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))

# HBM traffic of every term's stage kernel (tools/configs.py terms under --pmc FETCH_SIZE / --pmc WRITE_SIZE)
def per_kernel(sub, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "stage_kernel" in r["Kernel_Name"]:
                per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, _), v in per.items():
            out[k].append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}


F, W = per_kernel("terms_pmc_f", "FETCH_SIZE"), per_kernel("terms_pmc_w", "WRITE_SIZE")
try:
    terms = json.load(open(os.path.join(src, "terms.json")))[0]
except Exception:   # noqa: BLE001
    terms = {}
names = {"stage_kernel2<1, 0, 0, 0": "upwind adv (const)", "stage_kernel2<0, 1, 0, 0": "NormalMotion (const)", "stage_kernel2<0, 0, 0, 2": "Eikonal (current sign)",
         "<3, 1, 0, 0, 0": "upwind adv (const)", "<3, 0, 1, 0, 0": "NormalMotion (const)", "<3, 0, 0, 1, 0": "Curvature (const)",
         "<3, 0, 0, 0, 2": "Eikonal (current sign)", "<3, 0, 1, 1, 0": "NormalMotion + Curvature",
         "<3, 2, 0, 1, 0": "WENO5 adv (rotation) + Curvature", "<3, 2, 0, 0, 2": "WENO5 adv (vortex) + Eikonal"}
alg = 512 ** 3 * 16 / 1e9
lines = ["# tools/configs.py terms (512^3, one ForwardEuler stage = read psi + write: %.3f GB algorithmic) under" % alg,
         "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); read = 2 x FETCH_SIZE (8-byte-per-lane reads count half on gfx950)",
         "# %-44s %8s %8s %8s %8s %22s %9s" % ("stage kernel", "read GB", "write GB", "total GB", "x alg.", "stage ms (un-profiled)", "HBM TB/s")]
for k in sorted(F):
    key = next((n for n in names if n in k), None)
    if key is None or k not in W:
        continue
    pair = "stage_kernel2" in k          # own nodes by 16-byte accesses, halo by 8-byte ones: the factor 2 calibrated on 8-byte reads is applied unchanged
    rd, wr = 2 * F[k] * 1024 / 1e9, W[k] * 1024 / 1e9     # (with factor 1 the reads would come out below the algorithmic 1.07 GB)
    ms = terms.get(names[key], {}).get("stage_ms")
    tile = "128x8, two nodes per thread" if pair else ("64x8" if ", 64, 8, " in k else "32x8")
    lines.append("  %-44s %8.2f %8.2f %8.2f %8.2f %22s %9s" % (names[key] + " [" + tile + "]", rd, wr, rd + wr, (rd + wr) / alg,
                                                              "%.3f" % ms if ms else "-", "%.2f" % ((rd + wr) / ms) if ms else "-"))
open(os.path.join(dst, "terms_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
