#!/usr/bin/env python
"""profiles/rN/band_step/ from a tools/band_prof.sh output directory (trace + three PMC passes of tools/band_probe.py 768):
tools/band_step_summary.py <band_prof-dir> <profiles/rN/band_step>
pmc_per_dispatch.json: per-dispatch averages of the lsm kernels of the band step, and for the kernels that use the LDS the ratio the
round-3 review asked for — SQ_LDS_BANK_CONFLICT cycles per SQ_ACTIVE_INST_LDS cycle."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
f = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if f:
    shutil.copy(f[0], os.path.join(dst, "kernel_stats.csv"))
out = collections.OrderedDict()
for tag in ("pmc1", "pmc2", "pmc3"):
    for f in sorted(glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        for r in csv.DictReader(open(f)):
            if "lsm::" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, disp in per.items():
            if len(disp) < 6:                       # set-up kernels (the dense first build of the band)
                continue
            e = out.setdefault(k, collections.OrderedDict())
            e["dispatches_" + tag] = len(disp)
            for n in sorted(set(n for d in disp.values() for n in d)):
                e[n] = round(sum(d.get(n, 0.0) for d in disp.values()) / len(disp), 1)
for k, e in out.items():
    if e.get("SQ_ACTIVE_INST_LDS"):
        e["lds_bank_conflict_cycles_per_lds_active_cycle"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_ACTIVE_INST_LDS"], 3)
    if e.get("SQ_BUSY_CU_CYCLES") and e.get("SQ_ACTIVE_INST_LDS") is not None:
        e["lds_cycles_per_busy_cu_cycle"] = round((e.get("SQ_ACTIVE_INST_LDS", 0.0) + e.get("SQ_LDS_BANK_CONFLICT", 0.0)) / e["SQ_BUSY_CU_CYCLES"], 3)
json.dump({"command": "tools/band_prof.sh <dir> lds (tools/band_probe.py 768: config 5 on one device, float32 band); per-dispatch averages of three counter passes",
           "kernels": out}, open(os.path.join(dst, "pmc_per_dispatch.json"), "w"), indent=1)
for k, e in out.items():
    print(k[:70], {x: e[x] for x in e if x.startswith("lds_") or x in ("SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS")})
