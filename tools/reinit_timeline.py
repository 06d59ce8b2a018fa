#!/usr/bin/env python
"""Timeline of one reinitialize! call from a tools/reinit_prof.sh directory: tools/reinit_timeline.py <dir> [which band call, default -2]"""
import csv
import json
import sys

d = sys.argv[1]
k = json.load(open(d + "/summary/pmc_per_dispatch.json"))["kernels"]
for name in k:
    if "search" in name or "newton" in name or "sample" in name:
        e = k[name]
        print(name[:60], {x: e[x] for x in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU") if x in e})
rows = list(csv.DictReader(open(d + "/trace/ri_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "reinit_band_nodes" in r["Kernel_Name"]]
i0 = idx[int(sys.argv[2]) if len(sys.argv) > 2 else -2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 26]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} v{r['VGPR_Count']:>4} {r['Kernel_Name'][:70]}")
