#!/usr/bin/env python
"""profiles/rN/pmc_terms.json from the two SQ counter passes of `tools/configs.py terms` (tools/profile_round.sh):
tools/pmc_terms_summary.py <profile-dir> <out.json>

Per stage kernel of the single-term / fused-term family (512^3, one ForwardEuler stage): per-dispatch averages of the counters and
what DESIGN.md §3.1 derives from them — vector instructions per node, the share of the launch the vector pipe is held, LDS and
wait shares, resident waves.  GRBM_GUI_ACTIVE counts cycles on each of the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles summed over
the SIMDs (x4 = cycles a pipe was held; 1024 SIMDs)."""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
NAMES = {"stage_kernel2<1, 0, 0, 0": "upwind adv (const) [128x8 pairs]", "stage_kernel2<0, 1, 0, 0": "NormalMotion (const) [128x8 pairs]",
         "stage_kernel2<0, 0, 0, 2": "Eikonal (current sign) [128x8 pairs]", "<3, 1, 0, 0, 0": "upwind adv (const)", "<3, 2, 0, 0, 0": "WENO5 adv",
         "<3, 0, 1, 0, 0": "NormalMotion (const)", "<3, 0, 0, 1, 0": "Curvature (const)", "<3, 0, 0, 0, 2": "Eikonal (current sign)",
         "<3, 0, 1, 1, 0": "NormalMotion + Curvature (config 3)", "<3, 2, 0, 1, 0": "WENO5 adv (rotation) + Curvature",
         "<3, 2, 0, 0, 2": "WENO5 adv (vortex) + Eikonal (headline)"}
avg = collections.defaultdict(dict)
for sub in ("terms_pmc_1", "terms_pmc_2"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "stage_kernel" in r["Kernel_Name"]:
                per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for (k, _), cs in per.items():
            for n, v in cs.items():
                by[k][n].append(v)
        for k, cs in by.items():
            for n, vs in cs.items():
                vs = vs[len(vs) // 3:]            # the first launches of a kernel run while the device ramps up
                avg[k][n] = sum(vs) / len(vs)
nodes = 512 ** 3
out = {"command": "tools/configs.py terms under rocprofv3 --pmc (two passes: issue / wait counters, LDS / memory counters); 512^3, one ForwardEuler stage per launch",
       "kernels": {}}
for k, c in sorted(avg.items()):
    key = next((n for n in NAMES if n in k), None)
    if key is None:
        continue
    d = {"kernel": k.split("(")[0].replace("void lsm::fast_math::", ""), "counters_per_dispatch": {n: round(v, 1) for n, v in sorted(c.items())}}
    if "SQ_INSTS_VALU" in c:
        d["valu_lane_instructions_per_node"] = round(c["SQ_INSTS_VALU"] * 64.0 / nodes, 1)
    if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        d["valu_pipe_held_fraction"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / cyc, 3)
    if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"]:
        d["wave_cycles_per_busy_cu_cycle"] = round(c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CU_CYCLES"], 2)
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        d["wave_cycles_waiting_for_an_instruction_fraction"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if "SQ_WAIT_INST_LDS" in c and "SQ_WAVE_CYCLES_2" not in c and "SQ_WAIT_ANY" in c and c["SQ_WAIT_ANY"]:
        d["lds_share_of_waits"] = round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAIT_ANY"], 3)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_ACTIVE_INST_LDS"):
        d["lds_bank_conflict_cycles_per_lds_active_cycle"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"], 3)
    out["kernels"][NAMES[key]] = d
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x not in ("counters_per_dispatch", "kernel")} for k, v in out["kernels"].items()}, indent=1))
