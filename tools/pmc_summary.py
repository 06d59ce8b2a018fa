#!/usr/bin/env python
"""Aggregate the PMC passes of tools/profile.sh into per-kernel averages per dispatch:
tools/pmc_summary.py <profile-dir> <out.json>   (counters as reported; FETCH_SIZE / WRITE_SIZE in KiB)"""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
out = collections.OrderedDict()
for f in sorted(glob.glob(src + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, disp in per.items():
        e = out.setdefault(k, collections.OrderedDict(dispatches=len(disp)))
        names = set(n for d in disp.values() for n in d)
        for n in sorted(names):
            e[n] = sum(d.get(n, 0.0) for d in disp.values()) / len(disp)
json.dump(out, open(dst, "w"), indent=1)
print("kernels:", len(out))
