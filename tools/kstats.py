#!/usr/bin/env python
"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f'{r["Name"][:72]:72s} {r["Calls"]:>6s} {int(r["TotalDurationNs"]) / 1e6:10.3f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]}%')
