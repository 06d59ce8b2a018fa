#!/usr/bin/env python
"""Print the last N kernel launches (name, µs) of a rocprofv3 kernel trace CSV under a directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -30:]:
    print(f'{r["Kernel_Name"][:60]:60s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us')
