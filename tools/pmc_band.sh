#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/bandpmc -o band -- python tools/band_bench.py 512 band > gpurun_out/bandpmc.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/bandpmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
last = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    last.setdefault(k, {})
    last[k].setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in last.items():
    if "band" not in k and "stage" not in k: continue
    ids = sorted(d, key=int)
    print(k)
    for i in ids[-2:]:
        print("   ", {c: int(v) for c, v in d[i].items()})
PY
