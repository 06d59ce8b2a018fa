#!/bin/bash
# HBM read traffic of the headline stage kernel under an environment setting (GPU box): tools/traffic_probe.sh <outdir> VAR=VALUE ...
OUT=/root/repo/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  export "$kv"
  tag=$(echo $kv | tr '=' '_')
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$tag -o f -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  unset "${kv%%=*}"
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "stage_kernel<3, 2, 0, 0, 2" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
v=v[len(v)//2:]
print("$kv", "launches", len(v), "FETCH_SIZE avg KB", round(sum(v)/len(v)), "-> read GB (x2 on gfx950, 8-byte lanes)", round(2*sum(v)/len(v)*1024/1e9,3))
PY
done
