#!/bin/bash
# usage: tools/variants.sh name1 name2 ... — bench each kernel-variant library (GPU box)
for v in "$@"; do
  LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$v.so timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
