#!/bin/bash
# bench config 2 (2-D 2048^2) for each kernel-variant library
for v in "$@"; do
  echo -n "$v: "; LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$v.so timeout -k 10 120 python tools/configs.py 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print(round(d['advect_stage_ms'],4), round(d['advect_ms_per_step'],4), round(d['reinit_stage_ms'],4))"
done
echo -n "default: "; timeout -k 10 120 python tools/configs.py 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print(round(d['advect_stage_ms'],4), round(d['advect_ms_per_step'],4), round(d['reinit_stage_ms'],4))"
