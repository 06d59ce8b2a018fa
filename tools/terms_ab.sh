#!/bin/bash
# usage: tools/terms_ab.sh name1 name2 ... — single-term stage timings (tools/configs.py terms) and config 3 for the in-tree
# library ("main") and kernel-variant libraries (GPU box)
for v in main "$@"; do
  if [ $v = main ]; then unset LSM_AMD_LIB; else export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/configs.py terms 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())[0]
for k,v in d.items():
    if isinstance(v,dict): print('  %-34s %.4f ms  %7.1f GB/s' % (k, v['stage_ms'], v['GBs_algorithmic']))
"
  timeout -k 10 200 python tools/configs.py 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())[0]; print('  config3: %.3f ms/step, stage %.4f ms' % (d['ms_per_step'], d['stage_ms']))
"
done
