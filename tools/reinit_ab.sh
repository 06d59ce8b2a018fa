#!/bin/bash
# interleaved A/B of reinitialize! (256^3 band, small dense): variants/libhiplsm_A.so vs the tree's library
cd /root/repo
for i in 1 2; do
  for v in A T; do
    unset LSM_AMD_LIB
    if [ $v = A ]; then export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_A.so; fi
    echo -n "$v "; python tools/reinit_bench.py 256 48 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print([(x['n'], x['ms'], x['max_err']) for x in r])"
  done
done
