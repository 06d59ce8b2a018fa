#!/bin/bash
# interleaved A/B of reinitialize! (256^3 band): the variants/libhiplsm_<X>.so named on the command line vs the tree's library (T)
cd /root/repo
for i in 1 2; do
  for v in "$@" T; do
    unset LSM_AMD_LIB
    if [ $v != T ]; then export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$v.so; fi
    echo -n "$v "; python tools/reinit_bench.py 256 32 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print([(x['n'], x['ms'], x['max_err']) for x in r])"
  done
done
