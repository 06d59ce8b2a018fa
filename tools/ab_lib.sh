#!/bin/bash
# interleaved: variants/libhiplsm_A.so vs the tree's library on config 5
for i in 1 2 3; do
  for v in A T; do
    if [ $v = A ]; then export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_A.so; else unset LSM_AMD_LIB; fi
    echo -n "$v "; python tools/band_probe.py 768 20 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
