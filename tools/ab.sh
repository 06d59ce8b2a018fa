#!/bin/bash
# usage: tools/ab.sh [rounds] name1 name2 ... — bench the in-tree library ("main") and kernel-variant libraries
# (levelsetmethods.jl_amd/variants/libhiplsm_<name>.so) alternately, `rounds` times each (GPU box): interleaving keeps
# box-to-box and thermal drift out of the comparison.
R=$1; shift
for r in $(seq 1 $R); do
  for v in main "$@"; do
    if [ $v = main ]; then unset LSM_AMD_LIB; else export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$v.so; fi
    timeout -k 10 180 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
