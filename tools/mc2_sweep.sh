#!/bin/bash
# usage: tools/mc2_sweep.sh "libs" "mcs" "sizes" — 2-D stage kernel: rows per march chunk (LSM_STAGE_MC2) × tile width
# (variant libraries built with EXTRA=-DLSM_TX2=…) × grid size; BASELINE config 2's advection and reinit equations (GPU box)
for n in $3; do
for v in $1; do
  if [ $v = main ]; then unset LSM_AMD_LIB; else export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$v.so; fi
  for mc in $2; do
    LSM_STAGE_MC2=$mc timeout -k 10 120 python3 -c "
import sys; sys.path.insert(0,'tools'); import configs
r = configs.config2(n=$n, steps=40)
print('n=$n lib=$v mc2=$mc advect stage %.4f step %.4f | reinit stage %.4f step %.4f' % (r['advect_stage_ms'], r['advect_ms_per_step'], r['reinit_stage_ms'], r['reinit_ms_per_step']))
" 2>/dev/null
  done
done
done
