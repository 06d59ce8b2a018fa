#!/bin/bash
# timing experiments on the band kernels (LSM_BAND_EXP bits; results of non-zero runs are WRONG, only the times count)
mkdir -p gpurun_out/r3a
for e in "$@"; do
  LSM_BAND_EXP=$e python tools/band_probe.py 768 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp $e', d['ms_per_step'])"
done
