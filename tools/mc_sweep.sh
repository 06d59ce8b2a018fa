#!/bin/bash
# usage: tools/mc_sweep.sh mc1 mc2 ... — headline bench per march-chunk length (GPU box), two rounds interleaved
for r in 1 2; do
for mc in "$@"; do
  LSM_STAGE_MC=$mc timeout -k 10 180 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mc=$mc', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
done
