#!/bin/bash
# usage: tools/tail_sweep.sh t1 t2 ... — headline bench per graded-tail chunk length (LSM_STAGE_TAIL, 0 = off), two rounds interleaved (GPU box)
for r in 1 2; do
for t in "$@"; do
  LSM_STAGE_TAIL=$t timeout -k 10 180 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail=$t', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
done
