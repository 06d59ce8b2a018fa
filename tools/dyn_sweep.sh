#!/bin/bash
# usage: tools/dyn_sweep.sh d1 d2 ... — headline bench per dynamic-tail setting (LSM_STAGE_TAIL_DYN = % spare tail workgroups, 0 = static tail), three rounds interleaved (GPU box)
for r in 1 2 3; do
for d in "$@"; do
  LSM_STAGE_TAIL_DYN=$d timeout -k 10 180 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dyn=$d', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
done
