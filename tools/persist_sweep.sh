#!/bin/bash
# usage: tools/persist_sweep.sh "p:t" ... — headline bench per (LSM_STAGE_PERSIST, LSM_STAGE_TAIL) pair, two rounds interleaved (GPU box)
for r in 1 2; do
for c in "$@"; do
  p=${c%%:*}; t=${c#*:}
  LSM_STAGE_PERSIST=$p LSM_STAGE_TAIL=$t timeout -k 10 180 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$p tail=$t', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
done
