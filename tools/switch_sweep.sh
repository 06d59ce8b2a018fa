#!/bin/bash
# usage: tools/switch_sweep.sh "ENV=val ..." ... — headline bench (steady state: bench.py pre-warms) for each environment setting, three
# rounds interleaved with the default (GPU box)
for r in 1 2 3; do
for e in "" "$@"; do
  env $e timeout -k 10 180 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${e:-default}', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
done
