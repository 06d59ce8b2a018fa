#!/bin/bash
# usage: tools/layout_ab.sh — row-aligned padded layout (LSM_LAYOUT_ALIGN, default 1) against the compact one: headline bench, the
# single-term stages, config 3 (GPU box)
for r in 1 2; do
for al in 0 1; do
  LSM_LAYOUT_ALIGN=$al timeout -k 10 180 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('align=$al', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
done
for al in 0 1; do
  echo "== align=$al"
  LSM_LAYOUT_ALIGN=$al timeout -k 10 300 python tools/configs.py terms 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())[0]
for k,v in d.items():
    if isinstance(v,dict): print('  %-34s %.4f ms  %7.1f GB/s' % (k, v['stage_ms'], v['GBs_algorithmic']))
"
  LSM_LAYOUT_ALIGN=$al timeout -k 10 200 python tools/configs.py 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())[0]; print('  config3: %.3f ms/step, stage %.4f ms' % (d['ms_per_step'], d['stage_ms']))
"
done
