#!/bin/bash
# interleaved A/B of an environment switch on config 5's step (tools/band_probe.py 768): tools/band_ab.sh VAR A B [rounds]
# usage on the GPU box: tools/band_ab.sh LSM_BAND_BRICKS 0 1 3
var=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq $n); do
  for v in $a $b; do
    env $var=$v python tools/band_probe.py 768 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['ms_per_step'],4))"
  done
done
