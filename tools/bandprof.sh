#!/bin/bash
# GPU box: narrow-band tests, then a kernel trace of the band step (tools/band_bench.py) with per-kernel stats
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_narrowband.py -x -q > gpurun_out/nb.log 2>&1 || { tail -30 gpurun_out/nb.log; exit 1; }
tail -2 gpurun_out/nb.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bandprof -o band -- python tools/band_bench.py ${1:-512} band > gpurun_out/bandprof.log 2>&1
python tools/kstats.py gpurun_out/bandprof 16
grep -E "ms_per_step|active" gpurun_out/bandprof.log
