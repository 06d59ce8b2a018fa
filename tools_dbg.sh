#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 1 2; do
export LSM_GROW_DBG=$d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dbg$d -o band -- python tools_band_bench.py 512 band > gpurun_out/dbg$d.log 2>&1
echo "dbg $d"; python tools_kstats.py gpurun_out/dbg$d 3
done
