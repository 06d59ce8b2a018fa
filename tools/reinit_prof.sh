#!/bin/bash
# kernel trace and two PMC passes of reinitialize! (GPU box): tools/reinit_prof.sh <outdir-under-gpurun_out> [band n, default 256] [dense n, default 24]
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o ri -- python3 /root/repo/tools/reinit_bench.py ${2:-256} ${3:-24} > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o ri -- python3 /root/repo/tools/reinit_bench.py ${2:-256} ${3:-24} > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc2 -o ri -- python3 /root/repo/tools/reinit_bench.py ${2:-256} ${3:-24} > $OUT/pmc2.log 2>&1
cd /root/repo
python3 tools/band_step_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
cat $OUT/summary.log
