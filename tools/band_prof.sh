#!/bin/bash
# kernel trace and PMC passes of config 5's band step (GPU box): tools/band_prof.sh <outdir-under-gpurun_out>
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c5 -- python3 /root/repo/tools/band_probe.py 768 40 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o c5 -- python3 /root/repo/tools/band_probe.py 768 6 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc2 -o c5 -- python3 /root/repo/tools/band_probe.py 768 6 > $OUT/pmc2.log 2>&1
cd /root/repo
python tools/kstats.py $OUT/trace 10
# a third pass: LDS bank conflicts of the brick stage
if [ -n "$2" ]; then
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --output-format csv -d $OUT/pmc3 -o c5 -- python3 /root/repo/tools/band_probe.py 768 6 > $OUT/pmc3.log 2>&1
cd /root/repo
fi
