#!/bin/bash
# usage: tools/profile_variant.sh <outdir> [variant]  — rocprofv3 kernel trace + SQ/GRBM PMC passes of the bench for the
# in-tree library or a kernel-variant library (levelsetmethods.jl_amd/variants/libhiplsm_<variant>.so) (GPU box)
set -u
OUT=$1
V=${2:-main}
mkdir -p $OUT
export TMPDIR=/tmp
if [ $V != main ]; then export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$V.so; fi
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.json 2> $OUT/pmc2.err
python3 tools/pmc_summary.py $OUT $OUT/pmc_per_dispatch.json
