#!/bin/bash
# usage: tools/profile_extra.sh <outdir> [variant] — instruction-cache / scalar / memory-level PMC passes of the bench (GPU box)
set -u
OUT=$1
V=${2:-main}
mkdir -p $OUT
export TMPDIR=/tmp
if [ $V != main ]; then export LSM_AMD_LIB=$PWD/levelsetmethods.jl_amd/variants/libhiplsm_$V.so; fi
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc5 -- $B > $OUT/pmc5.json 2> $OUT/pmc5.err
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc6 -- $B > $OUT/pmc6.json 2> $OUT/pmc6.err
python3 tools/pmc_summary.py $OUT $OUT/pmc_extra.json
