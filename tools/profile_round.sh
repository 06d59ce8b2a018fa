#!/bin/bash
# usage: tools/profile_round.sh <outdir> [round, default r4]  (GPU box) — everything profiles/<round>/ is made from, for the build in the tree:
#   rocprofv3 kernel trace + PMC passes of the default bench (separate passes per counter group, as the guide prescribes),
#   the in-kernel clock of the diagnostic build, the bench line itself, configs 2/3/5, the single-term stages with their
#   HBM traffic, the slab-overhead run and the reinit run.  tools/make_profile_summary.py turns the CSVs into
#   pmc_per_dispatch.json (keyed by the sha256 of the kernel sources).
set -u
OUT=$1
RND=${2:-r4}
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.json 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $B > $OUT/pmc3.json 2> $OUT/pmc3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- $B > $OUT/pmc4.json 2> $OUT/pmc4.err
python3 tools/clock_probe.py 512 2.5 > $OUT/clock_probe.json 2> $OUT/clock_probe.err
python3 tools/timeline_probe.py 512 > $OUT/timeline_probe.json 2> $OUT/timeline_probe.err
LSM_STAGE_TAIL=0 python3 tools/timeline_probe.py 512 > $OUT/timeline_probe_no_tail.json 2> $OUT/timeline_probe_no_tail.err
(cd tools && python3 tail_probe.py && LSM_STAGE_TAIL=0 python3 tail_probe.py) > $OUT/tail_probe.jsonl 2> $OUT/tail_probe.err
python3 tools/make_profile_summary.py $OUT $OUT/pmc_per_dispatch.json > $OUT/summary.log 2>&1
mkdir -p profiles/$RND && cp $OUT/pmc_per_dispatch.json profiles/$RND/pmc_per_dispatch.json   # the bench line below reads it (same build, same box)
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
for c in 2 3 5; do python3 bench.py --config $c > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c3 -- python3 bench.py --config 3 --steps 4 > /dev/null 2> $OUT/trace_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- python3 bench.py --config 2 --steps 20 > /dev/null 2> $OUT/trace_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -- python3 bench.py --config 5 --steps 6 > /dev/null 2> $OUT/trace_c5.err
python3 tools/configs.py terms > $OUT/terms.json 2> $OUT/terms.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/terms_pmc_f -- python3 tools/configs.py terms > /dev/null 2> $OUT/terms_pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/terms_pmc_w -- python3 tools/configs.py terms > /dev/null 2> $OUT/terms_pmc_w.err
# issue / wait / LDS counters of every member of the stage-kernel family (DESIGN.md §3.1's "issue-bound" / "latency-bound" sentences cite these)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/terms_pmc_1 -- python3 tools/configs.py terms > /dev/null 2> $OUT/terms_pmc_1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/terms_pmc_2 -- python3 tools/configs.py terms > /dev/null 2> $OUT/terms_pmc_2.err
python3 tools/pmc_terms_summary.py $OUT $OUT/pmc_terms.json > $OUT/pmc_terms.log 2>&1
python3 tools/slab_overhead.py > $OUT/slab_overhead.json 2> $OUT/slab_overhead.err
python3 bench.py --transport local --gpus 8 --steps 6 --warmup 2 --prewarm 4 --no-cpu-baseline > $OUT/bench_local8.json 2> $OUT/bench_local8.err
python3 tools/reinit_bench.py > $OUT/reinit_bench.json 2> $OUT/reinit_bench.err
python3 tools/reinit_bench.py 512 32 > $OUT/reinit_bench_512.json 2> $OUT/reinit_bench_512.err
python3 bench.py --config 5r > $OUT/bench_config5r.json 2> $OUT/bench_config5r.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reinit -- python3 tools/reinit_bench.py > /dev/null 2> $OUT/trace_reinit.err
tools/reinit_prof.sh ${OUT#*gpurun_out/}/reinit_prof > /dev/null 2>&1 && python3 tools/reinit_timeline.py $OUT/reinit_prof > $OUT/reinit_timeline_256.txt 2>&1   # (reinit_prof.sh writes under /root/repo/gpurun_out/)
ls $OUT
