#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench + separate PMC passes (SQ counters in two sets,
# FETCH_SIZE, WRITE_SIZE -- never combined with each other or with other traces).  usage: tools/profile_round.sh r02b
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
# per-kernel durations and counters are only attributable in SERIAL stream order: the default step overlaps three HIP streams
# (weight gradients, discriminator step, second decoder branch); the last line below is the default (concurrent) bench
# (the secondary configs of the bench line -- scp, 10 s replay, CDiffuSE -- stay out of the per-step statistics; rocprofv3 --pmc also
# crashed in the CDiffuSE leg)
export SE_NO_WGRAD_STREAM=1 SE_NO_D_OVERLAP=1 SE_NO_BRANCH_STREAM=1 SE_BENCH_NO_SECONDARY=1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/sq1 -o p -- $B > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -o p -- $B > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p -- $B > $OUT/write.log 2>&1
unset SE_NO_WGRAD_STREAM SE_NO_D_OVERLAP SE_NO_BRANCH_STREAM SE_BENCH_NO_SECONDARY
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err

# keep only the summaries (gpurun copies back at most 64 MiB)
python3 $R/tools/pmc_summary.py $OUT/pmc_summary.json $OUT/sq1/p_results.db $OUT/sq2/p_results.db $OUT/fetch/p_results.db $OUT/write/p_results.db > $OUT/pmc_summary.txt 2>&1
cp $OUT/trace/*/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write
ls -la $OUT
