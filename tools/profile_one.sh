#!/bin/bash
# PMC passes (SQ counters in two sets + FETCH_SIZE + WRITE_SIZE, each its own run) over ONE micro-benchmark command.
# usage: tools/profile_one.sh TAG python3 tools/attn_bwd_ab.py time      (the program itself after the tag: no env / bash -c hops)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
CMD="$1 $R/$2 ${@:3}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_summary.json $OUT/sq1/p_results.db $OUT/sq2/p_results.db $OUT/fetch/p_results.db $OUT/write/p_results.db > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write
cat $OUT/pmc_summary.txt
