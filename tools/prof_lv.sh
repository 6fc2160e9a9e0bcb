#!/bin/bash
# rocprofv3 of the loop-verification leg (20 candidates on a street driven twice): kernel trace + PMC passes
#   usage (GPU box): tools/prof_lv.sh <tag>   -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-lv}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_loopverify.py lv"
KR="--kernel-include-regex (icp_|vg_|lv_)"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
timeout 300 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT --all-grids > $OUT/summary.txt 2>&1
head -30 $OUT/summary.txt | cut -c1-160
grep -E "^\{" $OUT/trace.log | cut -c1-700
rm -rf $OUT/*/*.db $OUT/*/*/*.db
