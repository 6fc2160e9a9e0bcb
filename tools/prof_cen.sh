#!/bin/bash
# Kernel trace + two PMC passes of the cen2019 chain (batched device entry).  Usage: prof_cen.sh <tag>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/tools/bench_cen2019.py 5 64"
KR="--kernel-include-regex cen_"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/bench_pmc1.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/bench_pmc2.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2
cat $OUT/summary.txt
