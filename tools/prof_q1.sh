#!/bin/bash
# rocprofv3 of the one-launch single-query path (sc_q1.hip): kernel trace + the PMC passes, each in its own run.
# usage (on the GPU box, through gpurun): tools/prof_q1.sh <tag> [sizes]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-q1}
SIZES=${2:-10000,100000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_q1.py --modes q1 --sizes $SIZES --k 1 --nq 1 --reps 100"
KR="--kernel-include-regex sc_q1_kernel"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS --kernel-trace -d $OUT/pmc6 -o pmc6 -- $CMD > $OUT/pmc6.log 2>&1
timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
timeout 300 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT --all-grids > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -80
