#!/bin/bash
# rocprofv3 kernel trace + six PMC passes (each its own run) of ANY command, summarised to gpurun_out/prof_<tag>/summary.txt
#   usage (GPU box): tools/prof_six.sh <tag> <kernel regex> <command ...>
set -u
TAG=$1; KREG=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KR="--kernel-include-regex $KREG"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- "$@" > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- "$@" > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc3 -o pmc3 -- "$@" > $OUT/pmc3.log 2>&1
timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- "$@" > $OUT/pmc4.log 2>&1
timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc5 -o pmc5 -- "$@" > $OUT/pmc5.log 2>&1
timeout 300 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc6 -o pmc6 -- "$@" > $OUT/pmc6.log 2>&1
{ echo "== command: $*"; echo "== output of the traced run:"; grep -v "simple_timer\|amdgpu.ids\|^W2026\|^E2026" $OUT/trace.log | tail -6 | cut -c1-1500; python $ROOT/tools/rocpd_summary.py $OUT --all-grids; } > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; rm -rf $OUT/pmc? $OUT/trace
head -24 $OUT/summary.txt | cut -c1-170
