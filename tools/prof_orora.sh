#!/bin/bash
# rocprofv3 passes of the ORORA registration leg (3500 pairs x 300-1500 matches).  Usage: prof_orora.sh <tag>
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_orora_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python $ROOT/tools/debug/orora_only.py"
KR="--kernel-include-regex orora_register"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $RUN > $OUT/trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $RUN > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $RUN > $OUT/pmc2.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT
rm -rf $OUT/*/*.db $OUT/*/*/*.db
