#!/bin/bash
# Issue / wait / LDS / VMEM counters of the kernels matching a regex, headline step.  Usage: prof_kernel.sh <tag> <regex> [env...]
TAG=$1; KREG=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-main ${BENCH_EXTRA:-}"
KR="--kernel-include-regex $KREG"
env "$@" timeout 240 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/bench_pmc1.log 2>&1
env "$@" timeout 240 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/bench_pmc2.log 2>&1
env "$@" timeout 240 rocprofv3 $KR --pmc SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/bench_pmc3.log 2>&1
env "$@" timeout 240 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/bench_pmc4.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; rm -rf $OUT/pmc*
cat $OUT/summary.txt
