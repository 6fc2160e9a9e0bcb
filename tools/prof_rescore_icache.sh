#!/bin/bash
# instruction-fetch counters of the re-scoring kernel, window previews on / off (experiments build).  Usage: prof_rescore_icache.sh <tag>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RSX_LIB_PATH=$ROOT/abtest/librsx_exp.so
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -40 > $OUT/avail.txt
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-main"
KR="--kernel-include-regex sc_rescore_kernel"
for w in 1 0; do
  mkdir -p $OUT
  RSX_SC_WINDOW=$w timeout 240 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmcA$w -o a -- $BENCH > $OUT/log_a$w.txt 2>&1
  RSX_SC_WINDOW=$w timeout 240 rocprofv3 $KR --pmc SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace -d $OUT/pmcB$w -o b -- $BENCH > $OUT/log_b$w.txt 2>&1
  RSX_SC_WINDOW=$w timeout 240 rocprofv3 $KR --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $OUT/pmcC$w -o c -- $BENCH > $OUT/log_c$w.txt 2>&1
done
python $ROOT/tools/rocpd_summary.py $OUT 2>&1 | head -80 > $OUT/summary.txt
cat $OUT/avail.txt | head -30
cat $OUT/summary.txt
tail -3 $OUT/log_c1.txt
find $OUT -name "*.db" | head; 
