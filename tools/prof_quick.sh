#!/bin/bash
# kernel-trace stats + MFMA-busy + FETCH / WRITE passes of the headline step (4 runs).  Usage: prof_quick.sh <tag>
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --only-main"
KR="--kernel-include-regex sc_spec2_filter_kernel|sc_rescore_wave_kernel|sc_window_kernel|sc_select_kernel|sc_spec_query_kernel"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc6 -o pmc6 -- $BENCH > $OUT/bench_pmc6.log 2>&1
timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/bench_pmc3.log 2>&1
timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/bench_pmc4.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; rm -rf $OUT/pmc* $OUT/trace
cat $OUT/summary.txt
