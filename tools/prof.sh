#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel-trace stats + PMC passes, each in its
# own run (never combined with sys/runtime trace).  Outputs under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --only-main"
# PMC passes: only the query-path kernels (the 18k build-path dispatches of the data set-up would each be serialised)
KR="--kernel-include-regex sc_spec2_filter_kernel|sc_spec_filter_kernel|sc_filter_kernel|sc_rescore_kernel|sc_rescore_wave_kernel|sc_window_kernel|sc_select_kernel|sc_spec_query_kernel|sc_keys_kernel|sc_merge_kernel"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/bench_pmc1.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/bench_pmc2.log 2>&1
timeout 300 rocprofv3 $KR --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS --kernel-trace -d $OUT/pmc6 -o pmc6 -- $BENCH > $OUT/bench_pmc6.log 2>&1
timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/bench_pmc3.log 2>&1
timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/bench_pmc4.log 2>&1
timeout 300 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc5 -o pmc5 -- $BENCH > $OUT/bench_pmc5.log 2>&1
find $OUT -name "*.csv" | head -50
du -sh $OUT
