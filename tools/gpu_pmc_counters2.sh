set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/prof_r06_pmc5; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="env NO_ORACLE=1 PAIRS=1750 python $GRAFT_REPO_ROOT/tools/bench_pmc.py"
  KR="--kernel-include-regex pmc_build_kernel|pmc_cores_kernel|pmc_walk_kernel"
  timeout 300 rocprofv3 $KR --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1 )
python tools/rocpd_summary.py $OUT > gpurun_out/r06_pmc_v4_counters.txt 2>&1; cat gpurun_out/r06_pmc_v4_counters.txt | cut -c1-160
tail -3 $OUT/pmc2.log
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
