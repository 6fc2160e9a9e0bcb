set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/prof_r06_cenl; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="python $GRAFT_REPO_ROOT/tools/bench_cen2019.py 5 64"
  KR="--kernel-include-regex cen_runs|cen_hist"
  for v in new base; do
    if [ $v = base ]; then export RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_pmcbase.so; else unset RSX_LIB_PATH; fi
    timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc$v -o pmc$v -- $CMD > $OUT/$v.log 2>&1
  done )
python tools/rocpd_summary.py $OUT 2>&1 | grep -E "cen_runs|cen_hist" | cut -c1-160 | tee gpurun_out/cen_light_counters.txt
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
