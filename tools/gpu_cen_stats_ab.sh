set -u
cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_cen2019.py -x -q 2>&1 | tail -2

cd $GRAFT_REPO_ROOT
for lib in "" abtest/librsx_pmcbase.so ""; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export RSX_LIB_PATH=$PWD/$lib; else unset RSX_LIB_PATH; fi
  timeout 300 python tools/bench_cen2019.py 20 64 2>&1 | grep "batch_device"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc /tmp/trc_sum && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/tools/bench_cen2019.py 5 64 > /tmp/trc.log 2>&1; mkdir -p /tmp/trc_sum/trace && (cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep -E "^grid.*cen_stats" | head -2 | cut -c1-140 )
done 2>&1 | tee gpurun_out/cen_stats_ab.log
