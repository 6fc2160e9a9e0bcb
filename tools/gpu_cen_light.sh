set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_cen2019.py tests/test_gpu_odometry.py -x -q 2>&1 | tail -15) > gpurun_out/gpu_tests_cen.log 2>&1
tail -6 gpurun_out/gpu_tests_cen.log
for lib in "" abtest/librsx_pmcbase.so "" abtest/librsx_pmcbase.so; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export RSX_LIB_PATH=$PWD/$lib; else unset RSX_LIB_PATH; fi
  timeout 300 python tools/bench_cen2019.py 20 64 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/bench_odometry.py 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee gpurun_out/cen_light_ab.log
unset RSX_LIB_PATH
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/tools/bench_cen2019.py 20 64 > /tmp/trc.log 2>&1; mkdir -p /tmp/trc_sum/trace && cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep -E "^grid" | head -12 | cut -c1-150 | tee $GRAFT_REPO_ROOT/gpurun_out/cen_light_trace.log
