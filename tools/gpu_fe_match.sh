set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_odometry.py tests/test_gpu_host.py -x -q 2>&1 | tail -3) > gpurun_out/fe_match_tests.log 2>&1; cat gpurun_out/fe_match_tests.log
for lib in "" abtest/librsx_pmcbase.so ""; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export RSX_LIB_PATH=$PWD/$lib; else unset RSX_LIB_PATH; fi
  timeout 300 python tools/bench_fe_match.py 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python tools/bench_odometry.py 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
done 2>&1 | tee gpurun_out/fe_match_ab.log
unset RSX_LIB_PATH
bash tools/gpu_fe_match_trace.sh > /dev/null 2>&1; grep -E "fe_match" gpurun_out/r06_fe_match_mfma_rocprofv3.txt | cut -c1-140 | head -30
