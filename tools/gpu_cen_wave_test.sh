set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_cen2019.py tests/test_gpu_odometry.py tests/test_gpu_host.py -x -q --durations=5 2>&1 | tail -25) > gpurun_out/gpu_tests_cen_wave.log 2>&1
tail -25 gpurun_out/gpu_tests_cen_wave.log
for lib in "" abtest/librsx_pmcbase.so; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export RSX_LIB_PATH=$PWD/$lib; else unset RSX_LIB_PATH; fi
  timeout 300 python tools/bench_cen2019.py 20 64 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/bench_odometry.py 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee gpurun_out/cen_wave_ab.log
