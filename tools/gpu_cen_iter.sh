set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_cen2019.py tests/test_gpu_odometry.py -x -q 2>&1 | tail -15) > gpurun_out/gpu_tests_cen.log 2>&1
tail -6 gpurun_out/gpu_tests_cen.log
(timeout 300 python tools/bench_cen2019.py 20 64 2>&1 | grep -v amdgpu.ids) > gpurun_out/cen_bench.log; cat gpurun_out/cen_bench.log
for cfg in 0 2; do echo cfg $cfg; RSX_LIB_PATH=abtest/librsx_cen.so RSX_CEN_CFG=$cfg timeout 200 python tools/ab_cen.py 2>&1 | grep -v amdgpu.ids | tail -2; done
bash tools/prof_cen2.sh ${1:-r05_cen2019_b}
