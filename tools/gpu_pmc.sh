set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_pmc.py tests/test_gpu_orora.py tests/test_gpu_odometry.py tests/test_gpu_host.py 2>&1 | tail -40) > gpurun_out/pmc_tests.log 2>&1
tail -40 gpurun_out/pmc_tests.log
timeout 300 python tools/bench_pmc.py 2>&1 | tail -20 | tee gpurun_out/bench_pmc.log
