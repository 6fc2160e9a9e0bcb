set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_orora.py tests/test_gpu_pmc.py tests/test_gpu_odometry.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_pmc.py 2>&1 | grep -v amdgpu | tail -5
