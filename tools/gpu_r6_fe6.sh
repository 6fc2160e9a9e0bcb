set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_odometry.py tests/test_gpu_host.py -x -q 2>&1 | tail -5
timeout 300 python tools/bench_odometry.py 8 256 3 2>&1 | grep -v amdgpu | tail -4
bash tools/prof_odo.sh r6_strip > /dev/null 2>&1
