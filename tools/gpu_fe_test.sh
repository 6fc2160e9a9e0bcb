set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py -x -q --durations=3 2>&1 | tail -12
