set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sc_layouts.py tests/test_gpu_sc_sharded.py tests/test_gpu_sc_window.py -x -q 2>&1 | tail -4
timeout 300 python tools/bench_layouts.py 2>&1 | tail -1
