set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sc_api.py tests/test_gpu_sc_spec.py tests/test_gpu_sc_filter.py -x -q 2>&1 | grep -a "passed\|failed" | tail -2
AB_K=10 AB_PLANS="${PLANS}" timeout 1200 python tools/ab_host_pieces.py 2>&1 | grep "^[0-9]"
