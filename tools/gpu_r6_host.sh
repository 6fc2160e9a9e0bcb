set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sc_api.py -x -q 2>&1 | tail -3
AB_K=10 AB_PLANS="1024:25@p 1024:25 512:25 512:20 256:25 256:30 1024:15 2048:20 512:15 128:30" timeout 1200 python tools/ab_host_pieces.py 2>&1 | grep -v amdgpu
