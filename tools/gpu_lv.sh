set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in tests/test_gpu_voxelgrid.py tests/test_gpu_icp.py tests/test_gpu_loopverify.py; do
  echo "== $t"
  timeout -s KILL 300 python -m pytest $t -x -q 2>&1 | tail -15
done
