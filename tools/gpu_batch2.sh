set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests.log 2>&1
tail -8 gpurun_out/gpu_tests.log
bash tools/ab_window.sh prod win3 win3r25 > gpurun_out/ab_window.log 2>&1; cat gpurun_out/ab_window.log
