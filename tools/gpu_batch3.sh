set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_sc_layouts.py tests/test_gpu_sc_q1.py tests/test_gpu_sc_spec.py tests/test_gpu_sc_sum_order.py tests/test_gpu_sc_window.py tests/test_gpu_voxelgrid.py tests/test_gpu_odometry.py tests/test_gpu_orora.py tests/test_gpu_loopverify.py tests/test_gpu_icp.py tests/test_gpu_host.py tests/test_gpu_frontend.py -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests2.log 2>&1
tail -8 gpurun_out/gpu_tests2.log
bash tools/ab_window.sh prod win3 win3r25 > gpurun_out/ab_window.log 2>&1; cat gpurun_out/ab_window.log
