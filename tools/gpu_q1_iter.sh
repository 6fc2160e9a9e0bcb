set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_sc_q1.py -x -q 2>&1 | tail -15) > gpurun_out/q1_tests.log 2>&1
(timeout 600 python tools/bench_q1.py --sizes 32,1000,10000,100000 --k 1,10 --nq 1,8 --modes q1 --out gpurun_out/q1_bench.json 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/q1_bench.log 2>&1
(RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_exp.so RSX_RESCORE_PROF=1 timeout 300 python tools/bench_q1.py --sizes 32,10000,100000 --k 1,10 --nq 1 --modes q1 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/q1_phase.log 2>&1
tail -5 gpurun_out/q1_tests.log; cat gpurun_out/q1_bench.log; cat gpurun_out/q1_phase.log
