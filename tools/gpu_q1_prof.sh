set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_exp.so RSX_RESCORE_PROF=1
(timeout 300 python tools/bench_q1.py --sizes 32,1000,10000,100000 --k 1,10 --nq 1 --modes q1 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/q1_phase.log 2>&1
cat gpurun_out/q1_phase.log
