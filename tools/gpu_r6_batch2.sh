set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest -m gpu -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests.log 2>&1
tail -6 gpurun_out/gpu_tests.log
timeout 300 python tools/bench_pmc.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_pmc.log
echo "--- q1: queries sharing a launch"
timeout 600 python tools/bench_q1.py --sizes 10000,100000 --k 1 --nq 1,2,4,8 --modes q1,filter --reps 100 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee gpurun_out/bench_q1.log
echo "--- filter: DMA pieces split over both waves of a pair (A/B, same box)"
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --only-main 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', 'ms', round(d['ms_per_step'],4), 'filter ms', round(r['avg_launch_ms'],4), 'failures', d.get('failures'))"; }
for i in 1 2; do
  run base
  RSX_LIB_PATH=$PWD/abtest/librsx_dmasplit.so run dmasplit
done
RSX_LIB_PATH=$PWD/abtest/librsx_dmasplit.so timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_sc_spec.py tests/test_gpu_sc_filter.py 2>&1 | tail -3
