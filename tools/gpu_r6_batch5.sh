set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest -m gpu tests/test_gpu_pmc.py tests/test_gpu_orora.py tests/test_gpu_odometry.py -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests_b5.log 2>&1
tail -12 gpurun_out/gpu_tests_b5.log
timeout 300 python tools/bench_pmc.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_pmc.log
OUT=$PWD/gpurun_out/prof_r06_pmc3; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- env NO_ORACLE=1 python $GRAFT_REPO_ROOT/tools/bench_pmc.py > $OUT/trace.log 2>&1 )
python tools/rocpd_summary.py $OUT > gpurun_out/r06_pmc_v3_trace.txt 2>&1; head -24 gpurun_out/r06_pmc_v3_trace.txt | cut -c1-160
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
