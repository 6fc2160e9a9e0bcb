set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 300 python -m pytest tests/test_gpu_voxelgrid.py tests/test_gpu_icp.py tests/test_gpu_loopverify.py -x -q 2>&1 | tail -5
timeout -s KILL 300 python tools/bench_loopverify.py 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); lv=d['loop_verify']; print('lv ms', lv['ms_per_verification'], 'iters', lv['mean_icp_iterations'], 'oracle', lv['first_verdict_equals_oracle'], 'map ms', lv['map']['ms']); print('icp ms', d['icp']['ms_per_align'], d['icp']['iterations'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/tools/bench_loopverify.py lv > /tmp/trc.log 2>&1; mkdir -p /tmp/trc_sum/trace && cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | head -6 | cut -c1-150
