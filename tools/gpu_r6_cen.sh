set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cen2019.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_cen2019.py 2>&1 | grep -v amdgpu | tail -6
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/tools/bench_odometry.py 8 256 2 > /tmp/trc.log 2>&1; rm -rf /tmp/trc_sum; mkdir -p /tmp/trc_sum/trace && (cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep "cen_" | head -8 | cut -c1-110; grep resident /tmp/trc.log | tail -1
