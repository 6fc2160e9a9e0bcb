set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_odometry.py -x -q 2>&1 | tail -3
run() { cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/tools/bench_odometry.py 8 256 2 > /tmp/trc.log 2>&1; rm -rf /tmp/trc_sum; mkdir -p /tmp/trc_sum/trace && (cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep "fe_describe\|fe_match_cons\|fe_cart" | head -3 | cut -c1-110; grep resident /tmp/trc.log | tail -1; cd $GRAFT_REPO_ROOT; }
echo base; run
for v in ${VARIANTS:-}; do echo $v; export RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_fe_$v.so; run; unset RSX_LIB_PATH; done
