set -u
cd $GRAFT_REPO_ROOT
run() { cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 10 --no-cpu-baseline > /tmp/trc.log 2>&1; rm -rf /tmp/trc_sum; mkdir -p /tmp/trc_sum/trace && (cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep "sc_window_kernel\|sc_rescore_wave\|sc_select" | head -3 | cut -c1-120; cd $GRAFT_REPO_ROOT; }
echo base; run
for v in ${VARIANTS}; do echo $v; export RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_$v.so; run; unset RSX_LIB_PATH; done
