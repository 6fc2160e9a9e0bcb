set -u
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_sc_filter.py tests/test_gpu_sc_spec.py tests/test_gpu_sc.py tests/test_gpu_sc_api.py tests/test_gpu_sc_window.py tests/test_gpu_sc_layouts.py -x -q 2>&1 | grep -a "passed\|failed\|rror" | tail -3
run() { cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 10 --no-cpu-baseline > /tmp/trc.log 2>&1; rm -rf /tmp/trc_sum; mkdir -p /tmp/trc_sum/trace && (cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep "sc_select\|sc_rescore_wave\|sc_window_kernel" | head -3 | cut -c1-120; tail -1 /tmp/trc.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'failures', d.get('failures'))"; cd $GRAFT_REPO_ROOT; }
echo wave; run
echo block; export RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_exp.so RSX_SC_SELECT=block; run
