set -u
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --only-main 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', 'ms', round(d['ms_per_step'],4), 'filter ms', round(r['avg_launch_ms'],4), 'failures', d.get('failures'))"; }
for i in 1 2; do
  run new
  RSX_LIB_PATH=abtest/librsx_oldspec.so run old
done
timeout 900 python -m pytest tests/test_gpu_sc_filter.py -x -q 2>&1 | tail -2
