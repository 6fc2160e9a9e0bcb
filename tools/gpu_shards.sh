set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sc_layouts.py tests/test_gpu_sc_window.py tests/test_gpu_sc_filter.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --only-main 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value', d['value'], 'ms', d['ms_per_step'], 'filter ms', r['avg_launch_ms'], 'failures', d.get('failures'))"
