set -u
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --only-main 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', 'ms', round(d['ms_per_step'],4), 'filter ms', round(r['avg_launch_ms'],4), 'failures', d.get('failures'))"; }
for rep in 1 2; do
  run base
  for v in ${VARIANTS}; do RSX_LIB_PATH=$PWD/abtest/librsx_$v.so run $v; done
done
