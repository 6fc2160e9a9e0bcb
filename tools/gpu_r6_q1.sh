set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sc_q1.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_q1.py --sizes 10000,50000 --k 1 --nq 8,12,16 --modes q1,filter 2>&1 | grep "^n[0-9]" | python -c "
import sys, json
for line in sys.stdin:
    name, js = line.split(' ', 1)
    d = json.loads(js)
    print(name, {m: (d[m]['us_per_call_stream'], d[m]['kernel']) for m in ('q1', 'filter')}, d.get('identical'))
"
