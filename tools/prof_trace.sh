#!/bin/bash
# Kernel trace of the headline step (trajectory DB).  Usage: prof_trace.sh <tag> [env...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 7 --warmup 2 --no-cpu-baseline --only-main > $OUT/bench_trace.log 2>&1
python - <<PY > $OUT/summary.txt
import sqlite3, glob
db = glob.glob("$OUT/trace/*.db")[0]
con = sqlite3.connect(db)
print("== rocprofv3 --kernel-trace --stats: bench.py --steps 7 --warmup 2 --no-cpu-baseline --only-main ==")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 16"):
    print(f"{calls:6d} {total/1e3:12.1f} {avg/1e3:10.2f} {pct:6.2f}  {name[:110]}")
PY
python - <<PY >> $OUT/summary.txt
import json
try:
    d = json.loads([l for l in open("$OUT/bench_trace.log") if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "exact_evals_per_query", "previewed_candidates_per_query", "window_previews_per_query", "valu_previews_per_query", "failures")})
except Exception as e:
    print("bench line:", e)
PY
rm -rf $OUT/trace
cat $OUT/summary.txt
