#!/bin/bash
# Kernel trace of the windowed odometry pipeline.  Usage: prof_odo.sh <tag>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/tools/bench_odometry.py 8 256 2 > $OUT/bench_trace.log 2>&1
python - <<PY > $OUT/summary.txt
import sqlite3, glob
db = glob.glob("$OUT/trace/*.db")[0]
con = sqlite3.connect(db)
print("== rocprofv3 --kernel-trace: tools/bench_odometry.py 8 256 2 (windows of 64 scans) ==")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 24"):
    print(f"{calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:100]}")
PY
tail -4 $OUT/bench_trace.log >> $OUT/summary.txt
rm -rf $OUT/trace
cat $OUT/summary.txt
