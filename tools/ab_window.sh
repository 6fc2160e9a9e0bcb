#!/bin/bash
# the window kernel (and the step) of several librsx builds on ONE box: rocprofv3 kernel trace of the main bench leg each
#   usage: tools/ab_window.sh name ...    ("prod" = the product library, else abtest/librsx_<name>.so)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for name in "$@"; do
    lib=$ROOT/abtest/librsx_$name.so; [ $name = prod ] && lib=$ROOT/navtech-radar-slam_amd/librsx.so
    rm -rf /tmp/abw_$name
    RSX_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abw_$name -o t -- python $ROOT/bench.py --steps 20 --warmup 3 --only-main --no-cpu-baseline > /tmp/abw_$name.log 2>&1
    python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/abw_$name/**/*.db", recursive=True)
con = sqlite3.connect(db[0])
import re
rows = {}
for n, c, a in con.execute("select name,total_calls,average from top_kernels"):
    m = re.search(r"(sc_\w+?_kernel)", n)
    if m: rows[m.group(1)] = (c, a)
print("$name round $round", {k: round(v[1], 1) for k, v in rows.items() if any(t in k for t in ("window", "rescore_wave", "spec2_filter", "select"))})
PY
  done
done
