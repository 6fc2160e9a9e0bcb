#!/bin/bash
# A/B of the two mappings of the spectral filter on the GPU box (run through gpurun): parity tests of both, the region
# profile of the two-wave kernel (instrumented build, if abtest/librsx_instr.so is there), then the headline workload with
# each.  Outputs under gpurun_out/ab_spec2/.   usage: tools/ab_spec2.sh [quick]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_spec2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_sc_spec.py -x -q -m gpu > $OUT/pytest_spec.log 2>&1
echo "pytest spec rc=$?"; tail -5 $OUT/pytest_spec.log
if [ -f abtest/librsx_instr.so ]; then
  for dbg in ${DBGS:-0}; do
    echo "instrumented build, RSX_SPEC_DBG=$dbg"
    RSX_LIB_PATH=$PWD/abtest/librsx_instr.so RSX_SPEC_PROF=1 RSX_SPEC_DBG=$dbg timeout 300 python bench.py --steps 4 --warmup 1 --only-main --no-cpu-baseline --filter-kind spectral2 2>&1 | grep -a "spec2 prof\|ms_per_step" | cut -c1-330 | tail -3
  done
fi
KINDS="spectral spectral2 spectral spectral2"
[ "${1:-}" = quick ] && KINDS="spectral2 spectral2"
for kind in $KINDS; do
  timeout 300 python bench.py --steps 20 --warmup 3 --only-main --no-cpu-baseline --filter-kind $kind > $OUT/bench_$kind.json 2> $OUT/bench_$kind.err
  echo "bench $kind rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$kind.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$kind", "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "filter_ms", round(r["avg_launch_ms"],3), "frac", round(r["frac"],3), r["kernel"], "failures", d.get("failures"))
except Exception as e:
    print("parse failed", e); print(open("$OUT/bench_$kind.err").read()[-2000:])
PY
done
if [ "${1:-}" != quick ]; then
  timeout 900 python -m pytest tests/test_gpu_sc_filter.py -x -q -m gpu > $OUT/pytest_filter.log 2>&1
  echo "pytest filter rc=$?"; tail -5 $OUT/pytest_filter.log
fi
