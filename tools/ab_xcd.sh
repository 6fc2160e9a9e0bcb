#!/bin/bash
# A/B of the XCD-aware work split of sc_spec2_filter_kernel (experiments build abtest/librsx_exp.so): kernel time of the
# headline launch with the old contiguous split (RSX_SPEC_XCD=0), the default and forced sub-range counts; then the L2 ->
# fabric traffic (FETCH_SIZE / WRITE_SIZE passes) of both
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_xcd
mkdir -p $OUT
cd $ROOT
export RSX_LIB_PATH=$ROOT/abtest/librsx_exp.so
for cfg in "RSX_SPEC_XCD=0" "RSX_SPEC_XCD=1" "RSX_SPEC_XCD=0" "RSX_SPEC_XCD=1" "RSX_SPEC_XCD_S=3"; do
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --only-main --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<PY
import json, sys
try:
    d = json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 3), "filter_ms", round(r["avg_launch_ms"], 4), "failures", d.get("failures"))
except Exception as e:
    print(sys.argv[1], "parse failed", e, open("$OUT/b.err").read()[-800:])
PY
done
cd /tmp && export TMPDIR=/tmp
for cfg in "RSX_SPEC_XCD=1"; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
    d=$OUT/pmc_${cfg#*=}_$(echo $pmc | tr ' ' '_')
    env $cfg timeout 240 rocprofv3 --kernel-include-regex "sc_spec2_filter" --pmc $pmc --kernel-trace -d $d -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-main > $d.log 2>&1
  done
done
python $ROOT/tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/summary.txt | tail -40
