#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / TCC hit of the filter kernel only.  Usage: prof_fetch.sh <tag> [env...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-main"
KR="--kernel-include-regex sc_spec_filter_kernel"
env "$@" timeout 200 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/l3.log 2>&1
env "$@" timeout 200 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc5 -o pmc5 -- $BENCH > $OUT/l5.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT 2>&1 | grep -E "SpecArgs" | cut -c1-150
rm -rf $OUT
