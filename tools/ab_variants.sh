#!/bin/bash
# filter time of several librsx builds on ONE box (boxes differ by +-5 %): abtest/librsx_<name>.so ..., two rounds interleaved
#   usage: tools/ab_variants.sh name[:kind] ...     (kind = spectral2 by default; "prod" = the product library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; kind=spectral2; [[ $spec == *:* ]] && kind=${spec##*:}
    lib=$ROOT/abtest/librsx_$name.so; [ $name = prod ] && lib=$ROOT/navtech-radar-slam_amd/librsx.so
    RSX_LIB_PATH=$lib timeout 200 python bench.py --steps 20 --warmup 3 --only-main --no-cpu-baseline --filter-kind $kind 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$spec', 'round $round', 'filter_ms', round(r['avg_launch_ms'],3), 'step_ms', round(d['ms_per_step'],3), 'failures', d.get('failures'))"
  done
done
