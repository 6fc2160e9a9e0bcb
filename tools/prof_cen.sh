#!/bin/bash
# per-kernel times of the batched cen2019 chain under the debug configurations of the experiments build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/trc
  RSX_LIB_PATH=$ROOT/abtest/librsx_cen.so RSX_CEN_CFG=$c timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $ROOT/tools/ab_cen.py > /tmp/trc.log 2>&1
  mkdir -p /tmp/trc_sum/trace && cp /tmp/trc/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp -r /tmp/trc/* /tmp/trc_sum/trace/
  echo "== cfg $c"; tail -1 /tmp/trc.log
  python $ROOT/tools/rocpd_summary.py /tmp/trc_sum | grep -E "cen_" | head -9 | cut -c1-110
  rm -rf /tmp/trc_sum
done
