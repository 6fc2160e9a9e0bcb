#!/bin/bash
# kernel timeline of the host-buffer entry in pieces vs whole (needs /tmp/ab_host_data.npz: python tools/ab_host_pieces.py first)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for plan in 0 512:25; do
  for k in 1 10; do
    d=$R/gpurun_out/trace_host_${plan/:/_}_k$k
    (cd $R && AB_TRACE=1 AB_K=$k RSX_SC_HOST_PIECES=$plan RSX_LIB_PATH=$R/abtest/librsx_exp.so timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -- python tools/_ab_host_child.py > $d.log 2>&1)
  done
done
