set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sc_api.py -x -q 2>&1 | grep -a "passed\|failed" | tail -2
[ -f /tmp/ab_host_data.npz ] || AB_PLANS="" python tools/ab_host_pieces.py > /dev/null 2>&1
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd() + "/tools")
import ab_host_pieces
ab_host_pieces.make_data()
PY
cd /tmp && export TMPDIR=/tmp
for cfg in "512:25 one" "1024:25 pieces"; do
  set -- $cfg
  d=/tmp/trace_host_$2
  rm -rf $d
  (cd $GRAFT_REPO_ROOT && AB_TRACE=1 AB_K=10 RSX_SC_HOST_PIECES=$1 RSX_SC_HOST_TAIL=$2 RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_exp.so timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -- python tools/_ab_host_child.py > $d.log 2>&1)
  echo "== $cfg"; python $GRAFT_REPO_ROOT/tools/host_timeline.py $d | tail -60
done
