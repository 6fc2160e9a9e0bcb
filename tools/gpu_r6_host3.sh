set -u
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd() + "/tools")
import ab_host_pieces
ab_host_pieces.make_data()
PY
cd /tmp && export TMPDIR=/tmp
for R in 0 1 2 3 4; do
  d=/tmp/trace_host_R$R
  rm -rf $d
  (cd $GRAFT_REPO_ROOT && AB_TRACE=1 AB_K=10 RSX_SC_HOST_PIECES=512:25 RSX_SPEC_XCD_S=$R RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_exp.so timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -- python tools/_ab_host_child.py > $d.log 2>&1)
  echo "== R=$R"; python $GRAFT_REPO_ROOT/tools/host_timeline.py $d | grep "filter_kernel\|span" | tail -5
done
