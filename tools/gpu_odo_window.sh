set -u
cd $GRAFT_REPO_ROOT
for lib in "" abtest/librsx_odo_w32.so abtest/librsx_odo_w16.so abtest/librsx_odo_w128.so ""; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export RSX_LIB_PATH=$PWD/$lib; else unset RSX_LIB_PATH; fi
  timeout 300 python tools/bench_odometry.py 8 512 3 2>&1 | grep -v amdgpu.ids | tail -4
done 2>&1 | tee gpurun_out/odo_window_ab.log
