set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stream_cpp.log
import json, sys
sys.path.insert(0, '.')
import bench
from navtech_radar_slam_amd import synth
db_pts, db_off, *_ = synth.trajectory_keyframes(1234, 3000, 4321, 4, binary_z=True)
for rep in range(2):
    r = bench.slam_stream_leg(0, db_pts, db_off)
    print(json.dumps({k: r[k] for k in ("keyframes_per_sec", "cpp_host", "candidate_mode_identical_to_oracle", "exhaustive_mode_identical_to_oracle", "queries")}))
PY
rocprofv3 -L 2>/dev/null | grep -io "TCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*\|TCC_EA0_WRREQ[A-Z0-9_]*\|TCC_WRITE[A-Z0-9_]*" | sort -u | head -30 > gpurun_out/tcc_counters.txt; cat gpurun_out/tcc_counters.txt
OUT=$PWD/gpurun_out/prof_r06_cenw; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="python $GRAFT_REPO_ROOT/tools/bench_cen2019.py 10 64"
  KR="--kernel-include-regex cen_hist|cen_runs|cen_stats"
  timeout 300 rocprofv3 $KR --pmc TCC_ATOMIC_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $OUT/pmcA -o pmcA -- $CMD > $OUT/pmcA.log 2>&1
  timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmcB -o pmcB -- $CMD > $OUT/pmcB.log 2>&1
  timeout 300 rocprofv3 $KR --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WR_UNCACHED_32B_sum --kernel-trace -d $OUT/pmcC -o pmcC -- $CMD > $OUT/pmcC.log 2>&1 )
python tools/rocpd_summary.py $OUT --all-grids > gpurun_out/r06_cen_write_counters.txt 2>&1; grep "cen_hist\|cen_runs" gpurun_out/r06_cen_write_counters.txt | cut -c1-150; tail -2 $OUT/pmcA.log | cut -c1-200
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
