set -u
cd $GRAFT_REPO_ROOT
bash tools/prof_cen2.sh r06_cenwave > /dev/null 2>&1
( echo "== command: python tools/bench_cen2019.py 20 64 (wavefront-per-azimuth forms of cen_hist / cen_runs / cen_adjacent)"; echo "== output of the traced run:"; grep "cen2019" gpurun_out/prof_r06_cenwave/trace.log; cat gpurun_out/prof_r06_cenwave/summary.txt ) > gpurun_out/r06_cen2019_wave_rocprofv3.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $GRAFT_REPO_ROOT/tools/bench_odometry.py 8 256 2 > /tmp/trc.log 2>&1; rm -rf /tmp/trc_sum; mkdir -p /tmp/trc_sum/trace && (cp /tmp/trc/*/*.db /tmp/trc_sum/trace/ 2>/dev/null || cp /tmp/trc/*.db /tmp/trc_sum/trace/)
( echo "== rocprofv3 --kernel-trace: tools/bench_odometry.py 8 256 2 (windows of 64 scans; wavefront-per-azimuth cen2019 kernels) =="; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/trc_sum | sed -n 2,30p | cut -c1-170; grep resident /tmp/trc.log | tail -2 ) > $GRAFT_REPO_ROOT/gpurun_out/r06_odometry_wave_rocprofv3.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r06*/*/*.db $GRAFT_REPO_ROOT/gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
head -30 $GRAFT_REPO_ROOT/gpurun_out/r06_odometry_wave_rocprofv3.txt | cut -c1-150
