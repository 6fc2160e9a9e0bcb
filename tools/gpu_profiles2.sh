set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_six.sh r05_cen2019 "cen_" python $GRAFT_REPO_ROOT/tools/bench_cen2019.py 20 64 > /dev/null 2>&1
bash tools/prof_six.sh r05_loopverify "(icp_|vg_|lv_)" python $GRAFT_REPO_ROOT/tools/bench_loopverify.py lv > /dev/null 2>&1
bash tools/prof_six.sh r05_odometry "(cen_|fe_|odo_|orora)" python $GRAFT_REPO_ROOT/tools/bench_odometry.py 8 256 2 > /dev/null 2>&1
for t in r05_cen2019 r05_loopverify r05_odometry; do echo "=== $t"; sed -n 3,14p gpurun_out/prof_$t/summary.txt | cut -c1-140; done
