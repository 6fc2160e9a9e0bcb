set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest -m gpu -x -q 2>&1 | tail -15) > gpurun_out/gpu_tests.log 2>&1
tail -5 gpurun_out/gpu_tests.log
(timeout 1200 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err); echo "bench rc=$?"; tail -c 300 gpurun_out/bench_full.err | grep -v amdgpu
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "failures") if k in d})
print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms")})
print("secondary", d["config"].get("secondary"))
print("orora sel", {k: v for k, v in d["orora"]["with_max_clique_selection"].items() if k != "note"})
print("odo", {k: d["odometry_e2e"].get(k) for k in ("scans_per_sec_resident", "scans_per_sec_host_images", "worst_pair_error_vs_truth", "max_abs_pose_diff_vs_oracle", "counts_identical_to_oracle")}, d["odometry_e2e"].get("file_entry"))
print("host", d["host_entry"]["vs_resident"], "slam", d.get("slam_stream", {}).get("keyframes_per_sec"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
OUT=$PWD/gpurun_out/prof_r06_pmc_final; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="env NO_ORACLE=1 python $GRAFT_REPO_ROOT/tools/bench_pmc.py"
  KR="--kernel-include-regex pmc_build_kernel|pmc_cores_kernel|pmc_walk_kernel|orora_register_kernel"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
  timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
  timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
  timeout 300 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1 )
python tools/rocpd_summary.py $OUT > gpurun_out/r06_pmc_rocprofv3.txt 2>&1; head -12 gpurun_out/r06_pmc_rocprofv3.txt | cut -c1-150
bash tools/prof_odo.sh r06odo > gpurun_out/prof_odo.log 2>&1; ls gpurun_out/prof_r06odo 2>/dev/null | head; tail -5 gpurun_out/prof_odo.log | cut -c1-200
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
