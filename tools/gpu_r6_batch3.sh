set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err); echo "bench rc=$?"; tail -c 300 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "failures") if k in d})
print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms")})
print("secondary", d["config"].get("secondary"))
print("q1", {n: (round(v["us_per_query_stream"], 1), round(v["hbm_frac_read"], 3)) for n, v in d["latency_q1"].items() if isinstance(v, dict) and "us_per_query_stream" in v})
print("orora sel", d["orora"]["with_max_clique_selection"])
print("odo", {k: d["odometry_e2e"].get(k) for k in ("scans_per_sec_resident", "worst_pair_error_vs_truth", "max_abs_pose_diff_vs_oracle", "counts_identical_to_oracle")})
print("lv", d["loop_verify"].get("first_verdict_vs_sequential_float_oracle"))
PY
# --- profiles: the filter (kernel trace + six PMC passes), the single-query path at 100 k / 200 k / 400 k, the selection kernel
bash tools/prof.sh r06 > gpurun_out/prof_r06.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r06 > gpurun_out/r06_sc_spec_v16_rocprofv3.txt 2>&1; head -30 gpurun_out/r06_sc_spec_v16_rocprofv3.txt
OUT=$PWD/gpurun_out/prof_r06_q1; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="python $GRAFT_REPO_ROOT/tools/bench_q1.py --modes q1 --sizes 100000,200000,400000 --k 1 --nq 1 --reps 60"
  KR="--kernel-include-regex sc_q1_kernel"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  timeout 400 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
  timeout 400 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1 )
python tools/rocpd_summary.py $OUT --all-grids > gpurun_out/r06_q1_rocprofv3.txt 2>&1; grep -a "n[0-9]*_k1_nq1" $OUT/trace.log | cut -c1-200; head -40 gpurun_out/r06_q1_rocprofv3.txt
OUT=$PWD/gpurun_out/prof_r06_pmc; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="env NO_ORACLE=1 python $GRAFT_REPO_ROOT/tools/bench_pmc.py"
  KR="--kernel-include-regex pmc_select_kernel|orora_register_kernel"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
  timeout 300 rocprofv3 $KR --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
  timeout 300 rocprofv3 $KR --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1 )
python tools/rocpd_summary.py $OUT > gpurun_out/r06_pmc_rocprofv3.txt 2>&1; head -50 gpurun_out/r06_pmc_rocprofv3.txt
# --- the lever that did not pay: DMA pieces split over both waves of a pair -- its counters beside the baseline's
OUT=$PWD/gpurun_out/prof_r06_dmasplit; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  export RSX_LIB_PATH=$GRAFT_REPO_ROOT/abtest/librsx_dmasplit.so
  BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --only-main"
  KR="--kernel-include-regex sc_spec2_filter_kernel"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS --kernel-trace -d $OUT/pmc6 -o pmc6 -- $BENCH > $OUT/pmc6.log 2>&1 )
python tools/rocpd_summary.py $OUT > gpurun_out/r06_sc_spec_dmasplit_rocprofv3.txt 2>&1; head -30 gpurun_out/r06_sc_spec_dmasplit_rocprofv3.txt
du -sh gpurun_out; rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null; du -sh gpurun_out
