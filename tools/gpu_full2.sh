set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests.log 2>&1
tail -8 gpurun_out/gpu_tests.log
(timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err); tail -c 400 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "failures") if k in d})
print("roofline", d.get("roofline"))
print("cen", {k: d["cen2019"][k] for k in ("scans_per_sec", "batched_host_scans_per_sec", "batched_device_scans_per_sec")})
print("lv", {k: d["loop_verify"][k] for k in ("ms_per_verification", "first_verdict_equals_oracle", "mean_icp_iterations")}, d["loop_verify"]["map"])
print("icp", d["icp"]["ms_per_align"], "host_entry", d["host_entry"]["vs_resident"], d["host_entry"]["queries_per_sec"])
le = d["layout_emulation"]; print("layout 10k", le["per_rank_ms_per_step"]["8"], le["compute_speedup_db_shards_only"], le["compute_speedup_of_best_layout"])
print("q1", {n: (d["latency_q1"][n]["us_per_query_stream"], d["latency_q1"][n]["us_per_query_host_call"], d["latency_q1"][n]["default_path_kernel"]) for n in ("n1000", "n10000", "n100000")}, d["latency_q1"]["floor_us"])
print("odo", d["odometry_e2e"].get("pipeline_scans_per_sec"), "slam", d["slam_stream"].get("keyframes_per_sec"), "orora", d["orora"].get("pairs_per_sec"))
PY
