set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest -m gpu -x -q 2>&1 | grep -a "passed\|failed\|error" | tail -5) > gpurun_out/gpu_tests.log 2>&1
cat gpurun_out/gpu_tests.log
(timeout 1200 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err); echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "failures") if k in d})
print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms")})
print("secondary", d["config"].get("secondary"))
print("slam", {k: d["slam_stream"].get(k) for k in ("keyframes_per_sec", "cpp_host")})
print("odo", {k: d["odometry_e2e"].get(k) for k in ("scans_per_sec_resident", "scans_per_sec_host_images")})
print("host", d["host_entry"]["vs_resident"], d["host_entry"]["pinned_vs_resident"])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
