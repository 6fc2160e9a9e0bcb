set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest -m gpu -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests.log 2>&1
tail -8 gpurun_out/gpu_tests.log
(timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err); tail -c 400 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "failures") if k in d})
print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms")})
print("secondary", d["config"].get("secondary"))
PY
