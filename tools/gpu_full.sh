set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/gpu_tests.log 2>&1
tail -8 gpurun_out/gpu_tests.log
(timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err); tail -c 600 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "failures") if k in d})
print(json.dumps(d.get("latency_q1"), indent=0)[:3000])
print("roofline", d.get("roofline"))
PY
bash tools/prof_q1.sh r05_q1 10000,100000 > gpurun_out/prof_q1.log 2>&1; tail -60 gpurun_out/prof_q1.log
