"""bench.py must start its own ranks: `python bench.py --gpus N` with no torch.distributed environment
re-executes itself under torch.distributed.run with N processes.  There is no GPU
here, so the ranks run a CPU stand-in shard over gloo (tests/bench_stub.py); what is under test is the
launcher, the rank bookkeeping, the barrier-bracketed timing, the max-over-ranks reduction and the
one-JSON-line contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus, extra_env=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["RSX_BENCH_LOCAL_BACKEND"] = "tests.bench_stub:make"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--db", "150", "--queries", "4", "--topk", "3", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_spawns_its_own_ranks(gpus):
    p, lines = _run(gpus)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["rccl_ranks"] == gpus
    assert len(out["per_rank_ms_per_step"]) == gpus
    assert out["steps"] == 2 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["planted_loops_recovered"] is True and out["failures"] == []
    assert out["ms_per_step"] == pytest.approx(max(out["per_rank_ms_per_step"]))   # MAX over ranks
    assert out["value"] == pytest.approx(4 / (out["ms_per_step"] * 1e-3), rel=1e-6)
    if gpus > 1:
        # the headline at N > 1 is the layout north_star names -- the DB sharded over the GPUs, all-gather of the top-k (1xN) --
        # whatever the batch size; every other layout of the world is timed beside it and must return the same records
        assert out["backend"] == "gloo" and out["config"]["parallelism"] == "query_groups1_x_db_shards2" and out["config"]["layout"] == "1x2"
        assert set(out["layouts"]) == {"1x2", "2x1"} and out["layouts"]["1x2"]["headline"] is True
        assert out["layouts"]["2x1"]["identical_to_headline"] is True and len(out["layouts"]["2x1"]["per_rank_ms_per_step"]) == 2
        assert out["best_layout"] in out["layouts"] and out["best_layout_queries_per_sec"] > 0
        assert all(len(v["exchange_ms_per_step_per_rank"]) == 2 for v in out["layouts"].values())


def test_headline_is_the_db_shard_layout_for_a_large_batch_too():
    """1024 queries at world 2 would feed two query groups (the automatic layout: DB replicated, pure query parallelism);
    the headline stays 1x2, the automatic layout is timed beside it and marked"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["RSX_BENCH_LOCAL_BACKEND"] = "tests.bench_stub:make"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--db", "80", "--queries", "1024",
           "--topk", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["layout"] == "1x2" and out["layouts"]["1x2"]["headline"] is True and out["failures"] == []
    assert out["layouts"]["2x1"]["auto_layout"] is True and out["layouts"]["2x1"]["identical_to_headline"] is True


def test_bench_query_groups_flag():
    """--query-groups 2 at world 2: pure query parallelism (DB replicated), same planted loops recovered."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["RSX_BENCH_LOCAL_BACKEND"] = "tests.bench_stub:make"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--db", "150", "--queries", "4",
           "--topk", "3", "--no-cpu-baseline", "--query-groups", "2", "--only-main"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["layout"] == "2x1" and out["planted_loops_recovered"] is True and out["failures"] == [] and "layouts" not in out


def test_bench_refuses_a_world_that_does_not_match():
    p, _ = _run(2, {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_bench_refuses_more_gpus_than_visible():
    # without the test backend the launcher counts GPUs first: none here
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RSX_BENCH_LOCAL_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout)


def test_a_secondary_layout_that_raises_does_not_cost_the_headline():
    """N > 1: every secondary leg runs in its own try (bench.LegGuard); the failure is recorded, the headline printed, rc 0."""
    p, lines = _run(2, {"RSX_BENCH_TEST_FAULT": "raise:layout 2x1"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["failures"] == [] and out["layouts"]["1x2"]["headline"] is True
    assert "injected fault" in out["layouts"]["2x1"]["error"]
    assert any("layout 2x1" in m for m in out["secondary_failures"])


def test_a_secondary_layout_that_wedges_is_cut_at_its_deadline():
    """... and under a deadline: a leg that never returns (a wedged collective cannot be interrupted) ends the run through
    os._exit on every rank, after rank 0 has written the one JSON line with the headline and the failure."""
    import time
    t0 = time.time()
    p, lines = _run(2, {"RSX_BENCH_TEST_FAULT": "sleep:layout 2x1", "RSX_BENCH_LEG_TIMEOUT": "5"})
    assert time.time() - t0 < 300
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, (p.stdout, p.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["cut_short"] is True and out["n_gpus"] == 2
    assert any("layout 2x1" in m and "no result after 5 s" in m for m in out["secondary_failures"])
