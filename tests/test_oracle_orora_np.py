"""Cross-check of the ORORA oracle (oracle/orora_ref.c) with an independent numpy restatement written from SURVEY.md App.
B.3 / B.4 by a different computational route (explicit SVD; consensus sets evaluated from scratch): oracle/orora_np.py.
PARITY UNPINNED w.r.t. the reference (the solver's sources are absent) -- this is the only cross-check available; the
modelling choices both restatements share are listed in orora_np.CHOICES."""
import numpy as np

from navtech_radar_slam_amd import synth
from oracle import orora_np


def test_bench_pairs_agree(oracle):
    """The first pairs of bench.py's ORORA workload (seed 777: 300-1500 matches, 20-60 % outliers)."""
    src, dst, off, truth = synth.orora_pairs(777, 12)
    want = oracle.orora_register_batch(src, dst, off)
    for i in range(12):
        s, d = src[off[i]:off[i + 1]], dst[off[i]:off[i + 1]]
        got = orora_np.register(s, d)
        w = want[i]
        assert abs(got["x"] - w["x"]) < 1e-9 and abs(got["y"] - w["y"]) < 1e-9 and abs(got["yaw"] - w["yaw"]) < 1e-12, (i, got, w)
        assert (got["iterations"], got["rot_inliers"], got["trans_inliers"]) == (w["iterations"], w["rot_inliers"], w["trans_inliers"])
        assert np.hypot(got["x"] - truth[i, 0], got["y"] - truth[i, 1]) < 0.1 and abs(got["yaw"] - truth[i, 2]) < 3e-3


def test_switchable_choices_agree(oracle):
    """The two unpinned modelling choices that are parameters (complete TIM graph, TEASER++'s cost form)."""
    src, dst, off, _ = synth.orora_pairs(5, 4, k_range=(40, 120))
    for flags, kw in ((1, {"complete": True}), (2, {"teaser_cost": True}), (3, {"complete": True, "teaser_cost": True})):
        p = oracle.orora_default_params()
        p.flags = flags
        want = oracle.orora_register_batch(src, dst, off, params=p)
        for i in range(4):
            got = orora_np.register(src[off[i]:off[i + 1]], dst[off[i]:off[i + 1]], **kw)
            assert abs(got["x"] - want[i]["x"]) < 1e-9 and abs(got["y"] - want[i]["y"]) < 1e-9 and abs(got["yaw"] - want[i]["yaw"]) < 1e-12
            assert got["iterations"] == want[i]["iterations"] and got["rot_inliers"] == want[i]["rot_inliers"]


def test_scalar_tls_with_ties_and_degenerate_sets(oracle):
    rng = np.random.default_rng(3)
    for trial in range(30):
        n = int(rng.integers(1, 40))
        x = np.round(rng.normal(0, 1, n), 1)            # coarse values: equal endpoints are common
        beta = rng.choice([0.1, 0.2, 0.5], n)
        est, _ = oracle.orora_scalar_tls(x, beta)
        assert abs(orora_np.scalar_tls(x, beta) - est) < 1e-12, (trial, x, beta)
