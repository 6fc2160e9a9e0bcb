"""GPU parity of the ICP loop verification (icp.hip through the C-ABI) against the oracle.  Floating
point: the nearest-neighbour correspondences are identical by construction (same float expression,
same tie rule); the moment sums run in fp64 in a fixed tree order on the GPU, which the oracle restates
(ICPREF_SUM_TREE: 1e-6 and the same iteration count) beside its sequential float sums (1e-4, the
north-star pose tolerance)."""
import numpy as np
import pytest

from test_oracle_icp import rot, scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def icpmod():
    from navtech_radar_slam_amd import _rsx, icp
    assert _rsx.device_count() >= 1
    return icp


@pytest.mark.parametrize("seed,ns,nt", [(1, 500, 1500), (2, 2000, 6000), (3, 1200, 40000)])
def test_align_matches_oracle(icpmod, oracle, seed, ns, nt):
    tgt = scene(seed, nt)
    R, t = rot(0.05 * seed, 0.01, -0.01), np.array([0.5, -0.3 * seed, 0.04])
    rng = np.random.default_rng(seed)
    sub = tgt[rng.choice(len(tgt), ns, replace=False)] + rng.normal(0, 0.02, (ns, 3))
    src = ((sub - t) @ R).astype(np.float32)
    ic = icpmod.Icp()
    got = ic.align(src, tgt)
    want = oracle.icp_align(src, tgt)
    assert got["converged"] == want["converged"]
    assert np.abs(got["transform"] - want["transform"]).max() < TOL
    assert abs(got["fitness"] - want["fitness"]) < TOL * max(1.0, want["fitness"])
    assert abs(got["iterations"] - want["iterations"]) <= 2          # thresholds at 1e-6 see the sum order
    same = oracle.icp_align(src, tgt, sum_order=oracle.ICP_SUM_TREE)   # the device's order of additions
    assert got["iterations"] == same["iterations"] and got["state"] == same["state"]
    assert np.abs(got["transform"] - same["transform"]).max() < 1e-6 and abs(got["fitness"] - same["fitness"]) < 1e-6 * max(1.0, same["fitness"])
    assert np.allclose(got["transform"][:3, :3], R, atol=5e-3) and np.allclose(got["transform"][:3, 3], t, atol=5e-2)
    assert ic.accepts(got)
    # PointXYZI-strided input (32 bytes per point) gives the same answer
    s32 = np.zeros((ns, 8), np.float32)
    s32[:, :3] = src
    assert np.array_equal(ic.align(s32, tgt)["transform"], got["transform"])


def test_states_and_rejection(icpmod, oracle):
    tgt = scene(4)
    src = (tgt[::3] + np.float32([2.0, 1.0, 0.0])).astype(np.float32)
    ic = icpmod.Icp()
    ic.params.max_iterations = 1
    r1 = ic.align(src, tgt)
    w1 = oracle.icp_align(src, tgt, max_iterations=1)
    assert r1["converged"] and r1["state"] == 1 and r1["iterations"] == 1
    assert np.abs(r1["transform"] - w1["transform"]).max() < TOL
    ic.params.max_iterations = 100
    ic.params.max_corr_dist = 1e-4
    r0 = ic.align(src, tgt)
    assert not r0["converged"] and r0["state"] == 5 and r0["iterations"] == 0
    assert np.array_equal(r0["transform"], np.eye(4, dtype=np.float32))
    ic.params.max_corr_dist = 150.0
    g = np.eye(4, dtype=np.float32)
    g[:3, 3] = [-2.0, -1.0, 0.0]
    rg = ic.align(src, tgt, guess=g)
    assert rg["converged"] and rg["iterations"] <= 3 and rg["fitness"] < 1e-9
    wrong = ic.align(src, scene(9))
    assert not ic.accepts(wrong)                                       # a different place: fitness > 0.3
    assert abs(wrong["fitness"] - oracle.icp_align(src, scene(9))["fitness"]) < 1e-3 * wrong["fitness"]


@pytest.mark.parametrize("ns,nt", [(40000, 3000), (300, 70000), (1500, 200000), (129, 40), (3, 1), (5000, 5000)])
def test_shapes_of_the_persistent_kernel(icpmod, oracle, ns, nt):
    """what the (source block) x (target slice) grid has to cope with: more source blocks than workgroups (several blocks per
    workgroup, points re-read instead of kept in registers), a few source points against many targets, target slices that
    do not fit the LDS tile (re-loaded every iteration), slices of a single point, three points against one.  Compared with
    the oracle adding in the device's order: same iteration count, pose to 1e-6 (capped iterations keep the oracle quick)."""
    rng = np.random.default_rng(ns + nt)
    tgt = scene(7, max(nt, 40))
    tgt = tgt[rng.choice(len(tgt), nt, replace=len(tgt) < nt)]
    R, t = rot(0.03, 0.005, -0.004), np.array([0.4, -0.2, 0.02])
    sub = tgt[rng.choice(nt, ns, replace=True)] + rng.normal(0, 0.05, (ns, 3))
    src = ((sub - t) @ R).astype(np.float32)
    ic = icpmod.Icp()
    ic.params.max_iterations = 6
    got = ic.align(src, tgt)
    want = oracle.icp_align(src, tgt, max_iterations=6, sum_order=oracle.ICP_SUM_TREE)
    assert got["iterations"] == want["iterations"] and got["state"] == want["state"] and got["converged"] == want["converged"]
    assert np.abs(got["transform"] - want["transform"]).max() < 1e-6
    assert abs(got["fitness"] - want["fitness"]) < 1e-6 * max(1.0, want["fitness"])


def test_empty_clouds(icpmod):
    ic = icpmod.Icp()
    pts = scene(3, 200)
    for src, tgt in ((pts[:0], pts), (pts, pts[:0]), (pts[:2], pts)):
        r = ic.align(src, tgt)
        assert not r["converged"] and r["state"] == 5 and r["iterations"] == 0
        assert np.array_equal(r["transform"], np.eye(4, dtype=np.float32))
