"""Properties of the ICP oracle (oracle/icp_ref.c): the restatement of pcl::IterativeClosestPoint as
configured at laserPosegraphOptimization.cpp:371-392.  PCL is not available here (parity with it is
unpinned); these tests pin the restatement to the published algorithm's contract on synthetic truth."""
import numpy as np


def rot(yaw, pitch=0.0, roll=0.0):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def scene(seed, n=1500):
    """A structured 'street corner': walls and poles, so that the alignment is well conditioned."""
    rng = np.random.default_rng(seed)
    pts = []
    for _ in range(8):
        a, b = rng.uniform(-40, 40, 2), rng.uniform(-40, 40, 2)
        t = rng.uniform(0, 1, n // 10)[:, None]
        wall = a + t * (b - a)
        pts.append(np.c_[wall, rng.uniform(0, 3, len(wall))])
    pts.append(np.c_[rng.uniform(-40, 40, (n // 5, 2)), rng.uniform(0, 0.3, n // 5)])
    return np.concatenate(pts).astype(np.float32)


def test_rotation_from_covariance(oracle):
    rng = np.random.default_rng(1)
    for _ in range(20):
        R = rot(*rng.uniform(-np.pi, np.pi, 3))
        src = rng.normal(size=(50, 3))
        dst = src @ R.T
        H = (dst - dst.mean(0)).T @ (src - src.mean(0))
        got = oracle.icp_rotation_from_covariance(H)
        assert np.allclose(got, R, atol=1e-9) and abs(np.linalg.det(got) - 1) < 1e-12
    # planar data (rank 2) still gives a proper rotation, never a reflection
    src = np.c_[rng.normal(size=(40, 2)), np.zeros(40)]
    R = rot(0.7)
    H = (src @ R.T).T @ src
    got = oracle.icp_rotation_from_covariance(H)
    assert np.allclose(got, R, atol=1e-9)
    assert np.allclose(oracle.icp_rotation_from_covariance(np.zeros((3, 3))), np.eye(3))


def test_align_recovers_known_transform(oracle):
    tgt = scene(2)
    R, t = rot(0.06, 0.01, -0.015), np.array([0.6, -0.4, 0.05])
    rng = np.random.default_rng(3)
    sub = tgt[rng.choice(len(tgt), 600, replace=False)]
    src = ((sub - t) @ R).astype(np.float32)            # tgt = R src + t
    res = oracle.icp_align(src, tgt)
    T = res["transform"]
    assert res["converged"] and res["state"] in (2, 3, 4)
    assert np.allclose(T[:3, :3], R, atol=2e-4) and np.allclose(T[:3, 3], t, atol=2e-3)
    assert res["fitness"] < 1e-6 and 1 <= res["iterations"] < 100
    # the acceptance test of PGO.cpp:385: a wrong place fails on fitness
    other = scene(9)
    bad = oracle.icp_align(src, other)
    assert bad["fitness"] > 0.3


def test_convergence_states(oracle):
    tgt = scene(4)
    src = (tgt[::3] + np.float32([2.0, 1.0, 0.0])).astype(np.float32)
    r1 = oracle.icp_align(src, tgt, max_iterations=1)
    assert r1["converged"] and r1["state"] == 1 and r1["iterations"] == 1      # ITERATIONS counts as converged
    r0 = oracle.icp_align(src, tgt, max_corr_dist=1e-4)
    assert not r0["converged"] and r0["state"] == 5 and r0["iterations"] == 0  # no correspondences
    assert np.array_equal(r0["transform"], np.eye(4, dtype=np.float32))
    g = np.eye(4, dtype=np.float32)
    g[:3, 3] = [-2.0, -1.0, 0.0]
    rg = oracle.icp_align(src, tgt, guess=g)                                    # a perfect guess: converges at once
    assert rg["converged"] and rg["iterations"] <= 3 and rg["fitness"] < 1e-9


def test_the_two_orders_of_addition(oracle):
    """ICPREF_SUM_TREE (what the device kernel does: fp64, correspondence i into partial sum i mod 1024, a balanced tree
    over the partial sums) against the sequential float sums: the same algorithm, so on a well-conditioned problem the same
    pose to 1e-4 and the same verdicts -- and not the same bits, which is why the device is compared with its own order"""
    tgt = scene(2, 6000)
    R, t = rot(0.1, 0.01, -0.01), np.array([0.5, -0.6, 0.04])
    rng = np.random.default_rng(2)
    sub = tgt[rng.choice(len(tgt), 2000, replace=False)] + rng.normal(0, 0.02, (2000, 3))
    src = ((sub - t) @ R).astype(np.float32)
    a = oracle.icp_align(src, tgt)
    b = oracle.icp_align(src, tgt, sum_order=oracle.ICP_SUM_TREE)
    assert a["converged"] and b["converged"] and abs(a["iterations"] - b["iterations"]) <= 2
    assert np.abs(a["transform"] - b["transform"]).max() < 1e-4 and abs(a["fitness"] - b["fitness"]) < 1e-4 * max(1.0, a["fitness"])
    # fewer than 1024 correspondences, and fewer than 3: the tree has empty leaves
    few = oracle.icp_align(src[:100], tgt, sum_order=oracle.ICP_SUM_TREE)
    ref = oracle.icp_align(src[:100], tgt)
    assert few["converged"] == ref["converged"] and np.abs(few["transform"] - ref["transform"]).max() < 1e-3
    none = oracle.icp_align(src[:2], tgt, sum_order=oracle.ICP_SUM_TREE)
    assert not none["converged"] and none["state"] == 5 and none["iterations"] == 0
