"""GPU parity of the keyframe-cloud store and the loop-verification chain (csrc/loopverify.hip through the C-ABI) against
oracle/loopverify_ref.c: submap assembly -> VoxelGrid -> ICP -> gate (laserPosegraphOptimization.cpp:329-406) and the map
cloud (:631-655).  Stored clouds, submaps and maps are BIT-IDENTICAL (float transform in the reference's operation order,
VoxelGrid as voxelgrid_ref.c); the ICP pose agrees to 1e-4 with the oracle adding in the device's order (fp64, fixed tree:
icp_ref.c ICPREF_SUM_TREE) -- for the +-25 submap too, and for a fitness 6e-6 below the gate."""
import numpy as np
import pytest

from test_oracle_loopverify import street_drive

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def lv():
    from navtech_radar_slam_amd import _rsx, loopverify
    assert _rsx.device_count() >= 1
    return loopverify


@pytest.fixture(scope="module")
def drive():
    return street_drive(seed=3, n=64, step=1.0, revisit_at=48)


def same_set(a, b):
    """two (m, 4) clouds hold the same points in the same order, bit for bit"""
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_store_round_trip_and_strides(lv):
    rng = np.random.default_rng(1)
    kf = lv.KeyframeStore()
    clouds = [rng.normal(0, 30, (n, 4)).astype(np.float32) for n in (700, 1, 0, 1500)]
    for i, c in enumerate(clouds):
        assert kf.add(c) == i
    assert kf.size() == (4, sum(len(c) for c in clouds))
    for i, c in enumerate(clouds):
        assert same_set(kf.get(i), c.reshape(-1, 4))
    # pcl::PointXYZI layout: 32 bytes per point, intensity at byte 16
    p32 = np.zeros((300, 8), np.float32)
    p32[:, :3] = rng.normal(0, 10, (300, 3))
    p32[:, 4] = rng.uniform(0, 1, 300)
    import ctypes as C
    from navtech_radar_slam_amd._rsx import check
    idx = C.c_int32()
    check(kf._L.rsx_kfstore_add(kf.handle, p32.ctypes.data, C.c_size_t(300), C.c_size_t(32), 16, C.byref(idx)))
    assert idx.value == 4 and same_set(kf.get(4), np.c_[p32[:, :3], p32[:, 4]])
    xyz = rng.normal(0, 10, (50, 3)).astype(np.float32)
    assert kf.add(xyz) == 5 and same_set(kf.get(5), np.c_[xyz, np.zeros(50, np.float32)])
    with pytest.raises(Exception):
        kf.get(6)


@pytest.mark.parametrize("key,size", [(10, 0), (10, 25), (2, 25), (60, 25), (30, 3)])
def test_submap_bit_identical(lv, oracle, drive, key, size):
    clouds, pose6 = drive
    kf = lv.KeyframeStore()
    for c in clouds:
        kf.add(c)
    root = pose6[key] + np.array([0, 0, 0.3, 0.01, -0.02, 0.0])       # a full 6-DoF root pose
    got = kf.submap(key, size, root)
    want = oracle.loop_submap(clouds, key, size, root)
    assert len(want) > 100 and same_set(got, want)


@pytest.mark.parametrize("loop,curr,hist", [(2, 50, 25), (2, 50, 0), (10, 58, 25), (5, 40, 25)])
def test_verify_matches_oracle(lv, oracle, drive, loop, curr, hist):
    """(2, 50) and (10, 58): the second pass over the same street (accepted); (5, 40): 35 m down the road"""
    clouds, pose6 = drive
    kf = lv.KeyframeStore()
    for c in clouds:
        kf.add(c)
    kf.params.history_keyframe_search_num = hist
    got = kf.verify(loop, curr, pose6[loop])
    want = oracle.loop_verify(clouds, loop, curr, pose6[loop], history_num=hist, sum_order=oracle.ICP_SUM_TREE)
    assert (got["n_source"], got["n_target"]) == (want["n_source"], want["n_target"])
    assert got["converged"] == want["converged"] and got["accepted"] == want["accepted"]
    assert got["iterations"] == want["iterations"] and got["state"] == want["state"]
    # The two ICPs see identical clouds (above), find identical correspondences (same float expression, same tie rule) and add
    # their moments in the same order: 1e-4 on the pose whatever the target -- scan against scan (hist = 0) or the
    # reference's +-25 submap, 51 clouds in 51 different frames under ONE pose (PGO.cpp:340), a smear around the root with
    # many near-ties among the nearest neighbours.  (Rounds 1-4 added in another order than the oracle -- parallel fp64
    # atomics against sequential float -- and on the +-25 target a last-bit difference in a step flipped a correspondence:
    # the descents settled a centimetre apart and the check was 3e-2.  Below: that comparison, kept as what it is.)
    tol = TOL
    assert np.abs(got["transform"] - want["transform"]).max() < tol * max(1.0, np.abs(want["transform"]).max())
    assert abs(got["fitness"] - want["fitness"]) < TOL * max(1.0, want["fitness"])
    assert np.abs(got["xyz_rpy"] - want["xyz_rpy"]).max() < tol * max(1.0, np.abs(want["xyz_rpy"]).max())
    assert np.abs(got["relative"] - want["relative"]).max() < tol * max(1.0, np.abs(want["relative"]).max())
    seq = oracle.loop_verify(clouds, loop, curr, pose6[loop], history_num=hist)   # sequential float sums: another order
    assert got["converged"] == seq["converged"] and got["accepted"] == seq["accepted"] and abs(got["iterations"] - seq["iterations"]) <= 2
    assert np.abs(got["transform"] - seq["transform"]).max() < (TOL if hist == 0 else 3e-2) * max(1.0, np.abs(seq["transform"]).max())
    assert abs(got["fitness"] - seq["fitness"]) < (TOL if hist == 0 else 1e-2) * max(1.0, seq["fitness"])
    # what the caller does with the result is internally consistent whatever the tolerance: Euler angles and the
    # relative pose are functions of the returned transformation
    t = got["transform"].astype(np.float64)
    assert np.allclose(got["xyz_rpy"][:3], t[:3, 3], atol=1e-6) and abs(got["xyz_rpy"][5] - np.arctan2(t[1, 0], t[0, 0])) < 1e-6
    assert np.allclose(got["relative"][:3, :3] @ t[:3, :3], np.eye(3), atol=1e-5)
    if hist == 0 and curr - loop == 48:
        assert got["accepted"]


@pytest.mark.parametrize("sigma,accepted", [(0.38233837890625, True), (0.38433837890625, True), (0.38633837890625, False)])
def test_fitness_at_the_gate(lv, oracle, drive, sigma, accepted):
    """the gate is fitness <= 0.3 (PGO.cpp:384): a revisit blurred until its fitness is 0.2976, 0.299994 and 0.3026.  The
    middle one is 6e-6 below the gate when the moments are added in the device's order and 2e-7 ABOVE it when they are added
    sequentially in float (a different descent: 10 iterations against 8) -- the verdict is the device order's"""
    clouds, pose6 = drive
    rng = np.random.default_rng(77)
    noise = rng.normal(0, 1, clouds[50].shape).astype(np.float32)
    noise[:, 2:] = 0
    cl = list(clouds)
    cl[50] = (clouds[50] + np.float32(sigma) * noise).astype(np.float32)
    kf = lv.KeyframeStore()
    for c in cl:
        kf.add(c)
    kf.params.history_keyframe_search_num = 0
    got = kf.verify(2, 50, pose6[2])
    want = oracle.loop_verify(cl, 2, 50, pose6[2], history_num=0, sum_order=oracle.ICP_SUM_TREE)
    assert want["accepted"] == accepted and abs(want["fitness"] - 0.3) < 3e-3
    assert got["accepted"] == want["accepted"] and got["converged"] == want["converged"] and got["iterations"] == want["iterations"]
    assert abs(got["fitness"] - want["fitness"]) < 1e-6
    assert np.abs(got["transform"] - want["transform"]).max() < 1e-5


def test_map_bit_identical(lv, oracle, drive):
    clouds, pose6 = drive
    kf = lv.KeyframeStore()
    for c in clouds:
        kf.add(c)
    for skip, leaf, nposes in ((2, 0.4, len(clouds)), (1, 0.4, len(clouds)), (3, 1.0, 40)):
        got = kf.build_map(pose6[:nposes], skip=skip, leaf=leaf)
        want = oracle.map_build(clouds[:nposes], pose6[:nposes], skip=skip, leaf=leaf)
        assert len(want) > 1000 and same_set(got, want), (skip, leaf, nposes)


def test_fused_keyframe_add(lv, oracle, drive):
    """rsx_sc_add_keyframe = downSizeFilterScancontext.filter + keyframeLaserClouds.push_back +
    makeAndSaveScancontextAndKeys (PGO.cpp:482-492): the stored cloud is the VoxelGrid of the raw cloud, the descriptor the
    one the two-call path builds"""
    import ctypes as C
    from navtech_radar_slam_amd import scancontext, voxelgrid
    from navtech_radar_slam_amd._rsx import check
    clouds, _ = drive
    kf = lv.KeyframeStore()
    vg = voxelgrid.VoxelGrid()
    a = scancontext.SCManager()
    b = scancontext.SCManager()
    for i, c in enumerate(clouds[:6]):
        raw = np.zeros((len(c), 8), np.float32)                       # pcl::PointXYZI
        raw[:, :3] = c[:, :3] * np.float32(1.0 + 0.001 * i)
        raw[:, 4] = np.float32(0.25 * i)
        idx = C.c_int32()
        check(kf._L.rsx_sc_add_keyframe(a._h, vg._h, kf.handle, raw.ctypes.data, C.c_size_t(len(raw)), C.c_size_t(32), 16,
                                        C.c_float(0.4), C.byref(idx)))
        assert idx.value == i
        want, _ = oracle.voxelgrid_filter(np.c_[raw[:, :3], raw[:, 4]], leaf=0.4)
        assert same_set(kf.get(i), want)
        b.makeAndSaveScancontextAndKeys(want[:, :3])
        assert np.array_equal(a.getConstRefRecentSCD(), b.getConstRefRecentSCD())
    assert kf.size()[0] == 6


def test_bad_arguments(lv, drive):
    clouds, pose6 = drive
    kf = lv.KeyframeStore()
    for c in clouds[:3]:
        kf.add(c)
    with pytest.raises(Exception):
        kf.verify(0, 3, pose6[0])
    with pytest.raises(Exception):
        kf.verify(-1, 1, pose6[0])
    assert len(kf.build_map(pose6[:0])) == 0


def test_empty_and_tiny_keyframes(lv, oracle, drive):
    """an empty current keyframe (nearKeyframes->empty(), PGO.cpp:343-346) and one of two points: not converged, rejected,
    identity -- one of the two clouds of the cooperative VoxelGrid launch is then missing / a single workgroup"""
    clouds, pose6 = drive
    cl = [c.copy() for c in clouds[:20]]
    cl[15] = cl[15][:0]
    cl[16] = cl[16][:2]
    kf = lv.KeyframeStore()
    for c in cl:
        kf.add(c)
    kf.params.history_keyframe_search_num = 3
    for curr in (15, 16):
        got = kf.verify(2, curr, pose6[2])
        want = oracle.loop_verify(cl, 2, curr, pose6[2], history_num=3, sum_order=oracle.ICP_SUM_TREE)
        assert (got["n_source"], got["n_target"]) == (want["n_source"], want["n_target"])
        assert not got["converged"] and not got["accepted"] and got["converged"] == want["converged"] and got["iterations"] == want["iterations"]
        assert np.array_equal(got["transform"], np.eye(4, dtype=np.float32))
    good = kf.verify(2, 14, pose6[2])          # and the store still answers an ordinary candidate afterwards
    ref = oracle.loop_verify(cl, 2, 14, pose6[2], history_num=3, sum_order=oracle.ICP_SUM_TREE)
    assert good["iterations"] == ref["iterations"] and np.abs(good["transform"] - ref["transform"]).max() < TOL
