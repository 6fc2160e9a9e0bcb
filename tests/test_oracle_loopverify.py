"""CPU checks of the loop-verification oracle (oracle/loopverify_ref.c) against the reference text it restates
(pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp = PGO.cpp).  PARITY UNPINNED for the PCL / GTSAM pieces (absent from the
reference checkout); what is pinned here is the reference's own control flow: which keyframes, which pose, which gate."""
import numpy as np
import pytest


def rz_ry_rx(roll, pitch, yaw):
    cx, sx, cy, sy, cz, sz = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def street_drive(seed=3, n=40, step=1.0, revisit_at=30):
    """keyframes along a street, then a second pass over the first keyframes (a loop): clouds in the sensor frame + poses"""
    from navtech_radar_slam_amd import synth
    world = synth.World(seed, blocks=4)
    rng = np.random.default_rng(seed + 1)
    xs = np.arange(n) * step
    poses = np.stack([60.0 + xs, np.full(n, 110.0), np.zeros(n)], axis=1)
    poses[revisit_at:, 0] = 60.0 + (np.arange(n - revisit_at)) * step + 0.3     # second pass, 0.3 m further on, 0.4 m to the side
    poses[revisit_at:, 1] = 110.4
    poses[revisit_at:, 2] = 0.03
    pts, off = world.observe(rng, poses, p_detect=1.0, sigma=0.02, clutter_frac=0.0)
    clouds = [pts[off[i]:off[i + 1]] for i in range(n)]
    pose6 = np.zeros((n, 6))
    pose6[:, 0:2] = poses[:, :2]
    pose6[:, 5] = poses[:, 2]
    return clouds, pose6


def test_pose_matrix_is_pcl_get_transformation(oracle):
    """pcl::getTransformation(x, y, z, roll, pitch, yaw) = translation * Rz(yaw) Ry(pitch) Rx(roll), in float (PGO.cpp:206)"""
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = np.concatenate([rng.uniform(-500, 500, 3), rng.uniform(-np.pi, np.pi, 3)])
        t = oracle.pose_matrix(p)
        assert t.dtype == np.float32 and np.array_equal(t[3], [0, 0, 0, 1])
        assert np.allclose(t[:3, :3], rz_ry_rx(*p[3:]), atol=2e-6)
        assert np.array_equal(t[:3, 3], p[:3].astype(np.float32))


def test_submap_moves_every_neighbour_by_the_one_root_pose(oracle):
    """loopFindNearKeyframesCloud (PGO.cpp:329-352): keyframes key - size .. key + size (clipped to the stored range), each
    cloud in ITS OWN local frame, all through local2global with the pose of the ROOT keyframe (PGO.cpp:340), then one
    VoxelGrid.  With one point per keyframe in distinct voxels the output is exactly the moved points."""
    local = [np.array([[10.0 * (i + 1), -3.0 * i, 0.5 * i, float(i)]], dtype=np.float32) for i in range(6)]
    root = np.array([100.0, -50.0, 2.0, 0.02, -0.01, 0.7])
    T = oracle.pose_matrix(root)
    for key, size, want in ((2, 1, [1, 2, 3]), (0, 2, [0, 1, 2]), (5, 3, [2, 3, 4, 5]), (3, 0, [3])):
        got = oracle.loop_submap(local, key, size, root, leaf=0.4)
        moved = np.stack([np.r_[(T[:3, :3] @ local[i][0, :3] + T[:3, 3]).astype(np.float32), local[i][0, 3]] for i in want])
        assert len(got) == len(want)
        assert np.allclose(got[np.lexsort(got.T[::-1])], moved[np.lexsort(moved.T[::-1])], atol=1e-4)
    assert len(oracle.loop_submap(local, 9, 0, root)) == 0 and len(oracle.loop_submap(local, -1, 0, root)) == 0   # PGO.cpp:335-336


def test_verify_accepts_a_revisit_and_rejects_a_stranger(oracle):
    """doICPVirtualRelative (PGO.cpp:355-406): both clouds moved by the LOOP keyframe's pose, ICP, gate converged && fitness
    <= 0.3 (PGO.cpp:385), Euler angles of the final transformation, poseFrom.between(identity) = its inverse"""
    clouds, pose6 = street_drive()
    loop, curr = 2, 32           # keyframe 32 = the second pass at the place of keyframe 2 (+0.3 m, 0.4 m to the side, 0.03 rad)
    r = oracle.loop_verify(clouds, loop, curr, pose6[loop], history_num=0)
    assert r["converged"] and r["accepted"] and r["fitness"] <= 0.3
    assert r["n_source"] > 100 and r["n_target"] > 100
    # both clouds were moved by the loop keyframe's pose r (x, y, yaw = 0 here): a landmark w appears at w in the target and at
    # R(-dyaw)(w - c) + r in the source (c = the current sensor position), so ICP finds R(dyaw) and t = c - R(dyaw) r
    t = r["transform"].astype(np.float64)
    c, rr, dyaw = pose6[curr, :2], pose6[loop, :2], pose6[curr, 5] - pose6[loop, 5]
    want_t = c - rz_ry_rx(0, 0, dyaw)[:2, :2] @ rr
    assert np.abs(t[:2, 3] - want_t).max() < 0.15
    x, y, z, roll, pitch, yaw = r["xyz_rpy"]
    assert abs(yaw - 0.03) < 0.02 and abs(roll) < 1e-2 and abs(pitch) < 1e-2
    assert np.allclose([x, y, z], t[:3, 3], atol=1e-6)
    pose_from = np.eye(4)
    pose_from[:3, :3] = rz_ry_rx(roll, pitch, yaw)
    pose_from[:3, 3] = [x, y, z]
    assert np.allclose(r["relative"] @ pose_from, np.eye(4), atol=1e-9)      # between(identity) = inverse
    # with the reference's +-25 neighbours, every one in its own frame under the one root pose (PGO.cpp:340)
    r25 = oracle.loop_verify(clouds, loop, curr, pose6[loop])
    assert r25["n_target"] > r["n_target"] and r25["converged"]
    # a place that shares nothing with the loop keyframe: converged or not, the fitness gate rejects it
    rng = np.random.default_rng(5)
    stranger = np.c_[rng.uniform(-80, 80, (900, 2)), np.zeros(900), np.ones(900)].astype(np.float32)
    rs = oracle.loop_verify(clouds + [stranger], loop, len(clouds), pose6[loop], history_num=0)
    assert not rs["accepted"] and rs["fitness"] > 0.3


def test_map_takes_every_skipth_keyframe_through_its_own_pose(oracle):
    """pubMap (PGO.cpp:631-655): counter % SKIP_FRAMES == 0, local2global(cloud[i], pose[i]), VoxelGrid"""
    local = [np.array([[1.0 + i, 2.0, 0.0, 5.0]], dtype=np.float32) for i in range(5)]
    poses = np.zeros((5, 6))
    poses[:, 0] = 100.0 * np.arange(5)
    poses[:, 5] = 0.5 * np.arange(5)
    got = oracle.map_build(local, poses, skip=2, leaf=0.4)
    want = []
    for i in (0, 2, 4):
        T = oracle.pose_matrix(poses[i])
        want.append(np.r_[(T[:3, :3] @ local[i][0, :3] + T[:3, 3]).astype(np.float32), 5.0])
    want = np.stack(want)
    assert len(got) == 3 and np.allclose(got[np.argsort(got[:, 0])], want[np.argsort(want[:, 0])], atol=1e-4)
    assert len(oracle.map_build(local, poses, skip=1)) == 5
