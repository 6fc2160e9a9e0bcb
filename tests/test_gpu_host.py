"""The C++ host adapters above the C-ABI, run as the reference's node would use them."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "navtech-radar-slam_amd", "host")


def test_scmanager_shim_two_threads():
    """sc_shim_demo: SCManager shim driven like laserPosegraphOptimization.cpp (writer thread at
    PGO.cpp:492, reader thread at PGO.cpp:561); every planted revisit must be reported."""
    exe = os.path.join(HOST, "sc_shim_demo")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Loop detected! - between" in r.stdout
    assert "keyframes=120" in r.stdout


@pytest.mark.parametrize("matcher", ["orb", "nn"])
def test_file_based_odometry_entry(tmp_path, matcher):
    """odometry: <seq_dir>/polar_oxford_form/*.png -> cen2019 keypoints -> association -> ORORA,
    driven like the reference launch graph drives the upstream odometry.cpp (seq_dir arg).
    Four scans of the same scene from a parked sensor (independent speckle per scan): stamps must
    come from the image metadata and the accumulated pose must stay at the origin."""
    import numpy as np
    from PIL import Image
    from navtech_radar_slam_amd import synth
    d = tmp_path / "seq" / "polar_oxford_form"
    d.mkdir(parents=True)
    for i in range(4):
        img, _, _ = synth.polar_image(5, n_targets=900, noise_seed=100 + i, t0=1_560_000_000_000_000_000 + i * 250_000_000)
        Image.fromarray(img, mode="L").save(str(d / f"{1560000000000000000 + i * 250000000}.png"))
    exe = os.path.join(HOST, "odometry")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, f"seq_dir:={tmp_path / 'seq'}", "do_slam:=true", "--matcher", matcher], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [line.split() for line in r.stdout.strip().splitlines()]
    assert len(rows) == 4 and rows[0][1:4] == ["0.000000", "0.000000", "0.000000"]
    stamps = [int(x[0]) for x in rows]
    assert stamps == sorted(stamps) and stamps[1] - stamps[0] == 250_000_000
    pose = np.array([[float(v) for v in x[1:4]] for x in rows])
    assert np.abs(pose[:, :2]).max() < 0.1 and np.abs(pose[:, 2]).max() < 2e-3, pose
    # orb: ORB-style descriptors + Hamming knnMatch(2) + ratio + cross check on the GPU (the upstream front end);
    # nn: the round-1 stand-in (mutual nearest neighbours in the sensor frame)
    print(matcher, "matches per frame", [int(x[5]) for x in rows[1:]])
    assert all(int(x[4]) > 300 for x in rows) and all(int(x[5]) > (30 if matcher == "orb" else 100) for x in rows[1:])
    # a missing sequence directory is an error exit, not a crash
    r = subprocess.run([exe, str(tmp_path / "nope")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot list" in r.stderr


def _write_clouds(path, clouds):
    import numpy as np
    with open(path, "wb") as f:
        f.write(np.int32(len(clouds)).tobytes())
        for c in clouds:
            c = np.ascontiguousarray(c[:, :4], dtype=np.float32)
            f.write(np.int32(len(c)).tobytes())
            f.write(c.tobytes())


_REF_REPLAY = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import pyoracle as po
from navtech_radar_slam_amd import synth
clouds, _ = synth.keyframe_clouds(int(sys.argv[2]), int(sys.argv[3]), binary_z=False, loop_frac=0.3, min_gap=35, n_points=500)
m = po.RefManager(po.ORDER_EIGEN_SSE2, dist_thres=0.45)
for i, c in enumerate(clouds):
    m.add_points(c)
    lid, yaw = m.detect_loop_closure()
    sys.stdout.write(m.last_log())
    print("RESULT %d %d %.9g" % (i, lid, yaw))
"""


def test_shim_replay_reproduces_reference_stdout(tmp_path, oracle):
    """The SCManager shim in candidate mode against the reference build of Scancontext.cpp (oracle/_ref) on the same
    clouds, both in fresh processes: identical loop ids and yaws AND identical "[Loop found] / [Not loop] Nearest
    distance: ..." log lines (Scancontext.cpp:406,412), including the precision(3) switch after the first miss."""
    import sys
    from navtech_radar_slam_amd import synth
    root = os.path.dirname(HOST.rstrip("/")).rsplit("/navtech-radar-slam_amd", 1)[0]
    try:
        oracle.RefSC(oracle.ORDER_EIGEN_SSE2)
    except FileNotFoundError as e:
        pytest.skip(str(e))
    seed, n = 321, 90
    clouds, _ = synth.keyframe_clouds(seed, n, binary_z=False, loop_frac=0.3, min_gap=35, n_points=500)
    p = tmp_path / "clouds.bin"
    _write_clouds(p, clouds)
    exe = os.path.join(HOST, "sc_shim_demo")
    got = subprocess.run([exe, "--replay", str(p)], capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, got.stdout + got.stderr
    want = subprocess.run([sys.executable, "-c", _REF_REPLAY, root, str(seed), str(n)], capture_output=True, text=True, timeout=300)
    assert want.returncode == 0, want.stderr
    got_lines = [ln for ln in got.stdout.splitlines() if not ln.startswith(("HELPERS", "MEMBERS"))]
    assert got_lines == want.stdout.splitlines()
    assert sum(ln.startswith("[Loop found]") for ln in got_lines) >= 3 and sum(ln.startswith("[Not loop]") for ln in got_lines) >= 10
    # the public helpers printed at the end, against the oracle
    import numpy as np
    h = [ln for ln in got.stdout.splitlines() if ln.startswith("HELPERS")][0].split()[1:]
    a, b = oracle.make_scancontext(clouds[-1]), oracle.make_scancontext(clouds[-2])
    assert float(h[0]) == oracle.ringkey(a)[3] and float(h[1]) == oracle.sectorkey(a)[7]
    assert int(h[2]) == oracle.fast_align(oracle.sectorkey(a), oracle.sectorkey(b))
    d, s = oracle.distance(a, b)
    assert (float(h[3]), int(h[4])) == (d, s)
    assert float(h[5]) == oracle.dist_direct(a, b) or (np.isnan(float(h[5])) and np.isnan(oracle.dist_direct(a, b)))
    # the public data members (polarcontexts_ & co., Scancontext.h:110-115) read through the shim's views
    m = [ln for ln in got.stdout.splitlines() if ln.startswith("MEMBERS")][0].split()[1:]
    assert int(m[0]) == n and float(m[1]) == float(a.sum())
    assert np.float32(float(m[2])) == oracle.ringkey_f32(a)[3] and float(m[3]) == oracle.ringkey(a)[3] and float(m[4]) == oracle.sectorkey(a)[7]


def test_shim_multi_device_detector(tmp_path, oracle):
    """setDevices: one C++ process, the database sharded over several GPU handles (here: three shards on device 0, the
    box has one GPU), detector in exhaustive mode.  Results = the oracle's exhaustive search over the reference's
    frozen prefix, keyframe by keyframe."""
    import numpy as np
    from navtech_radar_slam_amd import synth
    clouds, _ = synth.keyframe_clouds(99, 80, binary_z=True, loop_frac=0.3, min_gap=35, n_points=400)
    p = tmp_path / "clouds.bin"
    _write_clouds(p, clouds)
    exe = os.path.join(HOST, "sc_shim_demo")
    got = subprocess.run([exe, "--replay", str(p), "--devices", "0,0,0"], capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, got.stdout + got.stderr
    res = [ln.split() for ln in got.stdout.splitlines() if ln.startswith("RESULT")]
    assert len(res) == 80
    m = oracle.Manager(dist_thres=0.45)
    counter, tree = 0, 0
    loops = 0
    for i, c in enumerate(clouds):
        m.add_points(c)
        n = i + 1
        want = (-1, 0.0)
        if n >= 31:
            if counter % 30 == 0:
                tree = n - 30
            counter += 1
            hit = m.exhaustive(m.descriptor(i), n_eligible=tree, k=1)[0]
            if hit["dist"] < 1e7:
                lid = int(hit["index"]) if hit["dist"] < 0.45 else -1
                want = (lid, float(np.float32(np.float64(np.float32(hit["shift"] * 6.0)) * np.pi / 180.0)))
        assert int(res[i][2]) == want[0], (i, res[i], want)
        assert float(res[i][3]) == pytest.approx(want[1], abs=0, rel=1e-7), (i, res[i], want)
        loops += want[0] >= 0
    assert loops >= 3


def test_ros_free_replay_of_the_loop_closure_side(tmp_path, oracle):
    """pgo_replay: a recording of /orora/odom + /orora/cloud_local (ROS 1 wire bytes, host/rosmsg.h) through what
    process_pg / performSCLoopClosure do with those messages (PGO.cpp:417-492, 556-571) on the GPU: stamp pairing,
    keyframe selection by travelled distance, VoxelGrid(0.4) + ScanContext build, detection per keyframe.  The loops
    reported must be the oracle's (oracle VoxelGrid + oracle ScanContext on the same keyframes)."""
    import struct
    import numpy as np
    from navtech_radar_slam_amd import synth

    def s(x):
        return struct.pack("<I", len(x)) + x.encode()

    def header(seq, t_ns, frame):
        return struct.pack("<III", seq, t_ns // 10**9, t_ns % 10**9) + s(frame)

    clouds, _ = synth.keyframe_clouds(41, 70, binary_z=True, loop_frac=0.3, min_gap=35, n_points=900)
    rec = bytearray(b"RSXREPLAY1")
    t0 = 1_560_000_000_000_000_000
    kept = []
    for i, c in enumerate(clouds):
        t = t0 + i * 250_000_000
        # the vehicle advances 1.5 m per scan along x for the first 20 scans (keyframe_meter_gap 2.0: every second one
        # is a keyframe), then 2.5 m per scan (every scan is one)
        xpos = 1.5 * i if i < 20 else 30.0 + 2.5 * (i - 20)
        od = header(i, t, "odom") + s("radar") + struct.pack("<3d", xpos, 0.0, 0.0) + struct.pack("<4d", 0, 0, 0, 1) + bytes(8 * 78)
        body = b"".join(struct.pack("<8f", p[0], p[1], p[2], 1.0, p[3], 0, 0, 0) for p in c)
        pc = header(i, t, "radar") + struct.pack("<II", 1, len(c)) + struct.pack("<I", 4)
        for name, off in (("x", 0), ("y", 4), ("z", 8), ("intensity", 16)):
            pc += s(name) + struct.pack("<IBI", off, 7, 1)
        pc += struct.pack("<BII", 0, 32, 32 * len(c)) + struct.pack("<I", len(body)) + body + struct.pack("<B", 1)
        if i == 5:   # a stale odometry message (older than the next cloud) must be dropped, PGO.cpp:425-426
            stale = header(999, t - 100_000_000, "odom") + s("radar") + struct.pack("<3d", -50.0, 0, 0) + struct.pack("<4d", 0, 0, 0, 1) + bytes(8 * 78)
            rec += struct.pack("<BI", 0, len(stale)) + stale
        rec += struct.pack("<BI", 0, len(od)) + od + struct.pack("<BI", 1, len(pc)) + pc
    p = tmp_path / "run.rsxreplay"
    p.write_bytes(bytes(rec))
    exe = os.path.join(HOST, "pgo_replay")
    r = subprocess.run([exe, str(p), "--keyframe_meter_gap", "2.0", "--sc_dist_thres", "0.45"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the oracle on the same keyframes
    m = oracle.Manager(dist_thres=0.45)
    want = []
    acc = 1e6
    for i, c in enumerate(clouds):
        acc += 0.0 if i == 0 else (1.5 if i <= 20 else 2.5)
        if not acc > 2.0:
            continue
        acc = 0.0
        ds, _ = oracle.voxelgrid_filter(c, 0.4)
        m.add_points(ds)
        if len(m) < 30:
            continue
        lid, _, _, _ = m.detect_loop_closure()
        if lid != -1:
            want.append(f"Loop detected! - between {lid} and {len(m) - 1}")
    lines = r.stdout.strip().splitlines()
    assert lines[:-1] == want and len(want) >= 1, (lines, want)
    assert lines[-1] == f"frames=70 keyframes={len(m)} loops={len(want)} dropped_odom=1"


def test_odometry_recording_feeds_the_replay(tmp_path):
    """odometry --record writes the two topics as ROS 1 wire bytes; pgo_replay reads them back."""
    from PIL import Image
    from navtech_radar_slam_amd import synth
    d = tmp_path / "seq" / "polar_oxford_form"
    d.mkdir(parents=True)
    for i in range(3):
        img, _, _ = synth.polar_image(5, n_targets=900, noise_seed=100 + i, t0=1_560_000_000_000_000_000 + i * 250_000_000)
        Image.fromarray(img, mode="L").save(str(d / f"{1560000000000000000 + i * 250000000}.png"))
    rec = tmp_path / "rec.rsxreplay"
    r = subprocess.run([os.path.join(HOST, "odometry"), str(tmp_path / "seq"), "--record", str(rec)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert rec.read_bytes()[:10] == b"RSXREPLAY1"
    r = subprocess.run([os.path.join(HOST, "pgo_replay"), str(rec), "--keyframe_meter_gap", "-1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1] == "frames=3 keyframes=3 loops=0 dropped_odom=0"


def _recording(clouds, xpos, yaw=None):
    """RSXREPLAY1 bytes of /orora/odom + /orora/cloud_local for clouds[i] seen at (xpos[i], 0, 0) with heading yaw[i]"""
    import struct
    import numpy as np

    def st(x):
        return struct.pack("<I", len(x)) + x.encode()

    def header(seq, t_ns, frame):
        return struct.pack("<III", seq, t_ns // 10**9, t_ns % 10**9) + st(frame)

    rec = bytearray(b"RSXREPLAY1")
    t0 = 1_560_000_000_000_000_000
    for i, c in enumerate(clouds):
        t = t0 + i * 250_000_000
        h = 0.0 if yaw is None else float(yaw[i])
        quat = struct.pack("<4d", 0.0, 0.0, np.sin(h / 2), np.cos(h / 2))
        od = header(i, t, "odom") + st("radar") + struct.pack("<3d", float(xpos[i]), 0.0, 0.0) + quat + bytes(8 * 78)
        body = b"".join(struct.pack("<8f", p[0], p[1], p[2], 1.0, p[3], 0, 0, 0) for p in c)
        pc = header(i, t, "radar") + struct.pack("<II", 1, len(c)) + struct.pack("<I", 4)
        for name, off in (("x", 0), ("y", 4), ("z", 8), ("intensity", 16)):
            pc += st(name) + struct.pack("<IBI", off, 7, 1)
        pc += struct.pack("<BII", 0, 32, 32 * len(c)) + struct.pack("<I", len(body)) + body + struct.pack("<B", 1)
        rec += struct.pack("<BI", 0, len(od)) + od + struct.pack("<BI", 1, len(pc)) + pc
    return bytes(rec)


def test_replay_verifies_loops_and_saves_the_map(tmp_path, oracle):
    """pgo_replay --verify-loops --save-map: behind every "Loop detected!" the chain of doICPVirtualRelative (PGO.cpp:355-406)
    on the keyframe clouds kept in HBM, with the reference's "[SC loop] ICP fitness test ..." lines, and at the end the cloud
    pubMap builds (PGO.cpp:631-655) as a PCD file.  A street driven twice: the second pass closes loops that the ICP gate
    accepts.  Verdicts and the map must be the oracle's (VoxelGrid + ScanContext + loop-verification oracles)."""
    import re
    import numpy as np
    from test_oracle_loopverify import street_drive
    clouds, pose6 = street_drive(seed=3, n=80, step=2.5, revisit_at=48)
    rec = tmp_path / "loop.rsxreplay"
    rec.write_bytes(_recording(clouds, pose6[:, 0], pose6[:, 5]))
    pcdf = tmp_path / "map.pcd"
    r = subprocess.run([os.path.join(HOST, "pgo_replay"), str(rec), "--keyframe_meter_gap", "2.0", "--sc_dist_thres", "0.45",
                        "--verify-loops", "--save-map", str(pcdf)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the oracle on the same stream: every scan is a keyframe (2.5 m apart; the odometry y is 0 in the recording)
    m = oracle.Manager(dist_thres=0.45)
    poses = pose6.copy()
    poses[:, 1] = 0.0
    kfs, want = [], []
    for i, c in enumerate(clouds):
        ds, _ = oracle.voxelgrid_filter(c, 0.4)
        kfs.append(ds)
        m.add_points(ds)
        if len(m) < 30:
            continue
        lid, _, _, _ = m.detect_loop_closure()
        if lid != -1:
            v = oracle.loop_verify(kfs, lid, len(m) - 1, poses[lid])
            want.append((lid, len(m) - 1, v))
    assert len(want) >= 3 and any(v["accepted"] for _, _, v in want)
    lines = r.stdout.strip().splitlines()
    got_loops = [ln for ln in lines if ln.startswith("Loop detected!")]
    got_icp = [ln for ln in lines if ln.startswith("[SC loop]")]
    assert got_loops == [f"Loop detected! - between {a} and {b}" for a, b, _ in want]
    assert len(got_icp) == len(want)
    for ln, (_, _, v) in zip(got_icp, want):
        mo = re.match(r"\[SC loop\] ICP fitness test (passed|failed) \(([-+0-9.e]+|inf|nan) ([<>]) 0\.3\)\. (Add|Reject) this SC loop\.", ln)
        assert mo, ln
        assert (mo.group(1) == "passed") == (mo.group(3) == "<") == (mo.group(4) == "Add")
        assert abs(float(mo.group(2)) - v["fitness"]) < 2e-2 * max(1.0, v["fitness"])      # (ICPs may stop an iteration apart)
        if abs(v["fitness"] - 0.3) > 2e-2:
            assert (mo.group(1) == "passed") == v["accepted"]
    n_acc = sum(1 for ln in got_icp if "passed" in ln)
    assert n_acc >= 3
    assert lines[-1] == f"frames=80 keyframes=80 loops={len(want)} dropped_odom=0 loops_accepted={n_acc}"
    # the saved map == the oracle's pubMap cloud, bit for bit
    raw = pcdf.read_bytes()
    head, body = raw.split(b"DATA binary\n", 1)
    assert b"FIELDS x y z intensity" in head and b"VERSION 0.7" in head
    got_map = np.frombuffer(body, dtype=np.float32).reshape(-1, 4)
    want_map = oracle.map_build(kfs, poses, skip=2, leaf=0.4)
    assert f"POINTS {len(want_map)}".encode() in head
    assert got_map.shape == want_map.shape and np.array_equal(got_map.view(np.uint32), want_map.view(np.uint32))
