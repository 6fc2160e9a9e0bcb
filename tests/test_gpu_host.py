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


def test_file_based_odometry_entry(tmp_path):
    """odometry: <seq_dir>/polar_oxford_form/*.png -> cen2019 keypoints -> association -> ORORA,
    driven like the reference launch graph drives the upstream odometry.cpp (seq_dir arg).
    Scene rolled by +2 azimuth rows per frame = sensor yawing by -1.8 deg per frame."""
    import numpy as np
    from PIL import Image
    from navtech_radar_slam_amd import synth
    d = tmp_path / "seq" / "polar_oxford_form"
    d.mkdir(parents=True)
    for i, shift in enumerate([0, 2, 4, 6]):
        img, _, _ = synth.polar_image(5, n_targets=900, shift_rows=shift, t0=1_560_000_000_000_000_000 + i * 250_000_000)
        Image.fromarray(img, mode="L").save(str(d / f"{1560000000000000000 + i * 250000000}.png"))
    exe = os.path.join(HOST, "odometry")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, f"seq_dir:={tmp_path / 'seq'}", "do_slam:=true"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [line.split() for line in r.stdout.strip().splitlines()]
    assert len(rows) == 4 and rows[0][1:4] == ["0.000000", "0.000000", "0.000000"]
    stamps = [int(x[0]) for x in rows]
    assert stamps == sorted(stamps) and stamps[1] - stamps[0] == 250_000_000
    yaw = np.array([float(x[3]) for x in rows])
    step = np.deg2rad(2 * 0.9)
    assert np.allclose(yaw, -step * np.arange(4), atol=2e-3), yaw
    xy = np.array([[float(x[1]), float(x[2])] for x in rows])
    assert np.abs(xy).max() < 0.1                      # pure rotation: no translation
    assert all(int(x[4]) > 300 for x in rows) and all(int(x[5]) > 100 for x in rows[1:])
