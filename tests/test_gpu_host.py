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
