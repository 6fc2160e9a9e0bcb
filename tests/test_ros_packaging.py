"""The ROS-facing packaging of the boundary (SURVEY 8b "ORORA package surface" / "build surface"): the `orora`
package with run_orora.launch that the reference's top-level launch file includes, and the sc_pgo CMakeLists with
the USE_RSX switch.  ROS is absent here, so: the launch/CMake files are checked against the reference's own launch
graph, and the orora package is configured and built with plain CMake (its non-catkin path)."""
import os
import re
import shutil
import subprocess
import xml.etree.ElementTree as ET

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROS = os.path.join(ROOT, "navtech-radar-slam_amd", "ros")
REF_TOP = "/root/reference/launch/navtech_radar_slam_mulran.launch"


def test_run_orora_launch_matches_the_reference_include():
    t = ET.parse(os.path.join(ROS, "orora", "launch", "run_orora.launch")).getroot()
    args = {a.get("name") for a in t.findall("arg")}
    assert {"seq_dir", "do_slam"} <= args
    node = t.find("node")
    assert node.get("pkg") == "orora" and "seq_dir:=$(arg seq_dir)" in node.get("args") and "do_slam:=$(arg do_slam)" in node.get("args")
    pkg = ET.parse(os.path.join(ROS, "orora", "package.xml")).getroot()
    assert pkg.find("name").text == "orora"
    cm = open(os.path.join(ROS, "orora", "CMakeLists.txt")).read()
    assert f"add_executable({node.get('type')} " in cm and "RSX_WITH_ROS" in cm
    if os.path.exists(REF_TOP):  # the reference includes $(find orora)/launch/run_orora.launch with exactly these args
        inc = ET.parse(REF_TOP).getroot().find("include")
        assert inc.get("file") == "$(find orora)/launch/run_orora.launch"
        assert {a.get("name") for a in inc.findall("arg")} <= args
    src = open(os.path.join(ROOT, "navtech-radar-slam_amd", "host", "odometry.cpp")).read()
    assert '"/orora/odom"' in src and '"/orora/cloud_local"' in src    # sc_pgo.launch:6-7 remaps these two


def test_sc_pgo_cmake_keeps_the_reference_target_and_adds_use_rsx():
    cm = open(os.path.join(ROS, "sc_pgo", "CMakeLists.txt")).read()
    assert "project(sc_pgo)" in cm and "option(USE_RSX" in cm
    assert "add_executable(alaserPGO src/laserPosegraphOptimization.cpp)" in cm                      # shim build
    assert "add_executable(alaserPGO src/laserPosegraphOptimization.cpp include/scancontext/Scancontext.cpp)" in cm
    assert re.search(r"include_directories\(BEFORE[^)]*navtech-radar-slam_amd/host", cm)
    if os.path.exists("/root/reference/pgo/SC-A-LOAM/CMakeLists.txt"):
        ref = open("/root/reference/pgo/SC-A-LOAM/CMakeLists.txt").read()
        for comp in re.search(r"find_package\(catkin REQUIRED COMPONENTS(.*?)\)", ref, re.S).group(1).split():
            assert comp in cm, comp                                                               # same catkin components
        for pkg in ("PCL", "OpenCV", "OpenMP", "GTSAM"):
            assert f"find_package({pkg} REQUIRED" in cm


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_orora_package_builds_with_plain_cmake(tmp_path):
    if not os.path.exists(os.path.join(ROOT, "navtech-radar-slam_amd", "librsx.so")):
        pytest.skip("librsx.so not built")
    b = str(tmp_path / "build")
    r = subprocess.run(["cmake", "-S", os.path.join(ROS, "orora"), "-B", b], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", b], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(b, "orora_odometry")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)      # no seq_dir: usage error, not a crash
    assert r.returncode == 1 and "usage" in r.stderr
