"""Static check of the filter kernel's hand-issued LDS reads (no GPU needed: hipcc cross-compiles).

sc_filter_kernel issues its A-fragment reads through inline asm with hand-counted s_waitcnt, which the
compiler cannot see.  tools/check_lds_ring.py compiles the kernel to gfx950 ISA and verifies that no
instruction touches the destination registers of a read that may still be in flight (a violation is
a timing-dependent wrong answer, invisible to most test runs)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("kind", ["direct", "spectral", "spectral2"])
def test_no_instruction_touches_inflight_lds_destinations(kind):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_ring.py"), kind], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 violations" in r.stdout and "ds_read_b128" in r.stdout


def _kernel_resources(src, extra=()):
    """(vgpr_count, vgpr_spill_count, scratch bytes) per kernel name fragment, from the ISA metadata hipcc emits for gfx950"""
    import re
    csrc = os.path.join(ROOT, "navtech-radar-slam_amd", "csrc")
    out = os.path.join("/tmp", f"rsx_static_{os.path.basename(src)}.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, *extra, "-x", "hip",
           "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = open(out).read()
    res = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        res[m.group(1)] = (int(m.group(3)), int(m.group(4)), int(m.group(2)))
    return res


def test_register_budgets_of_the_hand_scheduled_kernels():
    """Compile-time guards for what the schedules rest on (no GPU needed).  The two-wave filter lives on exactly 256 VGPRs per
    wave (two waves per SIMD) with a handful of spills outside its tile loop; a change that pushes fragments into scratch
    inside the loop costs a full s_waitcnt vmcnt(0) drain per reload.  The one-launch insert and the
    spectra kernels must not spill at all."""
    res = _kernel_resources("sc_spec.hip", extra=("-mllvm", "-amdgpu-mfma-vgpr-form"))
    by = lambda frag: next(v for k, v in res.items() if frag in k)   # noqa: E731
    vg, spill, scratch = by("sc_spec2_filter_kernel")
    assert vg == 256 and spill <= 8 and scratch <= 64, (vg, spill, scratch)
    for frag in ("sc_insert_kernel", "sc_spec_db_kernel", "sc_spec_query_kernel"):
        vg, spill, scratch = by(frag)
        assert spill == 0 and scratch == 0, (frag, vg, spill, scratch)
    res = _kernel_resources("sc_window.hip")
    vg, spill, scratch = next(v for k, v in res.items() if "sc_window_kernel" in k)
    # four waves per SIMD (WIN_OCC): the 128-register cap costs 16 spilled registers today (outside the K loop); more is a regression
    assert vg <= 128 and spill <= 16 and scratch <= 72, (vg, spill, scratch)
