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
