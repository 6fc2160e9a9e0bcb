#!/usr/bin/env python3
"""Static check of the filter kernels' hand-issued LDS reads (inline-asm ds_read_b128 + counted
s_waitcnt): no instruction may touch the destination registers of a read that can still be in
flight.  Usage: tools/check_lds_ring.py [direct|spectral|spectral2]  (compiles sc_filter.hip / sc_spec.hip to ISA
with hipcc and scans sc_filter_kernel / sc_spec_filter_kernel)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "navtech-radar-slam_amd", "csrc")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


KERNELS = {"direct": ("sc_filter.hip", "sc_filter_kernel", []),
           "spectral": ("sc_spec.hip", "sc_spec_filter_kernel", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
           "spectral2": ("sc_spec.hip", "sc_spec2_filter_kernel", ["-mllvm", "-amdgpu-mfma-vgpr-form"])}


def main():
    src, kernel, extra = KERNELS[sys.argv[1] if len(sys.argv) > 1 else "direct"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "f.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + extra + ["-x", "hip",
                               os.path.join(CSRC, src), "-S", "--cuda-device-only", "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    # the kernel's body starts at its label (mangled name + ":"), not at the first mention of the name
    m = re.search(r"^_Z\w*" + kernel + r"\w*:", text, re.M)
    body = text[m.start():]
    body = body[:body.index("s_endpgm")]
    outstanding, bad, nreads = [], 0, 0
    for i, line in enumerate(body.split("\n")):
        t = line.strip()
        if not t or t[0] in ";.":
            continue
        if t.endswith(":"):  # label: a branch target -- nothing may be in flight across it
            if any(r for r, _ in outstanding):
                bad += 1
                print("in-flight read across label", t)
            outstanding = []
            continue
        parts = re.split(r"[ ,]+", t)
        op = parts[0]
        if op.startswith("ds_read"):   # b128 (fragments), b64 / read2st64 (the two-wave kernel's exchange)
            outstanding.append((regs(parts[1]), i))
            nreads += op == "ds_read_b128"
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                while len(outstanding) > int(m.group(1)):
                    outstanding.pop(0)
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op == "s_memtime":
            outstanding.append((set(), i))
            continue
        used = set()
        for p in parts[1:]:
            used |= regs(p)
        for rs, li in outstanding:
            if rs & used:
                bad += 1
                print(f"line {i}: `{t}` touches the destination of the read issued at line {li}")
    print(f"{nreads} ds_read_b128, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
