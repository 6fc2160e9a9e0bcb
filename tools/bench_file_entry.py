#!/usr/bin/env python3
"""host/odometry on a directory of PNG scans (the bench's drive, 528 scans): total scans/s for --window / --threads settings.
usage: bench_file_entry.py [n_scans] ["W:T" ...]"""
import os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navtech_radar_slam_amd import synth
from PIL import Image
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 528
cfgs = sys.argv[2:] or ["64:64"]
n_unique = 24
imgs, az, poses, stamps = synth.polar_sequence(11, n_unique)
order, i, step = [], 0, 1
while len(order) < n_scans:
    order.append(i)
    if not 0 <= i + step < n_unique:
        step = -step
    i += step
tmp = tempfile.mkdtemp(prefix="rsx_fe_")
try:
    dd = os.path.join(tmp, "polar_oxford_form")
    os.makedirs(dd)
    uniq = []
    for k in range(n_unique):
        pth = os.path.join(tmp, f"u{k}.png")
        Image.fromarray(imgs[k], mode="L").save(pth)
        uniq.append(pth)
    for j, k in enumerate(order):
        os.link(uniq[k], os.path.join(dd, f"{1560000000000000000 + j * 250000000}.png"))
    exe = os.path.join(ROOT, "navtech-radar-slam_amd", "host", "odometry")
    ref = None
    for cfg in cfgs:
        W, T = cfg.split(":")
        out = os.path.join(tmp, f"poses_{W}_{T}.txt")
        for rep in range(2):
            r = subprocess.run([exe, f"seq_dir:={tmp}", "--timing", "--window", W, "--threads", T, "--out", out], capture_output=True, text=True, timeout=600)
            tl = [ln for ln in r.stderr.splitlines() if ln.startswith("timing:")]
            kv = dict(x.split("=") for x in tl[0].split()[1:]) if tl else {"error": r.stderr[-300:]}
            body = open(out).read() if os.path.exists(out) else ""
            if ref is None:
                ref = body
            print(cfg, {k: kv.get(k) for k in ("decode_ms_per_scan_per_thread", "decode_wait_s", "pipeline_scans_per_s", "total_scans_per_s")},
                  "same poses" if body == ref else "POSES DIFFER")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
