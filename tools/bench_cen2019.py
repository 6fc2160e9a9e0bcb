#!/usr/bin/env python3
"""Time rsx_cen2019_extract on MulRan-shape synthetic polar images (host buffers in/out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import cen2019, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
imgs = [synth.polar_image(100 + i)[0] for i in range(4)]
c = cen2019.Cen2019(rows=imgs[0].shape[0], cols=3360)
n = 0
for i in range(3):
    n = len(c.extract(imgs[i % 4]))
t0 = time.perf_counter()
for i in range(reps):
    n = len(c.extract(imgs[i % 4]))
dt = (time.perf_counter() - t0) / reps
print(f"cen2019 extract: {dt*1e3:.3f} ms per 400x3360 scan ({1/dt:.1f} scans/s), {n} keypoints")
