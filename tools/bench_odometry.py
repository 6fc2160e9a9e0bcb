#!/usr/bin/env python3
"""Time rsx_odometry_push_device on a synthetic moving-sensor sequence (images resident in HBM)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import odometry, synth
n_unique = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
imgs, az, poses, stamps = synth.polar_sequence(11, n_unique)
order, i, step = [], 0, 1
while len(order) < n_scans:
    order.append(i)
    if not 0 <= i + step < n_unique:
        step = -step
    i += step
seq = np.ascontiguousarray(imgs[np.asarray(order)])
od = odometry.Odometry(400, 3360)
d = torch.from_numpy(seq).cuda()
torch.cuda.synchronize()
od.push(seq[:200] if len(seq) >= 200 else seq, az)
for _ in range(reps):
    od.reset()
    t0 = time.perf_counter()
    res = od.push(seq, az, device_ptr=d.data_ptr())
    dt = time.perf_counter() - t0
    print(f"resident: {n_scans / dt:.0f} scans/s ({dt / n_scans * 1e6:.1f} us per scan), matches mean {res['n_matches'][1:].mean():.0f}")
od.reset()
t0 = time.perf_counter()
od.push(seq, az)
dt = time.perf_counter() - t0
print(f"host images: {n_scans / dt:.0f} scans/s")
