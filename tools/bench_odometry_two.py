#!/usr/bin/env python3
"""Two rsx_odometry handles pushing resident sequences from two host threads at once: does the device have room beside one handle's
two windows in flight?  (aggregate scans/s against one handle alone)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import odometry, synth
n_unique, n_scans = 8, 512
imgs, az, poses, stamps = synth.polar_sequence(11, n_unique)
order, i, step = [], 0, 1
while len(order) < n_scans:
    order.append(i)
    if not 0 <= i + step < n_unique:
        step = -step
    i += step
seq = np.ascontiguousarray(imgs[np.asarray(order)])
d = torch.from_numpy(seq).cuda()
torch.cuda.synchronize()
ods = [odometry.Odometry(400, 3360) for _ in range(2)]
for od in ods:
    od.push(seq[:70], az)
def run(od, reps, out):
    t0 = time.perf_counter()
    for _ in range(reps):
        od.reset()
        od.push(seq, az, device_ptr=d.data_ptr())
    out.append(time.perf_counter() - t0)
for trial in range(3):
    o = []
    run(ods[0], 2, o)
    one = 2 * n_scans / o[0]
    outs = [[], []]
    th = [threading.Thread(target=run, args=(ods[k], 2, outs[k])) for k in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    both = 4 * n_scans / (time.perf_counter() - t0)
    print(f"one handle {one:.0f} scans/s, two handles together {both:.0f} scans/s")
