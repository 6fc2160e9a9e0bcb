#!/usr/bin/env python3
"""knnMatch(2) + ratio between the consecutive scans of a window, alone: 65 slots x ~1900 keypoints, 40 % with a descriptor
(what a 200 m scan leaves inside the 250 m Cartesian image), random 256-bit descriptors with planted near-copies."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import frontend

rng = np.random.default_rng(5)
slots, stride = 65, 2048
counts = rng.integers(1800, 2000, slots).astype(np.int32)
desc = rng.integers(0, 256, (slots, stride, 32), dtype=np.uint8)
valid = (rng.uniform(size=(slots, stride)) < 0.4).astype(np.uint8)
for s in range(1, slots):  # planted matches: a third of the valid keypoints reappear with a few bits flipped
    src = rng.integers(0, counts[s - 1], 600)
    dst = rng.integers(0, counts[s], 600)
    desc[s, dst] = desc[s - 1, src] ^ (rng.uniform(size=(600, 32)) < 0.02).astype(np.uint8)
for s in range(slots):
    valid[s, counts[s]:] = 0
fe = frontend.Frontend(400, 3360)
d_desc, d_valid, d_cnt = torch.from_numpy(desc).cuda(), torch.from_numpy(valid).cuda(), torch.from_numpy(counts).cuda()
fwd = torch.zeros((slots - 1, stride), dtype=torch.int32, device="cuda")
bwd = torch.zeros_like(fwd)
st = torch.cuda.current_stream().cuda_stream
run = lambda: fe.match_consecutive_device(d_desc.data_ptr(), d_valid.data_ptr(), d_cnt.data_ptr(), stride, 0, slots - 1, 0.8, fwd.data_ptr(), bwd.data_ptr(), stream=st)
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
f = fwd.cpu().numpy()
print(f"match_consecutive: {dt * 1e6:.1f} us per window of {slots - 1} pairs, {int((f >= 0).sum())} forward matches, checksum {int(f.astype(np.int64).sum())}")
