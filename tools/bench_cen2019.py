#!/usr/bin/env python3
"""Time cen2019 keypoint extraction on MulRan-shape synthetic polar images: the synchronous single-scan host entry, the
batched host entry, and the batched device entry (images resident in HBM, nothing returns to the host)."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import _rsx, cen2019, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
imgs = [synth.polar_image(100 + i)[0] for i in range(4)]
c = cen2019.Cen2019(rows=imgs[0].shape[0], cols=3360)
n = 0
for i in range(3):
    n = len(c.extract(imgs[i % 4]))
t0 = time.perf_counter()
for i in range(reps):
    n = len(c.extract(imgs[i % 4]))
dt = (time.perf_counter() - t0) / reps
print(f"cen2019 extract (host buffers, single): {dt*1e3:.3f} ms per 400x3360 scan ({1/dt:.1f} scans/s), {n} keypoints")
stack = np.stack([imgs[i % 4] for i in range(batch)])
c.extract_batch(stack)
t0 = time.perf_counter()
for i in range(3):
    c.extract_batch(stack)
dt = (time.perf_counter() - t0) / 3 / batch
print(f"cen2019 extract_batch (host buffers, {batch} scans per call): {dt*1e3:.4f} ms per scan ({1/dt:.0f} scans/s)")
import torch
d = torch.from_numpy(stack).cuda()
tg = torch.zeros((batch, 20000, 2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(batch, dtype=torch.int32, device="cuda")
p = _rsx.Cen2019Params(10000, 58)
s = torch.cuda.current_stream().cuda_stream
def run():
    _rsx.check(_rsx.lib().rsx_cen2019_extract_batch_device(c._h, d.data_ptr(), batch, stack.strides[0], stack.shape[2], 11, C.byref(p), None, 0,
                                                          0.0595, tg.data_ptr(), None, 20000, cnt.data_ptr(), s))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps / batch
print(f"cen2019 extract_batch_device ({batch} resident scans per call): {dt*1e6:.2f} us per scan ({1/dt:.0f} scans/s), counts {cnt[:4].tolist()}")
