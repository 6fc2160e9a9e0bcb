#!/usr/bin/env python3
"""Time the Cartesian + smoothed image kernel of the front end on a window of scans resident in HBM.
usage: bench_fe_cart.py [n_images] [flags ...]   (flags: rsx_frontend_params.flags values to compare, default 0 4)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import frontend, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flag_list = [int(x) for x in sys.argv[2:]] or [0, 4]
imgs = [synth.polar_image(30 + k, n_targets=600)[0] for k in range(8)]
az = synth.polar_image(30, n_targets=1)[1]
batch = np.ascontiguousarray(np.stack([imgs[k % 8] for k in range(n)]))
d_img = torch.from_numpy(batch).cuda()
d_az = torch.from_numpy(np.ascontiguousarray(np.tile(az, (n, 1)))).cuda()
ref = None
for flags in flag_list:
    p = frontend.default_params()
    p.flags = flags
    g = frontend.Frontend(400, 3360, params=p)
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    call = lambda: g.cartesian_batch_device(d_img.data_ptr(), n, batch.shape[1] * batch.shape[2], batch.shape[2], d_az.data_ptr(), 400,
                                            synth.RADAR_RESOLUTION, stream=s.cuda_stream)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    cart, blur = g.read_images(n - 1)
    same = "" if ref is None else f"  images identical to flags {flag_list[0]}: {np.array_equal(cart, ref[0]) and np.array_equal(blur, ref[1])}"
    if ref is None:
        ref = (cart, blur)
    px = n * g.W * g.W
    print(f"flags {flags}: {us:.1f} us per {n}-scan window ({px * 8 / us / 1e6:.2f} TB/s of image writes){same}")
    g.close()
