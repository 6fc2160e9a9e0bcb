#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: every keyframe of an N-keyframe DB queried against the whole DB
(rsx_sc_query_self_device, top-10, each query limited to the keyframes at least 30 older than itself
like the streaming detector).  Usage: tools/bench_allpairs.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import scancontext as sc, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
k = 10
descs = synth.random_descriptors(777, n, binary=True)
rng = np.random.default_rng(1)
loops = rng.integers(n // 2, n, 200)                      # planted revisits: keyframe i repeats i - n/2 rotated
for i in loops:                                           # (sources lie in the first half, which stays untouched)
    descs[i] = synth.rotate_descriptor(descs[i - n // 2], int(rng.integers(0, 60)))
g = sc.SCManager(capacity_hint=n)
t0 = time.perf_counter()
g.add_descriptors_f32(descs)
t_load = time.perf_counter() - t0
tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream); st = tstream.cuda_stream
out = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
g.query_self_device(0, 4096, k, out.data_ptr(), exclude_recent=30, stream=st)   # warm-up (workspaces)
torch.cuda.synchronize()
t0 = time.perf_counter()
g.query_self_device(0, n, k, out.data_ptr(), exclude_recent=30, stream=st)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
res = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, k)
ok = np.all(res["index"][loops, 0] == loops - n // 2)
pairs = n * (n - 30) / 2
print(f"N={n}: load {t_load:.2f} s; all-queries top-{k}: {dt*1e3:.1f} ms = {n/dt:.0f} queries/s, {pairs/dt/1e9:.2f} G eligible pairs/s; planted loops found: {bool(ok)}")
