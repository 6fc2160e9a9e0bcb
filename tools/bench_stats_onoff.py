"""Does the re-scoring kernel's bookkeeping (profile_enable: a handful of global atomics per query) cost time?
ms per step of the headline query shape with the counters off / on.  Usage: python tools/bench_stats_onoff.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from navtech_radar_slam_amd import scancontext, synth

n_db, nq, k = 10000, 8192, 10
descs = synth.random_descriptors(1, n_db, binary=True)
rng = np.random.default_rng(2)
q = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n_db))], int(rng.integers(0, 60))) for _ in range(nq)])
q[rng.uniform(size=q.shape) < 0.05] = 0
m = scancontext.SCManager(device=0, capacity_hint=n_db + 8)
m.add_descriptors_f32(descs)
d_q = torch.from_numpy(q).cuda()
st = torch.cuda.Stream()
out = torch.zeros((nq, k, 2), dtype=torch.float64, device='cuda')
for on in (False, True, False, True):
    m.profile_enable(on)
    for _ in range(3):
        m.query_device(d_q.data_ptr(), nq, k, out.data_ptr(), n_eligible=n_db - 30, stream=st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.query_device(d_q.data_ptr(), nq, k, out.data_ptr(), n_eligible=n_db - 30, stream=st.cuda_stream)
    torch.cuda.synchronize()
    print(f"counters {'on ' if on else 'off'}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per step")
    if on:
        m.profile_read_rescoring()
