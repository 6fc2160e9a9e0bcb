#!/usr/bin/env python3
"""Time the exact-all path (filter off) on bench.py's trajectory workload: 8192 queries x 10 000 keyframes, top-10."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import scancontext, synth, _rsx
n_db, nq, k = 10000, 8192, 10
db_pts, db_off, q_pts, q_off, _ = synth.trajectory_keyframes(1234, n_db, 4321, nq, binary_z=True)
b = scancontext.SCManager(capacity_hint=n_db + nq + 8)
for i in range(n_db):
    b.makeAndSaveScancontextAndKeys(db_pts[db_off[i]:db_off[i + 1]])
for i in range(nq):
    b.makeAndSaveScancontextAndKeys(q_pts[q_off[i]:q_off[i + 1]])
allv = b.export_descriptors_f32(0, n_db + nq)
b.close()
res = {}
for mode, name in ((_rsx.FILTER_OFF, "exact_all"), (_rsx.FILTER_AUTO, "filtered")):
    g = scancontext.SCManager(filter_mode=mode, capacity_hint=n_db)
    g.add_descriptors_f32(allv[:n_db])
    d_q = torch.from_numpy(np.ascontiguousarray(allv[n_db:])).cuda()
    out = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    g.query_device(d_q.data_ptr(), nq, k, out.data_ptr(), n_eligible=n_db - 30, stream=st)
    torch.cuda.synchronize()
    reps = 3 if mode == _rsx.FILTER_OFF else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        g.query_device(d_q.data_ptr(), nq, k, out.data_ptr(), n_eligible=n_db - 30, stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res[name] = out.cpu().numpy().copy()
    print(f"{name}: {dt*1e3:.2f} ms per step, {nq/dt:.0f} queries/s")
    g.close()
print("identical:", np.array_equal(res["exact_all"], res["filtered"]))
