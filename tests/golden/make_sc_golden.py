"""Generates tests/golden/sc_golden.npz from the CPU oracle (oracle/sc_ref.c).

The reference has no golden vectors and cannot be compiled here, so these fixtures
pin the oracle restatement itself (regression + cross-CPU determinism) and give the GPU parity
tests a committed known-answer set that does not need /root/reference at run time.
Run:  python tests/golden/make_sc_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from navtech_radar_slam_amd import synth  # noqa: E402

N = 160
clouds_a, _ = synth.keyframe_clouds(1234, N // 2, binary_z=True, loop_frac=0.15, min_gap=35, n_points=600)
clouds_b, _ = synth.keyframe_clouds(4321, N // 2, binary_z=False, loop_frac=0.15, min_gap=35, n_points=600)
clouds = clouds_a + clouds_b
m = po.Manager(dist_thres=0.45)
out = {"n_clouds": np.int32(len(clouds))}
desc, det = [], []
for i, c in enumerate(clouds):
    out[f"cloud_{i}"] = c[:, :3].copy()
    m.add_points(c[:, :3])
    desc.append(m.descriptor(i).astype(np.float32))
    det.append(m.detect_loop_closure())
out["desc"] = np.stack(desc)
out["det_loop_id"] = np.array([d[0] for d in det], dtype=np.int32)
out["det_yaw"] = np.array([d[1] for d in det], dtype=np.float32)
out["det_min_dist"] = np.array([d[2] for d in det], dtype=np.float64)
out["det_nn_idx"] = np.array([d[3] for d in det], dtype=np.int32)
q = np.array([40, 79, 120, 159], dtype=np.int32)
pd, ps = zip(*(m.pair_distances(m.descriptor(int(i))) for i in q))
out["pair_query"] = q
out["pair_dist"] = np.stack(pd)
out["pair_shift"] = np.stack(ps)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sc_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes; loops found:", int((out["det_loop_id"] >= 0).sum()))
