"""Generates tests/golden/sc_golden.npz from the REFERENCE'S OWN Scancontext.cpp.

oracle/_ref/libref_sc_sse2.so is /root/reference/pgo/SC-A-LOAM/include/scancontext/Scancontext.cpp
compiled unmodified (oracle/ref_sc.cpp, oracle/standin/) in the summation order of the reference's
build (SSE2).  Descriptors, detector results (loop id, yaw) and pair distances / shifts below are what
that code returns.  The two values the reference only prints (min_dist, nn_idx of its log line,
Scancontext.cpp:406,412) come from the oracle and are cross-checked here against the parsed log line.
/root/reference does not exist on the GPU box, so the vectors are committed.
Run (in the build container):  python tests/golden/make_sc_golden.py
"""
import os
import re
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
ref = po.RefManager(po.ORDER_EIGEN_SSE2, dist_thres=0.45)
refsc = po.RefSC(po.ORDER_EIGEN_SSE2)
po.set_sum_order(po.ORDER_EIGEN_SSE2)
m = po.Manager(dist_thres=0.45)
out = {"n_clouds": np.int32(len(clouds)), "generator": np.array("reference Scancontext.cpp, " + refsc.build_info())}
desc, det = [], []
for i, c in enumerate(clouds):
    out[f"cloud_{i}"] = c[:, :3].copy()
    ref.add_points(c[:, :3])
    m.add_points(c[:, :3])
    d, rk, sk = ref.get(i)
    assert np.array_equal(d, d.astype(np.float32).astype(np.float64))
    desc.append(d.astype(np.float32))
    lid, yaw = ref.detect_loop_closure()
    o_lid, o_yaw, o_md, o_nn = m.detect_loop_closure()
    assert (lid, yaw) == (o_lid, o_yaw), (i, lid, yaw, o_lid, o_yaw)
    log = ref.last_log()
    if log:  # "[Loop found] Nearest distance: 0.123 btn 45 and 7."
        mm = re.search(r"Nearest distance: (\S+) btn (\d+) and (\d+)\.", log)
        assert int(mm.group(2)) == i and int(mm.group(3)) == o_nn, (log, o_nn)
        assert abs(float(mm.group(1)) - o_md) <= 5e-3 * max(1.0, abs(o_md)), (log, o_md)   # printed with 3-6 digits
    det.append((lid, yaw, o_md, o_nn))
out["desc"] = np.stack(desc)
out["det_loop_id"] = np.array([d[0] for d in det], dtype=np.int32)
out["det_yaw"] = np.array([d[1] for d in det], dtype=np.float32)
out["det_min_dist"] = np.array([d[2] for d in det], dtype=np.float64)
out["det_nn_idx"] = np.array([d[3] for d in det], dtype=np.int32)
q = np.array([40, 79, 120, 159], dtype=np.int32)
all_desc = np.stack([ref.get(i)[0] for i in range(N)])
pd, ps = zip(*(refsc.distances(all_desc[int(i)], all_desc) for i in q))
out["pair_query"] = q
out["pair_dist"] = np.stack(pd)
out["pair_shift"] = np.stack(ps)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sc_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes; loops found:", int((out["det_loop_id"] >= 0).sum()))
