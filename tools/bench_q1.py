#!/usr/bin/env python3
"""One query against N keyframes through the three paths (single-query launch sc_q1.hip / exact-all / batched filter chain):
us per call for back-to-back device-resident calls and per synchronous host call, records compared between the paths.
--sizes 1000,10000,100000 --k 1,10 --nq 1,8 --trajectory (the bench's 10k drive instead of random binary descriptors)"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import scancontext, synth, _rsx

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1000,10000,100000")
ap.add_argument("--k", default="1,10")
ap.add_argument("--nq", default="1,8")
ap.add_argument("--modes", default="q1,exact_all,filter")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--trajectory", action="store_true")
ap.add_argument("--out", default="")
args = ap.parse_args()
MODES = {"q1": _rsx.FILTER_Q1, "exact_all": _rsx.FILTER_OFF, "filter": _rsx.FILTER_FORCE}
st = torch.cuda.current_stream().cuda_stream
res = {}
for n in [int(x) for x in args.sizes.split(",")]:
    if args.trajectory:
        db_pts, db_off, q_pts, q_off, _ = synth.trajectory_keyframes(1234, n, 4321, 64, binary_z=True)
        b = scancontext.SCManager(capacity_hint=n + 72)
        for i in range(n):
            b.makeAndSaveScancontextAndKeys(db_pts[db_off[i]:db_off[i + 1]])
        for i in range(64):
            b.makeAndSaveScancontextAndKeys(q_pts[q_off[i]:q_off[i + 1]])
        allv = b.export_descriptors_f32(0, n + 64)
        b.close()
        descs, qs = allv[:n], np.ascontiguousarray(allv[n:])
    else:
        descs = synth.random_descriptors(77, n, binary=True)
        qs = synth.random_descriptors(78, 64, binary=True)
        qs[::2] = np.stack([synth.rotate_descriptor(descs[(i * 131) % (n - 40)], (7 * i) % 60) for i in range(32)])
    d_q = torch.from_numpy(qs).cuda()
    n_elig = n - 30
    hs = {}
    for name in args.modes.split(","):
        hs[name] = scancontext.SCManager(capacity_hint=n + 8, filter_mode=MODES[name])
        hs[name].add_descriptors_f32(descs)
    for k in [int(x) for x in args.k.split(",")]:
        for nq in [int(x) for x in args.nq.split(",")]:
            row = {}
            outs = {}
            for name, h in hs.items():
                reps = args.reps if name != "exact_all" or n <= 10000 else max(10, args.reps // 10)
                out = torch.zeros((64, k, 2), dtype=torch.float64, device="cuda")
                for i in range(8):
                    h.query_device(d_q[(i * nq) % max(1, 64 - nq):].data_ptr(), nq, k, out.data_ptr(), n_eligible=n_elig, stream=st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(reps):
                    q0 = (i * nq) % (64 - nq + 1)
                    h.query_device(d_q[q0:].data_ptr(), nq, k, out[q0:].data_ptr(), n_eligible=n_elig, stream=st)
                torch.cuda.synchronize()
                dev_us = (time.perf_counter() - t0) / reps * 1e6
                t0 = time.perf_counter()
                hr = max(10, reps // 4)
                for i in range(hr):
                    h.query(qs[:nq], k=k, n_eligible=n_elig)
                host_us = (time.perf_counter() - t0) / hr * 1e6
                full = torch.zeros((64, k, 2), dtype=torch.float64, device="cuda")
                for q0 in range(0, 64, nq):
                    h.query_device(d_q[q0:].data_ptr(), min(nq, 64 - q0), k, full[q0:].data_ptr(), n_eligible=n_elig, stream=st)
                torch.cuda.synchronize()
                outs[name] = full.cpu().numpy()
                row[name] = {"us_per_call_stream": round(dev_us, 2), "us_per_call_host": round(host_us, 2),
                             "kernel": h.profiled_kernel_name()}
                if name == "q1":
                    row[name]["alg_TBps"] = round(nq * n_elig * 4800 / dev_us / 1e6, 3)
                    row[name]["read_TBps"] = round(nq * n_elig * 2672 / dev_us / 1e6, 3)
                    h.profile_enable(True)
                    for q0 in range(0, 64, nq):
                        h.query_device(d_q[q0:].data_ptr(), min(nq, 64 - q0), k, full[q0:].data_ptr(), n_eligible=n_elig, stream=st)
                    torch.cuda.synchronize()
                    r3 = h.profile_read_rescoring()
                    row[name]["stats"] = r3
                    h.profile_enable(False)
            names = list(outs)
            row["identical"] = all(np.array_equal(outs[names[0]], outs[m]) for m in names[1:])
            res[f"n{n}_k{k}_nq{nq}"] = row
            print(f"n{n}_k{k}_nq{nq}", json.dumps(row), flush=True)
    for h in hs.values():
        h.close()
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
