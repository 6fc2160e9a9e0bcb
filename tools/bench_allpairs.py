#!/usr/bin/env python3
"""BASELINE config 5: every keyframe of an N-keyframe DB queried against the whole DB (top-10, each
query limited to the keyframes at least 30 older than itself, like the streaming detector).

  one GPU :  python tools/bench_allpairs.py [N]                 (rsx_sc_query_self_device)
  N GPUs  :  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
                 tools/bench_allpairs.py [N]
             DB striped over the ranks, queries replicated, two-stage query with per-query
             eligibility limits (ShardedScanContext.query_device(q_elig_ptr=...)); `--staged` runs
             that same protocol on one GPU.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navtech_radar_slam_amd import scancontext as sc, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
staged = "--staged" in sys.argv
n = int(args[0]) if args else 100_000
k, excl = 10, 30
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
descs = synth.random_descriptors(777, n, binary=True)
rng = np.random.default_rng(1)
loops = rng.integers(n // 2, n, 200)                      # planted revisits: keyframe i repeats i - n/2 rotated
for i in loops:                                           # (sources lie in the first half, which stays untouched)
    descs[i] = synth.rotate_descriptor(descs[i - n // 2], int(rng.integers(0, 60)))
tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream); st = tstream.cuda_stream
t0 = time.perf_counter()
if world > 1 or staged:
    from navtech_radar_slam_amd.sharded import ShardedScanContext
    g = ShardedScanContext(capacity_hint=n // world + 1)
    dq = torch.from_numpy(descs).cuda()
    g.add_descriptors_device(dq.data_ptr(), n, st)
    lim = torch.clamp(torch.arange(n, dtype=torch.int64, device="cuda") - excl, min=0)

    def run(cnt):
        return g.query_device(dq.data_ptr(), cnt, k, stream=st, q_elig_ptr=lim.data_ptr(), elig_monotone=True)
else:
    g = sc.SCManager(capacity_hint=n)
    g.add_descriptors_f32(descs)
    out = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")

    def run(cnt):
        g.query_self_device(0, cnt, k, out.data_ptr(), exclude_recent=excl, stream=st)
        return out
torch.cuda.synchronize()
t_load = time.perf_counter() - t0
run(min(n, 4096))                                          # warm-up (workspaces, RCCL channels)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
res_d = run(n)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    res = res_d.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, k)
    ok = np.all(res["index"][loops, 0] == loops - n // 2)
    pairs = n * (n - excl) / 2
    print(f"N={n} gpus={world}{' (staged)' if staged else ''}: load {t_load:.2f} s; all-queries top-{k}: {dt*1e3:.1f} ms = "
          f"{n/dt:.0f} queries/s, {pairs/dt/1e9:.2f} G eligible pairs/s; planted loops found: {bool(ok)}")
if world > 1:
    dist.destroy_process_group()
