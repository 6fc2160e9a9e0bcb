"""Per-rank cost of the two-stage sharded query, emulated on ONE GPU: all `world` shard handles live on
this device; stage 1 of every shard, merge, then stage 2 of every shard.  Prints the time of one
rank's share (shard 0) -- what a rank of a real multi-GPU run computes between the collectives."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from navtech_radar_slam_amd import scancontext as sc

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n, k = 10000, 10
descs, queries, src, rot = bench.random_db_and_queries(n, nq)
tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream); st = tstream.cuda_stream
shards = [sc.SCManager(shard_rank=r, shard_world=world, capacity_hint=n // world + 8) for r in range(world)]
d_db = torch.from_numpy(descs).cuda()
for s in shards:
    s.add_descriptors_device(d_db.data_ptr(), n, stream=st)
dq = torch.from_numpy(queries).cuda()
parts = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
glob = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
finals = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
out = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
def run(timed):
    for r, s in enumerate(shards):
        if timed and r == 0: ev[0].record()
        s.query_stage1_device(dq.data_ptr(), nq, k, parts[r].data_ptr(), n_eligible=n - 30, stream=st)
        if timed and r == 0: ev[1].record()
    if timed: ev[2].record()
    shards[0].merge_device(parts.data_ptr(), world, nq, k, glob.data_ptr(), stream=st)
    if timed: ev[3].record()
    for r, s in enumerate(shards):
        if timed and r == 0: ev[4].record()
        s.query_stage2_device(nq, k, glob.data_ptr(), finals[r].data_ptr(), stream=st)
        if timed and r == 0: ev[5].record()
    shards[0].merge_device(finals.data_ptr(), world, nq, k, out.data_ptr(), stream=st)
for _ in range(2): run(False)
run(True); torch.cuda.synchronize()
res = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k)
ok = src < n - 30
print("planted ok:", bool(np.all(res["index"][ok, 0] == src[ok])))
print(f"world={world} nq={nq}: stage1 {ev[0].elapsed_time(ev[1]):.3f} ms, merge {ev[2].elapsed_time(ev[3]):.3f} ms, stage2 {ev[4].elapsed_time(ev[5]):.3f} ms")
