"""Stress the two-stage sharded query on one GPU and print any mismatch in detail."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from navtech_radar_slam_amd import scancontext as sc, synth
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from test_gpu_sc_filter import make_db

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, nq, k = 6007, 48, 10
descs = make_db(31, n, binary=True)
rng = np.random.default_rng(5)
queries = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
queries[::3, rng.integers(0, 1200, 40)] = 0
queries[2] = 0
full = sc.SCManager(filter_mode=1)
full.add_descriptors_f32(descs)
want = full.query(queries, k=k, n_eligible=n - 30)
shards = [sc.SCManager(shard_rank=r, shard_world=world, filter_mode=2) for r in range(world)]
for s in shards:
    s.add_descriptors_f32(descs)
tstream = torch.cuda.Stream()
torch.cuda.set_stream(tstream)
st = tstream.cuda_stream
dq = torch.from_numpy(queries).cuda()
bad = 0
for rep in range(reps):
    parts = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
    glob = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        s.query_stage1_device(dq.data_ptr(), nq, k, parts[r].data_ptr(), n_eligible=n - 30, stream=st)
    shards[0].merge_device(parts.data_ptr(), world, nq, k, glob.data_ptr(), stream=st)
    finals = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        s.query_stage2_device(nq, k, glob.data_ptr(), finals[r].data_ptr(), stream=st)
    out = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
    shards[0].merge_device(finals.data_ptr(), world, nq, k, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k)
    if not np.array_equal(got, want):
        bad += 1
        p1 = parts.cpu().numpy().view(sc.HIT_DTYPE).reshape(world, nq, k)
        g1 = glob.cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k)
        f1 = finals.cpu().numpy().view(sc.HIT_DTYPE).reshape(world, nq, k)
        for qi in range(nq):
            if not np.array_equal(got[qi], want[qi]):
                print(f"rep {rep} query {qi}\n got  {got[qi]}\n want {want[qi]}\n stage1 glob {g1[qi]}")
                for r in range(world):
                    print(f"  shard {r} stage1 {p1[r, qi]}\n  shard {r} final  {f1[r, qi]}")
                break
print("mismatching reps:", bad, "of", reps)
