"""The multi-GPU layouts of the ScanContext query on real hardware.

  * the filter-shard layout (rsx_sc_filter_range_device / rsx_sc_query_bounds_device) emulated rank by rank on one GPU:
    the records of one unsharded call, bit for bit;
  * TWO PROCESSES on device 0, exchanging over gloo through host memory (RCCL refuses two ranks on one GPU): the first
    multi-process execution of rsx_sc_query_stage{1,2}_device + rsx_sc_merge_topk_device (DB shards) and of the
    filter-shard entries, against the unsharded handle."""
import os
import socket
import sys

import numpy as np
import pytest

from navtech_radar_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workload(n=2003, nq=67):
    descs = synth.random_descriptors(21, n, binary=True)
    descs[11].reshape(60, 20)[20:40] = 0
    queries = np.stack([synth.rotate_descriptor(descs[(i * 17) % n], i % 60) for i in range(nq)])
    queries[5].reshape(60, 20)[:25] = 0          # empty columns: the filter's slow path
    queries[9][:] = 0                            # no effective column at all
    return descs, queries


@pytest.mark.parametrize("n_db", [100, 1025])
def test_filter_shards_of_a_small_database_on_8_ranks(n_db):
    """100 or 1025 keyframes on 8 ranks (round-4 advisor finding): the trailing ranks' ranges are empty and start INSIDE the
    database -- rsx_sc_filter_range_device used to refuse them with RSX_ERR_RANGE while the other ranks were already waiting
    in the all-to-all -- and an empty range is a no-op whatever its first slot is"""
    import torch
    from navtech_radar_slam_amd import scancontext as sc, sharded
    descs, queries = _workload(n=n_db, nq=12)
    g = sc.SCManager(filter_mode=2)
    g.add_descriptors_f32(descs)
    lay = sharded.FilterShardedScanContext.__new__(sharded.FilterShardedScanContext)
    lay.world = 8
    ld_r, rng = sharded.FilterShardedScanContext.ranges(lay, n_db)
    assert sum(c for _, c in rng) == n_db and all(f % 32 == 0 and f + c <= n_db for f, c in rng), rng
    assert any(c == 0 for _, c in rng)
    dq = torch.from_numpy(queries).cuda()
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = side.cuda_stream
    sends = torch.full((8, len(queries), ld_r), float("nan"), dtype=torch.float16, device="cuda")
    for r, (first, cnt) in enumerate(rng):
        g.filter_range_device(dq.data_ptr(), len(queries), first, cnt, sends[r].data_ptr(), ld_r, stream=s)
    g.filter_range_device(dq.data_ptr(), len(queries), 7 * ld_r, 0, sends[7].data_ptr(), ld_r, stream=s)   # the old, unclamped start
    got = torch.zeros((len(queries), 5, 2), dtype=torch.float64, device="cuda")
    g.query_bounds_device(dq.data_ptr(), len(queries), 5, got.data_ptr(), sends.data_ptr(), 8, ld_r, len(queries) * ld_r, stream=s)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    assert np.array_equal(got.cpu().numpy().view(sc.HIT_DTYPE).reshape(len(queries), 5), g.query(queries, k=5))
    g.close()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("n_elig", [-1, 1973, 40, 0])
def test_filter_shards_emulated_on_one_gpu(world, n_elig):
    """what the ranks of sharded.FilterShardedScanContext execute, one after the other on this GPU: rank r's range filter for
    all queries into column block r, then every rank's slice through rsx_sc_query_bounds_device; 40 eligible entries leave
    most ranks with an empty range."""
    import torch
    from navtech_radar_slam_amd import scancontext as sc, sharded
    descs, queries = _workload()
    n, nq, k = len(descs), len(queries), 10
    g = sc.SCManager()
    g.add_descriptors_f32(descs)
    want = g.query(queries, k=k, n_eligible=n_elig)
    lay = sharded.FilterShardedScanContext.__new__(sharded.FilterShardedScanContext)
    lay.world = world
    n_e = n if n_elig < 0 else n_elig
    ld_r, rng = sharded.FilterShardedScanContext.ranges(lay, n_e)
    chunk = -(-nq // world)
    dq = torch.from_numpy(queries).cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()          # a non-default stream: librsx takes stream 0 as "the handle's own stream", which
    torch.cuda.set_stream(side)         # is not ordered with torch's copies below
    s = side.cuda_stream
    # recv[t][r] = what rank t holds after the all-to-all: rows of its slice, one column block per sender r
    sends = torch.full((world, world * chunk, ld_r), float("nan"), dtype=torch.float16, device="cuda")
    for r, (first, cnt) in enumerate(rng):
        g.filter_range_device(dq.data_ptr(), nq, first, cnt, sends[r].data_ptr(), ld_r, stream=s)
    got = torch.zeros((world * chunk, k, 2), dtype=torch.float64, device="cuda")
    for t in range(world):
        lo, hi = min(nq, t * chunk), min(nq, (t + 1) * chunk)
        if hi == lo:
            continue
        recv = sends[:, t * chunk:(t + 1) * chunk, :].contiguous()
        g.query_bounds_device(dq.data_ptr() + lo * 4800, hi - lo, k, got[lo:].data_ptr(), recv.data_ptr(), world, ld_r,
                              chunk * ld_r, n_eligible=n_elig, stream=s)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    assert np.array_equal(got[:nq].cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k), want)
    # and the bounds themselves are the single-GPU filter's, whoever computed them
    lb = g.filter_bounds(queries)
    mine = torch.cat([sends[r, :nq, :c] for r, (_, c) in enumerate(rng)], dim=1).float().cpu().numpy()
    assert np.array_equal(mine, lb[:, :n_e], equal_nan=True)
    g.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from navtech_radar_slam_amd import scancontext as sc, sharded
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        descs, queries = _workload()
        n, nq, k = len(descs), len(queries), 10
        full = sc.SCManager(device=0)
        full.add_descriptors_f32(descs)
        layouts = [sharded.ShardedScanContext(device=0, query_groups=1), sharded.ShardedScanContext(device=0, query_groups=world),
                   sharded.FilterShardedScanContext(device=0)]
        for lay in layouts:
            assert lay.on_gpu and lay._staged
            lay.add_descriptors_f32(descs[:1500])
            lay.add_descriptors_f32(descs[1500:])
            for n_elig in (-1, n - 30, 40, 0):
                want = full.query(queries, k=k, n_eligible=n_elig)
                got = lay.query(queries, k=k, n_eligible=n_elig)
                assert np.array_equal(got, want), (rank, lay.layout, n_elig)
            got1 = lay.query(queries[:1], k=k)                         # one query: ranks with an empty slice
            assert np.array_equal(got1, full.query(queries[:1], k=k)), (rank, lay.layout, "one query")
        assert layouts[0].backend.local_size == len(range(rank, n, world)) and len(layouts[2].backend) == n
        for lay in layouts:
            lay.close()
        full.close()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_processes_on_one_gpu():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {r: "ok" for r in range(world)}
