"""world_size-2/3 `gloo` tests of the sharded ScanContext query (N > 1 path) on CPU.

There is no GPU here, so each rank's LOCAL search is done by the oracle (allowed in tests); what
is under test is the product's distributed logic: block-cyclic ownership, the two-stage protocol
(stage-1 lists -> all-gather -> merge -> global bound tau -> stage 2 -> all-gather -> merge) with
16-byte records through torch.distributed, and librsx's host merge under the (dist, index) order.
The merged result must equal the unsharded oracle bit for bit on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShard:
    """Local backend for one rank: keeps keyframes i with i % world == rank, searches them with
    the oracle and reports GLOBAL indices (what the GPU SCManager does with shard_rank/world)."""

    def __init__(self, oracle, rank, world):
        self.o, self.rank, self.world = oracle, rank, world
        self.m = oracle.Manager()
        self.n_global = 0

    def add_descriptors_f32(self, descs):
        descs = np.asarray(descs, dtype=np.float32).reshape(-1, 1200)
        for d in descs:
            if self.n_global % self.world == self.rank:
                self.m.add_descriptor(d.astype(np.float64))
            self.n_global += 1

    def _search(self, q, k, n_eligible, subset_mod=None, tau=None, q_elig=None):
        """Exact local top-k; subset_mod: only local slots s with s % 3 == 0 (a stage-1 stand-in);
        tau (per query): drop hits a global bound already excludes (what stage 2 may skip);
        q_elig (per query): additional eligibility limit on the global index."""
        if n_eligible < 0:
            n_eligible = self.n_global
        out = np.zeros((q.shape[0], k), dtype=self.o.HIT_DTYPE)
        for i in range(q.shape[0]):
            lim = min(n_eligible, self.n_global) if q_elig is None else min(n_eligible, self.n_global, int(q_elig[i]))
            n_local_elig = len(range(self.rank, lim, self.world))
            r = self.m.exhaustive(q[i].astype(np.float64), n_eligible=n_local_elig, k=max(k, n_local_elig))
            keep = r["dist"] < 1e7
            if subset_mod is not None:
                keep &= (r["index"] % subset_mod) == 0
            if tau is not None:
                keep &= r["dist"] <= tau[i]
            r = r[keep][:k]
            r["index"] = r["index"] * self.world + self.rank
            out[i]["dist"] = 1e7
            out[i][:len(r)] = r
        return out

    def query_stage1(self, q, k, n_eligible, q_elig=None):
        self._q, self._k, self._ne, self._qe = q, k, n_eligible, q_elig
        return self._search(q, k, n_eligible, subset_mod=3, q_elig=q_elig)

    def query_stage2(self, global_topk):
        kth = np.where(global_topk["dist"][:, -1] < 1e7, global_topk["dist"][:, -1], np.inf)
        return self._search(self._q, self._k, self._ne, tau=kth, q_elig=self._qe)


def _worker(rank, world, port, ret, query_groups=1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import pyoracle as po
    from navtech_radar_slam_amd import sharded, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, nq, k = 301, 6, 10
        descs = synth.random_descriptors(1234, n, binary=True)
        queries = np.stack([synth.rotate_descriptor(descs[17 * i + 3], 5 * i) for i in range(nq)])
        queries[-1][:] = 0                                     # a query with no effective column
        sc = sharded.ShardedScanContext(local_backend=lambda sr, sw: OracleShard(po, sr, sw), query_groups=query_groups)
        assert (sc.n_qgroups, sc.shard_world) == (query_groups, world // query_groups) and sc.layout == f"{query_groups}x{world // query_groups}"
        sc.add_descriptors_f32(descs[:200])
        sc.add_descriptors_f32(descs[200:])                    # growing DB keeps the residue classes
        assert sc.backend.n_global == n and len(sc.backend.m) == len(range(sc.shard_rank, n, sc.shard_world))
        full = po.Manager()
        full.add_descriptors(descs.astype(np.float64))
        for n_elig in (-1, n - 30, 7, 1, 0):
            got = sc.query(queries, k=k, n_eligible=n_elig)
            for i in range(nq):
                want = full.exhaustive(queries[i].astype(np.float64), n_eligible=n if n_elig < 0 else n_elig, k=k)
                assert np.array_equal(got[i], want), (rank, n_elig, i, got[i], want)
        # every keyframe against the keyframes at least 30 older than itself (BASELINE configs 4/5 over shards)
        sub = np.arange(40, n, 23)
        lim = np.maximum(sub - 30, 0)
        got = sc.query(descs[sub], k=3, q_elig=lim)
        for j, i in enumerate(sub):
            want = full.exhaustive(descs[i].astype(np.float64), n_eligible=int(lim[j]), k=3)
            assert np.array_equal(got[j], want), (rank, i, got[j], want)
        n_sub = len(sc._subgroups)
        assert n_sub == (query_groups + world // query_groups if 1 < query_groups < world else 0)
        sc.close()                                             # collective: destroys the sub-communicators it created
        assert sc._subgroups == []
        dist.barrier()                                         # the default group is still alive
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_query_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {r: "ok" for r in range(world)}


@pytest.mark.parametrize("world,query_groups", [(4, 2), (4, 4), (2, 2)])
def test_two_dimensional_layout_gloo(world, query_groups):
    """query groups x DB shards (4 = 2 x 2, and the pure query-parallel layouts): the batch is cut into slices, each slice
    runs the two-stage protocol inside its group's shards, the slices are put together by an all-gather over the ranks
    with the same shard index; 6 queries over 4 groups leaves a group with an EMPTY slice (it still takes part in the
    collectives).  Results identical to the unsharded oracle on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, query_groups)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {r: "ok" for r in range(world)}


def test_auto_layout():
    from navtech_radar_slam_amd.sharded import auto_layout
    assert auto_layout(8, 8192) == 8 and auto_layout(8, 2048) == 4 and auto_layout(8, 1) == 1 and auto_layout(1, 8192) == 1
    assert auto_layout(4, 1500) == 2 and auto_layout(6, 8192) == 6


class OracleReplica:
    """Local backend of the filter-shard layout (sharded.FilterShardedScanContext): the whole DB on every rank, searched by
    the oracle.  Its "filter" returns 0.9 x the exact distance of every pair (a valid lower bound that identifies the pair);
    the scoring side checks that the bounds it was handed are exactly those -- i.e. that the all-to-all delivered the
    right rows of the right column blocks -- and prunes with them."""

    def __init__(self, oracle):
        self.o = oracle
        self.m = oracle.Manager()

    def __len__(self):
        return len(self.m)

    def add_descriptors_f32(self, descs):
        self.m.add_descriptors(np.asarray(descs, dtype=np.float32).reshape(-1, 1200).astype(np.float64))

    def _dists(self, qi, n):
        r = self.m.exhaustive(qi.astype(np.float64), n_eligible=n, k=max(n, 1))
        d = np.full(n, np.inf)
        ok = r["dist"] < 1e7
        d[r["index"][ok]] = r["dist"][ok]
        return d

    def filter_range(self, q, first, n):
        self.calls = getattr(self, "calls", 0) + 1
        out = np.empty((q.shape[0], n), dtype=np.float32)
        for i in range(q.shape[0]):
            out[i] = (0.9 * self._dists(q[i], first + n)[first:]).astype(np.float32)
        return out

    def query_bounds(self, q, k, n_eligible, lb):
        n_e = len(self) if n_eligible < 0 else min(n_eligible, len(self))
        assert lb.shape == (q.shape[0], n_e), (lb.shape, q.shape, n_e)
        out = np.zeros((q.shape[0], k), dtype=self.o.HIT_DTYPE)
        for i in range(q.shape[0]):
            d = self._dists(q[i], n_e)
            assert np.array_equal(lb[i], (0.9 * d).astype(np.float32)), "bounds of another query / another column block"
            out[i] = self.m.exhaustive(q[i].astype(np.float64), n_eligible=n_e, k=k)
            kth = out[i]["dist"][-1]
            assert not np.any((lb[i] > kth) & np.isin(np.arange(n_e), out[i]["index"][out[i]["dist"] < 1e7]))
        return out


def _worker_filter_shards(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import pyoracle as po
    from navtech_radar_slam_amd import sharded, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, nq, k = 171, 7, 4
        descs = synth.random_descriptors(77, n, binary=True)
        queries = np.stack([synth.rotate_descriptor(descs[13 * i + 3], 7 * i) for i in range(nq)])
        queries[-1][:] = 0
        sc = sharded.FilterShardedScanContext(local_backend=OracleReplica(po))
        assert sc.layout == f"{world}f"
        sc.add_descriptors_f32(descs[:100])
        sc.add_descriptors_f32(descs[100:])
        assert len(sc.backend) == n                                   # every rank keeps everything
        ld_r, rng = sc.ranges(n)
        assert ld_r % 32 == 0 and sum(c for _, c in rng) == n and all(f % 32 == 0 for f, _ in rng)
        full = po.Manager()
        full.add_descriptors(descs.astype(np.float64))
        for n_elig in (-1, n - 30, 33, 1, 0):                         # 33 entries: the last ranks' ranges are empty; 0: nothing eligible
            for qs in (queries, queries[:1]):                         # 1 query over 2-3 ranks: ranks with an empty slice
                got = sc.query(qs, k=k, n_eligible=n_elig)
                for i in range(len(qs)):
                    want = full.exhaustive(qs[i].astype(np.float64), n_eligible=n if n_elig < 0 else n_elig, k=k)
                    assert np.array_equal(got[i], want), (rank, n_elig, i, got[i], want)
        # a database smaller than the ranks' tiles (the live, growing DB): trailing ranks have an empty range that must not
        # name slots beyond the entries, and still take part in the exchange
        for n_small in (0, 5, 40, 100):
            small = sharded.FilterShardedScanContext(local_backend=OracleReplica(po))
            if n_small:
                small.add_descriptors_f32(descs[:n_small])
            ld_s, rng_s = small.ranges(n_small)
            assert all(f + c <= n_small and f % 32 == 0 for f, c in rng_s) and sum(c for _, c in rng_s) == n_small, rng_s
            ref = po.Manager()
            if n_small:
                ref.add_descriptors(descs[:n_small].astype(np.float64))
            got = small.query(queries, k=k)
            for i in range(nq):
                assert np.array_equal(got[i], ref.exhaustive(queries[i].astype(np.float64), n_eligible=n_small, k=k)), (rank, n_small, i)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_filter_shard_layout_gloo(world):
    """Replicated DB, the filter cut over the ranks by slot range, ONE all-to-all of bound-matrix row slices, every rank
    scores its slice of the batch, all-gather (sharded.FilterShardedScanContext): identical to the unsharded oracle on every
    rank, including eligibility limits that leave ranks without entries and batches that leave ranks without queries."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_filter_shards, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {r: "ok" for r in range(world)}
