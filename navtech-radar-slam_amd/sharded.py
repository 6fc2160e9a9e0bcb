"""ScanContext database sharded over the ranks of a torch.distributed process group.

Layout.  `world` ranks = `query_groups` (Q) x `shard_world` (S).  Rank r belongs to query group r // S and holds DB
shard r % S: keyframe i lives on the ranks with shard index i % S (block-cyclic, so a growing DB stays balanced), once
per query group.  A batch of nq queries is cut into Q contiguous slices, one per query group; inside a group the
slice runs against the group's S shards in two stages with one all-gather each -- RCCL over xGMI when the backend is
"nccl":
  stage 1  MFMA filter over the shard + exact scores of the shard's share of the lowest-bound entries; all-gather of
           the per-rank top-k lists (16-byte rsx_sc_hit records), merge: the k-th distance of the merged list is a
           GLOBAL upper bound tau of the final k-th best;
  stage 2  every shard scores only what its filter bounds still admit under tau (instead of under its own, much
           looser, local k-th best); all-gather + merge of the final per-rank lists;
and the Q slices are put together by one more all-gather over the ranks with the same shard index.  Q = 1 is the pure
DB-shard layout of rounds 1-2; S = 1 is pure query parallelism (the DB replicated, no exchange but the final one).
Why 2-D: per-rank work is  a * nq / Q  (per-query costs: query images, short-list selection, one re-scoring workgroup
per query)  +  b * nq * N / (Q * S)  (per-pair costs: the filter); the second term only depends on Q * S = world, the
first shrinks with Q alone -- with S = world a 10 k-keyframe DB scales ~4x on 8 GPUs, with Q > 1 the per-query part
scales too.  `auto_layout` picks Q from the batch size (a group needs enough queries to fill its GPU).
The merge is under the total order (dist, global index), which reproduces the sequential lowest-index-wins scan of the
reference exactly.  Messages are tiny (nq * k * 16 B per rank): the exchanges are latency-bound, so queries are batched.

`local_backend` is a seam for the CPU (gloo) tests, which have no GPU: an object -- or a callable (shard_rank,
shard_world) -> object -- with add_descriptors_f32(descs), query_stage1(q, k, n_eligible[, q_elig]) and
query_stage2(global_topk), both -> (nq, k) HIT_DTYPE records.  The default is the GPU SCManager; there is no CPU
fallback in the product path.
"""
import numpy as np

from . import scancontext
from ._rsx import HIT_DTYPE


def auto_layout(world, nq, min_queries_per_group=512):
    """-> query_groups Q (a divisor of world): as many query groups as the batch can feed with >= min_queries_per_group
    queries each (below that a GPU's 256 CUs are no longer filled by its slice and DB shards are the better use of the
    ranks: a single query, Q = 1, is the pure DB-shard layout).  The DB is tiny next to 288 GB of HBM (0.83 GB per
    100 000 keyframes), so replication per query group is never a capacity question."""
    q = 1
    for cand in range(1, world + 1):
        if world % cand == 0 and nq >= cand * min_queries_per_group:
            q = cand
    return q


def _host_staged(dist, group, on_gpu):
    """GPU ranks on a backend that cannot move device memory (gloo: several ranks on ONE device in the tests, or a box
    without RCCL): the exchanges are staged through host memory"""
    return bool(on_gpu and dist.is_initialized() and dist.get_backend(group) == "gloo")


def _all_gather(dist, out, inp, group, staged):
    if not staged:
        dist.all_gather_into_tensor(out, inp, group=group)
        return
    h = inp.new_empty(out.shape, device="cpu")
    dist.all_gather_into_tensor(h, inp.cpu(), group=group)
    out.copy_(h)


def _all_to_all(dist, out, inp, group, staged):
    if not staged:
        dist.all_to_all_single(out, inp, group=group)
        return
    h = inp.new_empty(out.shape, device="cpu")
    dist.all_to_all_single(h, inp.cpu(), group=group)
    out.copy_(h)


class _ExchangeTimer:
    """Opt-in (bench.py): device time between the start and the end of every collective on the stream it runs on, the wait
    for the slowest peer included (two events per exchange; off by default)."""
    time_exchanges = False

    def _timed_exchange(self, fn):
        if not self.time_exchanges or not getattr(self, "on_gpu", False):
            return fn()
        torch = self._torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.__dict__.setdefault("_ex_events", []).append((e0, e1))
        return out

    def exchange_ms(self):
        """sum over the exchanges recorded since the last call (synchronises)"""
        ev = self.__dict__.pop("_ex_events", [])
        if ev:
            self._torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in ev))


class ShardedScanContext(_ExchangeTimer):
    def __init__(self, group=None, device=None, local_backend=None, capacity_hint=1024, filter_mode=0, query_groups=1):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        q = int(query_groups)
        if q < 1 or self.world % q:
            raise ValueError(f"query_groups {q} does not divide the world size {self.world}")
        self.n_qgroups, self.shard_world = q, self.world // q
        self.qgroup, self.shard_rank = self.rank // self.shard_world, self.rank % self.shard_world
        self.shard_group, self.col_group = group, None
        self._subgroups = []  # the communicators this instance created (close() destroys them)
        if q > 1 and self.shard_world == 1:
            self.shard_group, self.col_group = None, group     # pure query parallelism: the only exchange is over everybody
        elif q > 1:
            # torch.distributed: EVERY rank of the default group has to call new_group for every subgroup, in the same
            # order -- ranks outside `group` would never get here and the members would wait for them forever
            if group is not None and dist.get_world_size(group) != dist.get_world_size():
                raise ValueError("query_groups > 1 with DB shards needs the default (WORLD) group: torch.distributed.new_group "
                                 "is collective over all ranks")
            ranks = list(range(self.world)) if group is None else dist.get_process_group_ranks(group)
            s_w = self.shard_world
            for g in range(q):
                grp = dist.new_group(ranks=[ranks[g * s_w + s] for s in range(s_w)]) if s_w > 1 else None
                self._subgroups.append(grp)
                if g == self.qgroup:
                    self.shard_group = grp
            for s in range(s_w):
                grp = dist.new_group(ranks=[ranks[g * s_w + s] for g in range(q)])
                self._subgroups.append(grp)
                if s == self.shard_rank:
                    self.col_group = grp
        self.on_gpu = local_backend is None
        if self.on_gpu:
            dev = torch.cuda.current_device() if device is None else device
            self.backend = scancontext.SCManager(device=dev, shard_rank=self.shard_rank, shard_world=self.shard_world,
                                                 capacity_hint=capacity_hint, filter_mode=filter_mode)
            self.device = torch.device("cuda", dev)
        else:
            self.backend = local_backend(self.shard_rank, self.shard_world) if callable(local_backend) else local_backend
            self.device = torch.device("cpu")
        self._staged = _host_staged(dist, group, self.on_gpu)
        self._bufs = {}

    @property
    def layout(self):
        return f"{self.n_qgroups}x{self.shard_world}"

    def owner(self, index):
        """shard index of keyframe `index` (the ranks qgroup * shard_world + owner hold it)"""
        return index % self.shard_world

    def add_descriptors_f32(self, descs):
        """Every rank passes every new keyframe (same order); each keeps its own residue class."""
        self.backend.add_descriptors_f32(descs)

    def add_descriptors_device(self, ptr, n, stream=0):
        self.backend.add_descriptors_device(ptr, n, stream)

    def _buf(self, name, shape):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._torch.zeros(shape, dtype=self._torch.float64, device=self.device)
            self._bufs[name] = t
        return t

    def _slice(self, nq):
        """this query group's slice [lo, hi) of a batch of nq queries, and the (padded) slice length"""
        chunk = -(-nq // self.n_qgroups)
        lo = min(nq, self.qgroup * chunk)
        return lo, min(nq, lo + chunk), chunk

    def _gather_merge(self, local, name, nq, k, stream):
        parts = self._buf(name + "_parts", (self.shard_world, nq, k, 2))
        self._timed_exchange(lambda: _all_gather(self._dist, parts.view(-1), local.view(-1), self.shard_group, self._staged))
        out = self._buf(name + "_merged", (nq, k, 2))
        self.backend.merge_device(parts.data_ptr(), self.shard_world, nq, k, out.data_ptr(), stream=stream)
        return out

    def _query_slice_device(self, q_ptr, nq, k, n_eligible, stream, q_elig_ptr, elig_monotone):
        """nq queries of this group against the group's shards -> device tensor (>= nq, k, 2); nq may be 0 (a group
        without queries still takes part in the exchanges of its shard group)."""
        pad = max(nq, 1)
        local = self._buf("local", (pad, k, 2))
        if self.shard_world == 1:
            if nq == 0:
                return local
            if not q_elig_ptr:
                self.backend.query_device(q_ptr, nq, k, local.data_ptr(), n_eligible=n_eligible, stream=stream)
                return local
            final = self._buf("final", (pad, k, 2))
            self.backend.query_stage1_device(q_ptr, nq, k, local.data_ptr(), n_eligible=n_eligible, stream=stream,
                                             q_elig_ptr=q_elig_ptr, elig_monotone=elig_monotone)
            self.backend.query_stage2_device(nq, k, local.data_ptr(), final.data_ptr(), stream=stream)
            return final
        # q_elig_ptr: device int64[nq], query i only sees global indices < q_elig[i] (kept alive by the caller)
        if nq:
            self.backend.query_stage1_device(q_ptr, nq, k, local.data_ptr(), n_eligible=n_eligible, stream=stream,
                                             q_elig_ptr=q_elig_ptr, elig_monotone=elig_monotone)
        bound = self._gather_merge(local, "s1", pad, k, stream)
        final = self._buf("final", (pad, k, 2))
        if nq:
            self.backend.query_stage2_device(nq, k, bound.data_ptr(), final.data_ptr(), stream=stream)
        return self._gather_merge(final, "s2", pad, k, stream)

    def query_device(self, q_ptr, nq, k, n_eligible=-1, stream=0, q_elig_ptr=0, elig_monotone=False):
        """GPU path: device query pointer in, device tensor (nq, k, 2) f64 = rsx_sc_hit records out (the whole batch,
        identical on every rank).  `stream` must be the (non-default) torch stream current on this device: the library
        launches on it and the collectives run on it too.  stream=0 would make librsx use the handle's private stream,
        unordered with torch's -- then everything runs on an own side stream instead."""
        if stream == 0:
            torch = self._torch
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                out = self.query_device(q_ptr, nq, k, n_eligible, stream=self._side.cuda_stream, q_elig_ptr=q_elig_ptr,
                                        elig_monotone=elig_monotone)
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            return out
        if self.n_qgroups == 1:
            return self._query_slice_device(q_ptr, nq, k, n_eligible, stream, q_elig_ptr, elig_monotone)[:nq]
        lo, hi, chunk = self._slice(nq)
        mine = self._buf("slice", (chunk, k, 2))
        res = self._query_slice_device(q_ptr + lo * 4800, hi - lo, k, n_eligible, stream, q_elig_ptr + lo * 8 if q_elig_ptr else 0,
                                       elig_monotone)
        if hi > lo:
            mine[:hi - lo].copy_(res[:hi - lo])
        whole = self._buf("whole", (self.n_qgroups * chunk, k, 2))
        self._timed_exchange(lambda: _all_gather(self._dist, whole.view(-1), mine.view(-1), self.col_group, self._staged))
        return whole[:nq]

    def _gather_merge_host(self, local, nq, k):
        torch = self._torch
        lt = torch.from_numpy(np.ascontiguousarray(local, dtype=HIT_DTYPE).view(np.float64).reshape(-1).copy())
        parts = torch.zeros(self.shard_world * lt.numel(), dtype=torch.float64)
        self._dist.all_gather_into_tensor(parts, lt, group=self.shard_group)
        return scancontext.merge_topk(parts.numpy().view(HIT_DTYPE).reshape(self.shard_world, nq, k))

    def _query_slice_host(self, q, k, n_eligible, q_elig):
        """local_backend path for this group's slice (possibly empty: the collectives still run)"""
        nq = q.shape[0]
        pad = max(nq, 1)
        empty = np.zeros((pad, k), dtype=HIT_DTYPE)
        if nq:
            part = self.backend.query_stage1(q, k, n_eligible, q_elig) if q_elig is not None else self.backend.query_stage1(q, k, n_eligible)
            part = np.ascontiguousarray(part, dtype=HIT_DTYPE)
        else:
            part = empty
        if self.shard_world == 1:
            return np.ascontiguousarray(self.backend.query_stage2(part), dtype=HIT_DTYPE) if nq else empty[:0]
        bound = self._gather_merge_host(part, pad, k)
        final = np.ascontiguousarray(self.backend.query_stage2(bound), dtype=HIT_DTYPE) if nq else empty
        return self._gather_merge_host(final, pad, k)[:nq]

    def query(self, q_descs, k=1, n_eligible=-1, q_elig=None):
        """Host-array convenience form -> (nq, k) HIT_DTYPE, identical on every rank.
        q_elig (optional, int64[nq]): query i only sees global indices < min(n_eligible, q_elig[i])."""
        torch = self._torch
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        nq = q.shape[0]
        if q_elig is not None:
            q_elig = np.ascontiguousarray(q_elig, dtype=np.int64).reshape(nq)
        if self.on_gpu:
            dq = torch.from_numpy(q).to(self.device)
            s = torch.cuda.current_stream().cuda_stream
            de = torch.from_numpy(q_elig).to(self.device) if q_elig is not None else None
            mono = bool(q_elig is not None and np.all(np.diff(q_elig) >= 0))
            out = self.query_device(dq.data_ptr(), nq, k, n_eligible, stream=s, q_elig_ptr=de.data_ptr() if de is not None else 0,
                                    elig_monotone=mono)
            torch.cuda.synchronize()
            return out.cpu().numpy().view(HIT_DTYPE).reshape(nq, k)
        if self.n_qgroups == 1:
            return self._query_slice_host(q, k, n_eligible, q_elig)
        lo, hi, chunk = self._slice(nq)
        mine = np.zeros((chunk, k), dtype=HIT_DTYPE)
        mine[:hi - lo] = self._query_slice_host(q[lo:hi], k, n_eligible, q_elig[lo:hi] if q_elig is not None else None)
        lt = torch.from_numpy(mine.view(np.float64).reshape(-1).copy())
        whole = torch.zeros(self.n_qgroups * lt.numel(), dtype=torch.float64)
        self._dist.all_gather_into_tensor(whole, lt, group=self.col_group)
        return whole.numpy().view(HIT_DTYPE).reshape(self.n_qgroups * chunk, k)[:nq].copy()

    def close(self):
        """frees the shard and the sub-communicators this instance created (collective: every rank closes)"""
        if hasattr(self.backend, "close"):
            self.backend.close()
        for g in self._subgroups:
            if g is not None and g != self._dist.GroupMember.NON_GROUP_MEMBER:
                self._dist.destroy_process_group(g)
        self._subgroups = []


class FilterShardedScanContext(_ExchangeTimer):
    """Filter shards over a REPLICATED database (rsx.h: rsx_sc_filter_range_device / rsx_sc_query_bounds_device).

    Every rank holds every keyframe -- the DB is small next to HBM (0.83 GB per 100 000 keyframes) -- and a batch costs
      per pair   the lower-bound filter: rank r runs it for ALL nq queries against slots [first_r, first_r + n_r), 1 / world
                 of the DB (ranges are cut at multiples of 32 slots, the tile of the filter images);
      per query  short list, window previews, exact re-scoring: rank t runs them for ITS nq / world queries against the
                 whole DB, with the bounds the other ranks computed.
    Between the two sits ONE all-to-all (RCCL over xGMI): rank t receives rows [q_t, q_t + nq_t) of every rank's bound
    matrix (fp16), nq * N * 2 / world bytes per rank (20 MB for 8192 queries x 10 000 keyframes on 8 ranks); a last all-gather puts
    the slices together (nq * k * 16 B).  Unlike the Q x S layouts there is no second stage and no replicated per-query work
    except the query images of the filter (every rank needs all of them), so both cost terms shrink with the world size.
    The records are those of one GPU: a pair's bound does not depend on who computed it.

    `local_backend` (CPU tests, no GPU): an object with __len__, add_descriptors_f32(descs),
    filter_range(q, first, n) -> (nq, n) float32 bounds, query_bounds(q, k, n_eligible, lb) -> (nq, k) HIT_DTYPE."""

    def __init__(self, group=None, device=None, local_backend=None, capacity_hint=1024, filter_mode=0):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.on_gpu = local_backend is None
        if self.on_gpu:
            dev = torch.cuda.current_device() if device is None else device
            self.backend = scancontext.SCManager(device=dev, capacity_hint=capacity_hint, filter_mode=filter_mode)
            self.device = torch.device("cuda", dev)
        else:
            self.backend = local_backend
            self.device = torch.device("cpu")
        self._staged = _host_staged(dist, group, self.on_gpu)
        self._bufs = {}
        # the layout bookkeeping of ShardedScanContext, for callers that treat both alike (bench.py)
        self.n_qgroups, self.shard_world, self.qgroup, self.shard_rank = self.world, 1, self.rank, 0

    @property
    def layout(self):
        return f"{self.world}f"

    def add_descriptors_f32(self, descs):
        """Every rank passes every new keyframe (same order) and keeps all of them."""
        self.backend.add_descriptors_f32(descs)

    def add_descriptors_device(self, ptr, n, stream=0):
        self.backend.add_descriptors_device(ptr, n, stream)

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    def _slice(self, nq):
        chunk = -(-nq // self.world)
        lo = min(nq, self.rank * chunk)
        return lo, min(nq, lo + chunk), chunk

    def ranges(self, n_entries):
        """-> (columns per rank (a multiple of 32), [(first slot, slots)] per rank) for a filter over entries [0, n_entries)"""
        tiles = -(-max(n_entries, 1) // 32)
        ld_r = -(-tiles // self.world) * 32
        out = []
        for r in range(self.world):
            first = min(n_entries, r * ld_r)
            cnt = min(n_entries, first + ld_r) - first
            # an empty range starts at the last tile boundary inside the database (a rank beyond the entries -- a small or
            # growing DB on many ranks -- must not name slots that do not exist; its column block is never read)
            out.append((r * ld_r if cnt else n_entries // 32 * 32, cnt))
        return ld_r, out

    def _n_entries(self, n_eligible):
        n = len(self.backend)
        return n if n_eligible < 0 or n_eligible > n else n_eligible

    def query_device(self, q_ptr, nq, k, n_eligible=-1, stream=0, q_elig_ptr=0, elig_monotone=False):
        """device query pointer in, device tensor (nq, k, 2) f64 = rsx_sc_hit records out, identical on every rank.
        `stream`: the torch stream current on this device (see ShardedScanContext.query_device)."""
        torch = self._torch
        if q_elig_ptr:
            raise NotImplementedError("per-query eligibility limits are not carried through the filter-shard layout")
        if stream == 0:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                out = self.query_device(q_ptr, nq, k, n_eligible, stream=self._side.cuda_stream)
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            return out
        lo, hi, chunk = self._slice(nq)
        ld_r, rng = self.ranges(self._n_entries(n_eligible))
        first, n = rng[self.rank]
        mine = self._buf("mine", (chunk, k, 2), torch.float64)
        if self.world == 1:
            send = self._buf("send", (chunk, ld_r), torch.float16)
            if n:
                self.backend.filter_range_device(q_ptr, nq, first, n, send.data_ptr(), ld_r, stream=stream)
            self.backend.query_bounds_device(q_ptr, nq, k, mine.data_ptr(), send.data_ptr(), 1, ld_r, chunk * ld_r,
                                             n_eligible=n_eligible, stream=stream)
            return mine[:nq]
        send = self._buf("send", (self.world * chunk, ld_r), torch.float16)
        recv = self._buf("recv", (self.world, chunk, ld_r), torch.float16)
        if n:  # (a rank without entries still takes part in the exchange: the others are already waiting in it)
            self.backend.filter_range_device(q_ptr, nq, first, n, send.data_ptr(), ld_r, stream=stream)
        self._timed_exchange(lambda: _all_to_all(self._dist, recv.view(-1).view(torch.uint8), send.view(-1).view(torch.uint8), self.group,
                                                 self._staged))  # fp16 as bytes (gloo has no half)
        if hi > lo:
            self.backend.query_bounds_device(q_ptr + lo * 4800, hi - lo, k, mine.data_ptr(), recv.data_ptr(), self.world, ld_r,
                                             chunk * ld_r, n_eligible=n_eligible, stream=stream)
        whole = self._buf("whole", (self.world * chunk, k, 2), torch.float64)
        self._timed_exchange(lambda: _all_gather(self._dist, whole.view(-1), mine.view(-1), self.group, self._staged))
        return whole[:nq]

    def query(self, q_descs, k=1, n_eligible=-1):
        """Host-array convenience form -> (nq, k) HIT_DTYPE, identical on every rank."""
        torch = self._torch
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        nq = q.shape[0]
        if self.on_gpu:
            dq = torch.from_numpy(q).to(self.device)
            out = self.query_device(dq.data_ptr(), nq, k, n_eligible, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            return out.cpu().numpy().view(HIT_DTYPE).reshape(nq, k)
        lo, hi, chunk = self._slice(nq)
        n_e = self._n_entries(n_eligible)
        ld_r, rng = self.ranges(n_e)
        first, n = rng[self.rank]
        send = np.zeros((self.world * chunk, ld_r), dtype=np.float32)
        if n:
            send[:nq, :n] = self.backend.filter_range(q, first, n)
        recv = torch.zeros(self.world * chunk * ld_r, dtype=torch.float32)
        if self.world > 1:
            self._dist.all_to_all_single(recv, torch.from_numpy(send.reshape(-1)), group=self.group)
        else:
            recv = torch.from_numpy(send.reshape(-1))
        blocks = recv.numpy().reshape(self.world, chunk, ld_r)
        mine = np.zeros((chunk, k), dtype=HIT_DTYPE)
        if hi > lo:
            lb = np.concatenate([blocks[r, :hi - lo] for r in range(self.world)], axis=1)[:, :n_e]
            mine[:hi - lo] = self.backend.query_bounds(q[lo:hi], k, n_eligible, lb)
        lt = torch.from_numpy(mine.view(np.float64).reshape(-1).copy())
        whole = torch.zeros(self.world * lt.numel(), dtype=torch.float64)
        if self.world > 1:
            self._dist.all_gather_into_tensor(whole, lt, group=self.group)
        else:
            whole = lt
        return whole.numpy().view(HIT_DTYPE).reshape(self.world * chunk, k)[:nq].copy()

    def close(self):
        if hasattr(self.backend, "close"):
            self.backend.close()
