"""ScanContext database sharded over the ranks of a torch.distributed process group.

Keyframe i lives on rank i % world (block-cyclic, so a growing DB stays balanced).  A query runs
on every rank against its shard (HIP kernels through librsx.so) in two stages with one all-gather
each -- RCCL over xGMI when the group's backend is "nccl":
  stage 1  MFMA filter over the shard + exact scores of the shard's share of the lowest-bound
           entries; all-gather of the per-rank top-k lists (16-byte rsx_sc_hit records), merge:
           the k-th distance of the merged list is a GLOBAL upper bound tau of the final k-th best;
  stage 2  every shard scores only what its filter bounds still admit under tau (instead of under
           its own, much looser, local k-th best); all-gather + merge of the final per-rank lists.
The merge is under the total order (dist, global index), which reproduces the sequential
lowest-index-wins scan of the reference exactly.  Messages are tiny (nq * k * 16 B per rank): the
exchanges are latency-bound, so queries are batched.

`local_backend` is a seam for the CPU (gloo) tests, which have no GPU: anything with
add_descriptors_f32(descs), query_stage1(q, k, n_eligible[, q_elig]) and query_stage2(global_topk), both
-> (nq, k) HIT_DTYPE records.  The default is the GPU SCManager; there is no CPU fallback in the
product path.
"""
import numpy as np

from . import scancontext
from ._rsx import HIT_DTYPE


class ShardedScanContext:
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
            self.backend = scancontext.SCManager(device=dev, shard_rank=self.rank, shard_world=self.world,
                                                 capacity_hint=capacity_hint, filter_mode=filter_mode)
            self.device = torch.device("cuda", dev)
        else:
            self.backend = local_backend
            self.device = torch.device("cpu")
        self._bufs = {}

    def owner(self, index):
        return index % self.world

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

    def _gather_merge(self, local, name, nq, k, stream):
        parts = self._buf(name + "_parts", (self.world, nq, k, 2))
        self._dist.all_gather_into_tensor(parts.view(-1), local.view(-1), group=self.group)
        out = self._buf(name + "_merged", (nq, k, 2))
        self.backend.merge_device(parts.data_ptr(), self.world, nq, k, out.data_ptr(), stream=stream)
        return out

    def query_device(self, q_ptr, nq, k, n_eligible=-1, stream=0, q_elig_ptr=0, elig_monotone=False):
        """GPU path: device query pointer in, device tensor (nq, k, 2) f64 = rsx_sc_hit records out.
        `stream` must be the (non-default) torch stream current on this device: the library launches on
        it and the collectives run on it too.  stream=0 would make librsx use the handle's private
        stream, unordered with torch's -- then everything runs on an own side stream instead."""
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
        local = self._buf("local", (nq, k, 2))
        if self.world == 1 and not q_elig_ptr:
            self.backend.query_device(q_ptr, nq, k, local.data_ptr(), n_eligible=n_eligible, stream=stream)
            return local
        # q_elig_ptr: device int64[nq], query i only sees global indices < q_elig[i] (kept alive by the caller)
        self.backend.query_stage1_device(q_ptr, nq, k, local.data_ptr(), n_eligible=n_eligible, stream=stream,
                                         q_elig_ptr=q_elig_ptr, elig_monotone=elig_monotone)
        if self.world == 1:
            final = self._buf("final", (nq, k, 2))
            self.backend.query_stage2_device(nq, k, local.data_ptr(), final.data_ptr(), stream=stream)
            return final
        bound = self._gather_merge(local, "s1", nq, k, stream)
        final = self._buf("final", (nq, k, 2))
        self.backend.query_stage2_device(nq, k, bound.data_ptr(), final.data_ptr(), stream=stream)
        return self._gather_merge(final, "s2", nq, k, stream)

    def _gather_merge_host(self, local, nq, k):
        torch = self._torch
        lt = torch.from_numpy(np.ascontiguousarray(local, dtype=HIT_DTYPE).view(np.float64).reshape(-1).copy())
        parts = torch.zeros(self.world * lt.numel(), dtype=torch.float64)
        self._dist.all_gather_into_tensor(parts, lt, group=self.group)
        return scancontext.merge_topk(parts.numpy().view(HIT_DTYPE).reshape(self.world, nq, k))

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
        if q_elig is not None:
            part = np.ascontiguousarray(self.backend.query_stage1(q, k, n_eligible, q_elig), dtype=HIT_DTYPE)
        else:
            part = np.ascontiguousarray(self.backend.query_stage1(q, k, n_eligible), dtype=HIT_DTYPE)
        if self.world == 1:
            return np.ascontiguousarray(self.backend.query_stage2(part), dtype=HIT_DTYPE)
        bound = self._gather_merge_host(part, nq, k)
        final = self.backend.query_stage2(bound)
        return self._gather_merge_host(final, nq, k)
