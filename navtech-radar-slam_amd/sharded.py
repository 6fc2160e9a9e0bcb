"""ScanContext database sharded over the ranks of a torch.distributed process group.

Keyframe i lives on rank i % world (block-cyclic, so a growing DB stays balanced).  A query runs
on every rank against its shard (HIP kernels through librsx.so), the per-rank top-k lists
(16-byte rsx_sc_hit records) are exchanged with ONE all-gather -- RCCL over xGMI when the group's
backend is "nccl" -- and merged under the total order (dist, global index), which reproduces the
sequential lowest-index-wins scan of the reference exactly.  The message is tiny (nq * k * 16 B per
rank): the exchange is latency-bound, so queries are batched.

`local_backend` is a seam for the CPU (gloo) tests, which have no GPU: anything with
add_descriptors_f32(descs) and query(q, k, n_eligible) -> (nq, k) HIT_DTYPE records.  The default
is the GPU SCManager; there is no CPU fallback in the product path.
"""
import numpy as np

from . import scancontext
from ._rsx import HIT_DTYPE


class ShardedScanContext:
    def __init__(self, group=None, device=None, local_backend=None, capacity_hint=1024):
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
                                                 capacity_hint=capacity_hint)
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

    def query_device(self, q_ptr, nq, k, n_eligible=-1, stream=0):
        """GPU path: device query pointer in, device tensor (nq, k, 2) f64 = rsx_sc_hit records out."""
        local = self._buf("local", (nq, k, 2))
        self.backend.query_device(q_ptr, nq, k, local.data_ptr(), n_eligible=n_eligible, stream=stream)
        if self.world == 1:
            return local
        parts = self._buf("parts", (self.world, nq, k, 2))
        self._dist.all_gather_into_tensor(parts.view(-1), local.view(-1), group=self.group)
        out = self._buf("out", (nq, k, 2))
        self.backend.merge_device(parts.data_ptr(), self.world, nq, k, out.data_ptr(), stream=stream)
        return out

    def query(self, q_descs, k=1, n_eligible=-1):
        """Host-array convenience form -> (nq, k) HIT_DTYPE, identical on every rank."""
        torch = self._torch
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        nq = q.shape[0]
        if self.on_gpu:
            dq = torch.from_numpy(q).to(self.device)
            s = torch.cuda.current_stream().cuda_stream
            out = self.query_device(dq.data_ptr(), nq, k, n_eligible, stream=s)
            torch.cuda.synchronize()
            return out.cpu().numpy().view(HIT_DTYPE).reshape(nq, k)
        local = np.ascontiguousarray(self.backend.query(q, k, n_eligible), dtype=HIT_DTYPE)
        if self.world == 1:
            return local
        lt = torch.from_numpy(local.view(np.float64).reshape(-1).copy())
        parts = torch.zeros(self.world * lt.numel(), dtype=torch.float64)
        self._dist.all_gather_into_tensor(parts, lt, group=self.group)
        return scancontext.merge_topk(parts.numpy().view(HIT_DTYPE).reshape(self.world, nq, k))
