"""Host-side mirror of the reference `SCManager` (Scancontext.h:62-122) on top of librsx.so.

Method names and argument meaning follow the reference so parity tests read like reference
code: makeAndSaveScancontextAndKeys / detectLoopClosureID / saveScancontextAndKeys /
detectLoopClosureIDBetweenSession / getConstRefRecentSCD / setSCdistThres.  Everything is
computed by the HIP kernels behind the C-ABI; this class only marshals buffers.
"""
import ctypes as C
import os

import numpy as np

from . import _rsx
from ._rsx import HIT_DTYPE, MODE_CANDIDATE, MODE_EXHAUSTIVE, check, lib

WINDOW_P = 320            # rsx.h RSX_SC_WINDOW_P
WINDOW_MARGIN = 1.25e-3   # rsx.h RSX_SC_WINDOW_MARGIN


class SCManager:
    # hyper parameters (Scancontext.h:83-104); fixed at construction like the reference's consts
    LIDAR_HEIGHT = 2.0
    PC_NUM_RING = 20
    PC_NUM_SECTOR = 60
    PC_MAX_RADIUS = 80.0
    NUM_EXCLUDE_RECENT = 30
    NUM_CANDIDATES_FROM_TREE = 3
    SEARCH_RATIO = 0.1
    TREE_MAKING_PERIOD_ = 30

    def __init__(self, device=0, shard_rank=0, shard_world=1, capacity_hint=1024, sc_dist_thres=0.2,
                 lidar_height=None, num_exclude_recent=None, num_candidates=None, tree_making_period=None,
                 filter_mode=_rsx.FILTER_AUTO, filter_kind=_rsx.KIND_AUTO, sum_order=_rsx.SUM_EIGEN_SSE2):
        L = lib()
        p = _rsx.ScParams()
        check(L.rsx_sc_default_params(C.byref(p)))
        p.device, p.shard_rank, p.shard_world = device, shard_rank, shard_world
        p.capacity_hint = capacity_hint
        p.dist_thres = sc_dist_thres
        p.filter_mode = filter_mode
        p.filter_kind = filter_kind or _rsx.default_filter_kind
        p.sum_order = sum_order
        if lidar_height is not None:
            p.lidar_height = lidar_height
        if num_exclude_recent is not None:
            p.num_exclude_recent = num_exclude_recent
        if num_candidates is not None:
            p.num_candidates = num_candidates
        if tree_making_period is not None:
            p.tree_making_period = tree_making_period
        self.NUM_EXCLUDE_RECENT = p.num_exclude_recent
        self.NUM_CANDIDATES_FROM_TREE = p.num_candidates
        self.SC_DIST_THRES = sc_dist_thres
        self._h = C.c_void_p()
        self._L = L
        check(L.rsx_sc_create(C.byref(p), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_sc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference user API -------------------------------------------------------------
    def makeAndSaveScancontextAndKeys(self, scan_down):
        """scan_down: (n, >=3) float32 points x,y,z[,intensity...] (pcl::PointCloud<PointXYZI>)."""
        pts = np.ascontiguousarray(scan_down, dtype=np.float32)
        idx = C.c_int32()
        check(self._L.rsx_sc_add_points(self._h, pts.ctypes.data, pts.shape[0], pts.shape[1] * 4, C.byref(idx)))
        return idx.value

    def makeAndSaveScancontextAndKeysDownsampled(self, scan, voxelgrid):
        """downSizeFilterScancontext.filter + makeAndSaveScancontextAndKeys (PGO.cpp:482-492) in one
        call: the downsampled cloud stays on the GPU.  voxelgrid: navtech_radar_slam_amd.voxelgrid.VoxelGrid."""
        pts = np.ascontiguousarray(scan, dtype=np.float32)
        idx = C.c_int32()
        check(self._L.rsx_sc_add_points_downsampled(self._h, voxelgrid._h, pts.ctypes.data, pts.shape[0], pts.shape[1] * 4,
                                                    voxelgrid.leaf, C.byref(idx)))
        return idx.value

    def saveScancontextAndKeys(self, scd):
        """scd: 20x60 descriptor, column-major double (1200,) like Eigen::MatrixXd."""
        d = np.ascontiguousarray(scd, dtype=np.float64).reshape(-1)
        idx = C.c_int32()
        check(self._L.rsx_sc_add_descriptor(self._h, d.ctypes.data, C.byref(idx)))
        return idx.value

    def saveScancontextAndKeysRounded(self, scd):
        """The same for arbitrary doubles (e.g. SCDs re-read from text files): rounds to fp32, returns
        (index, largest absolute rounding error)."""
        d = np.ascontiguousarray(scd, dtype=np.float64).reshape(-1)
        idx, err = C.c_int32(), C.c_double()
        check(self._L.rsx_sc_add_descriptor_rounded(self._h, d.ctypes.data, C.byref(idx), C.byref(err)))
        return idx.value, err.value

    def detectLoopClosureID(self, mode=MODE_CANDIDATE, full=False):
        """-> (loop_id, yaw_diff_rad) like the reference; full=True adds (min_dist, nn_idx)."""
        lid, yaw, md, nn = C.c_int32(), C.c_float(), C.c_double(), C.c_int32()
        check(self._L.rsx_sc_detect_loop_closure(self._h, mode, C.byref(lid), C.byref(yaw), C.byref(md), C.byref(nn)))
        if full:
            return lid.value, yaw.value, md.value, nn.value
        return lid.value, yaw.value

    def detectLoopClosureIDBetweenSession(self, curr_key, curr_desc, full=False):
        key = np.ascontiguousarray(curr_key, dtype=np.float32)
        d = np.ascontiguousarray(curr_desc, dtype=np.float64).reshape(-1)
        lid, yaw, md, nn = C.c_int32(), C.c_float(), C.c_double(), C.c_int32()
        check(self._L.rsx_sc_detect_between_session(self._h, key.ctypes.data, d.ctypes.data, C.byref(lid),
                                                    C.byref(yaw), C.byref(md), C.byref(nn)))
        if full:
            return lid.value, yaw.value, md.value, nn.value
        return lid.value, yaw.value

    # ---- the reference's public helpers (Scancontext.h:60-66), stateless ------------------------
    def makeScancontext(self, scan_down):
        pts = np.ascontiguousarray(scan_down, dtype=np.float32)
        out = np.empty(1200, dtype=np.float64)
        check(self._L.rsx_sc_make_scancontext(self._h, pts.ctypes.data, pts.shape[0], pts.shape[1] * 4, out.ctypes.data))
        return out

    def makeRingkeyFromScancontext(self, desc):
        d = np.ascontiguousarray(desc, dtype=np.float64).reshape(-1)
        out = np.empty(20, dtype=np.float64)
        check(self._L.rsx_sc_make_keys(self._h, d.ctypes.data, out.ctypes.data, None))
        return out

    def makeSectorkeyFromScancontext(self, desc):
        d = np.ascontiguousarray(desc, dtype=np.float64).reshape(-1)
        out = np.empty(60, dtype=np.float64)
        check(self._L.rsx_sc_make_keys(self._h, d.ctypes.data, None, out.ctypes.data))
        return out

    def distDirectSC(self, sc1, sc2):
        a = np.ascontiguousarray(sc1, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(sc2, dtype=np.float64).reshape(-1)
        d = C.c_double()
        check(self._L.rsx_sc_dist_direct(self._h, a.ctypes.data, b.ctypes.data, C.byref(d)))
        return d.value

    def fastAlignUsingVkey(self, vkey1, vkey2):
        a = np.ascontiguousarray(vkey1, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(vkey2, dtype=np.float64).reshape(-1)
        k = C.c_int32()
        check(self._L.rsx_sc_fast_align(self._h, a.ctypes.data, b.ctypes.data, C.byref(k)))
        return k.value

    def distanceBtnScanContext(self, sc1, sc2):
        a = np.ascontiguousarray(sc1, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(sc2, dtype=np.float64).reshape(-1)
        d, k = C.c_double(), C.c_int32()
        check(self._L.rsx_sc_distance(self._h, a.ctypes.data, b.ctypes.data, C.byref(d), C.byref(k)))
        return d.value, k.value

    def detect_ex(self, mode=MODE_CANDIDATE):
        """-> _rsx.ScDetection (everything of the reference's log line, one lock)."""
        r = _rsx.ScDetection()
        check(self._L.rsx_sc_detect_loop_closure_ex(self._h, mode, C.byref(r)))
        return r

    def getConstRefRecentSCD(self):
        return self.descriptor(len(self) - 1)

    def setSCdistThres(self, new_thres):
        self.SC_DIST_THRES = new_thres
        check(self._L.rsx_sc_set_dist_thres(self._h, float(new_thres)))

    # ---- data access (public members polarcontexts_ etc., Scancontext.h:110-115) ---------
    def __len__(self):
        n = C.c_int64()
        check(self._L.rsx_sc_size(self._h, C.byref(n)))
        return n.value

    @property
    def local_size(self):
        n = C.c_int64()
        check(self._L.rsx_sc_local_size(self._h, C.byref(n)))
        return n.value

    @property
    def tree_size(self):
        n = C.c_int64()
        check(self._L.rsx_sc_tree_size(self._h, C.byref(n)))
        return n.value

    def descriptor(self, i):
        out = np.empty(1200, dtype=np.float64)
        check(self._L.rsx_sc_get_descriptor(self._h, i, out.ctypes.data))
        return out

    def ringkey(self, i):
        out = np.empty(20, dtype=np.float32)
        check(self._L.rsx_sc_get_ringkey(self._h, i, out.ctypes.data))
        return out

    def sectorkey(self, i):
        out = np.empty(60, dtype=np.float64)
        check(self._L.rsx_sc_get_sectorkey(self._h, i, out.ctypes.data))
        return out

    # ---- batched / exhaustive extensions ------------------------------------
    def add_descriptors_f32(self, descs):
        d = np.ascontiguousarray(descs, dtype=np.float32).reshape(-1, 1200)
        check(self._L.rsx_sc_add_descriptors_f32(self._h, d.ctypes.data, d.shape[0]))

    def add_descriptors_device(self, dev_ptr, n, stream=0):
        check(self._L.rsx_sc_add_descriptors_f32_device(self._h, dev_ptr, n, stream))

    def export_descriptors_f32(self, first_slot=0, count=None):
        """f32 sector-major descriptors of local slots [first_slot, first_slot + count) -> (count, 1200)."""
        if count is None:
            count = self.local_size - first_slot
        out = np.empty((count, 1200), dtype=np.float32)
        check(self._L.rsx_sc_export_descriptors_f32(self._h, first_slot, count, out.ctypes.data))
        return out

    def save(self, path):
        check(self._L.rsx_sc_save(self._h, os.fsencode(path)))

    def load(self, path):
        n = C.c_int64()
        check(self._L.rsx_sc_load(self._h, os.fsencode(path), C.byref(n)))
        return n.value

    def query(self, q_descs, k=1, n_eligible=-1, out=None):
        """out: optional (nq, k) HIT_DTYPE array to fill (e.g. over pinned memory, _rsx.PinnedArray)"""
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        if out is None:
            out = np.zeros((q.shape[0], k), dtype=HIT_DTYPE)
        elif out.dtype != HIT_DTYPE or out.shape != (q.shape[0], k) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (nq, k) array of HIT_DTYPE")
        check(self._L.rsx_sc_query(self._h, q.ctypes.data, q.shape[0], k, n_eligible, out.ctypes.data))
        return out

    def query_device(self, q_ptr, nq, k, out_ptr, n_eligible=-1, stream=0):
        check(self._L.rsx_sc_query_device(self._h, q_ptr, nq, k, n_eligible, out_ptr, stream))

    def query_stage1_device(self, q_ptr, nq, k, partial_ptr, n_eligible=-1, stream=0, q_elig_ptr=0, elig_monotone=False):
        """q_elig_ptr: optional device int64[nq] per-query eligibility limits (must outlive stage 2)."""
        if q_elig_ptr:
            check(self._L.rsx_sc_query_stage1_elig_device(self._h, q_ptr, nq, k, n_eligible, q_elig_ptr,
                                                          1 if elig_monotone else 0, partial_ptr, stream))
        else:
            check(self._L.rsx_sc_query_stage1_device(self._h, q_ptr, nq, k, n_eligible, partial_ptr, stream))

    def filter_range_device(self, q_ptr, nq, first_slot, n_slots, lb_ptr, ld, stream=0):
        """bounds of nq device queries against slots [first_slot, first_slot + n_slots) -> device float [nq][ld]"""
        check(self._L.rsx_sc_filter_range_device(self._h, q_ptr, nq, first_slot, n_slots, lb_ptr, ld, stream))

    def query_bounds_device(self, q_ptr, nq, k, out_ptr, lb_blocks_ptr, n_blocks, block_ld, block_stride, n_eligible=-1, stream=0):
        """rsx_sc_query_device with the bounds supplied as column blocks (what the filter shards of a replicated DB deliver)"""
        check(self._L.rsx_sc_query_bounds_device(self._h, q_ptr, nq, k, n_eligible, lb_blocks_ptr, n_blocks, block_ld,
                                                 block_stride, out_ptr, stream))

    def query_stage2_device(self, nq, k, global_ptr, out_ptr, stream=0):
        check(self._L.rsx_sc_query_stage2_device(self._h, nq, k, global_ptr, out_ptr, stream))

    def query_self_device(self, q_first, nq, k, out_ptr, n_eligible=-1, exclude_recent=-1, stream=0):
        check(self._L.rsx_sc_query_self_device(self._h, q_first, nq, k, n_eligible, exclude_recent, out_ptr, stream))

    def pair_distances(self, q_desc, first=0, count=None):
        q = np.ascontiguousarray(q_desc, dtype=np.float32).reshape(-1)
        if count is None:
            count = self.local_size - first
        dist = np.empty(count, dtype=np.float64)
        shift = np.empty(count, dtype=np.int32)
        check(self._L.rsx_sc_pair_distances(self._h, q.ctypes.data, first, count, dist.ctypes.data, shift.ctypes.data))
        return dist, shift

    def filter_bounds(self, q_descs):
        """MFMA filter lower bounds of dist(query, entry) for every local entry: (nq, n_local) fp32."""
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        out = np.empty((q.shape[0], self.local_size), dtype=np.float32)
        check(self._L.rsx_sc_filter_bounds(self._h, q.ctypes.data, q.shape[0], out.ctypes.data))
        return out

    def window_previews(self, q_descs, k=10):
        """The stage between the filter and the exact re-scoring (csrc/sc_window.hip), for diagnostics and tests:
        -> (slots, pv, kstar, shift_mask, counts): the first WINDOW_P short-list entries of every query (local slots, -1 past the
        end), their matrix-core preview of the pair distance and sector-key alignment (-1: not unique, the preview is a
        lower bound only; -2 with a NaN preview: no record, the entry's filter bound cannot reach the top-k); bit t of
        shift_mask: the window shift kstar - 3 + t can be the minimum (the exact evaluation skips the others)."""
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        nq = q.shape[0]
        slots = np.empty((nq, WINDOW_P), dtype=np.int32)
        pv = np.empty((nq, WINDOW_P), dtype=np.float32)
        ks = np.empty((nq, WINDOW_P), dtype=np.int32)
        sm = np.empty((nq, WINDOW_P), dtype=np.int32)
        cnt = np.empty(nq, dtype=np.int32)
        check(self._L.rsx_sc_window_previews(self._h, q.ctypes.data, nq, k, slots.ctypes.data, pv.ctypes.data, ks.ctypes.data,
                                             sm.ctypes.data, cnt.ctypes.data))
        return slots, pv, ks, sm, cnt

    @staticmethod
    def filter_eps():
        return lib().rsx_sc_filter_eps()

    def profiled_kernel_name(self):
        return self._L.rsx_sc_profiled_kernel_name(self._h).decode()

    def merge_device(self, parts_ptr, nparts, nq, k, out_ptr, stream=0):
        check(self._L.rsx_sc_merge_topk_device(self._h, parts_ptr, nparts, nq, k, out_ptr, stream))

    def profile_enable(self, on=True):
        check(self._L.rsx_sc_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """-> (launches, total_ms) of the dominant kernel since the last read (HIP events)."""
        n, ms = C.c_int64(), C.c_double()
        check(self._L.rsx_sc_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_read_rescoring(self):
        """-> (candidates, exact window evaluations, queries that scored any, candidates served by the window kernel,
        candidates that needed a per-wavefront alignment, window shifts evaluated exactly): rsx_sc_rescoring_stats."""
        from ._rsx import RescoringStats
        st = RescoringStats()
        st.struct_size = C.sizeof(RescoringStats)
        check(self._L.rsx_sc_profile_read_rescoring(self._h, C.byref(st)))
        return (st.candidates, st.exact_evals, st.queries_rescored, st.window_previews, st.valu_previews, st.exact_window_shifts)

    def hit_to_loop(self, hit):
        h = np.zeros(1, dtype=HIT_DTYPE)
        h[0] = hit
        lid, yaw = C.c_int32(), C.c_float()
        check(self._L.rsx_sc_hit_to_loop(self._h, h.ctypes.data, C.byref(lid), C.byref(yaw)))
        return lid.value, yaw.value


class ShardedSet:
    """rsx_scs_*: ONE process, the database over several GPUs: len(devices) = query_groups x DB shards (keyframe i on the
    shards i % S of every group); the exchanges of the two-stage query are peer copies or, with exchange="rccl",
    ncclAllGather.  What a single C++ host (alaserPGO) uses instead of torch.distributed."""

    def __init__(self, devices, sc_dist_thres=0.2, capacity_hint=1024, query_groups=1, exchange="peer"):
        L = lib()
        p = _rsx.ScParams()
        check(L.rsx_sc_default_params(C.byref(p)))
        p.dist_thres = sc_dist_thres
        p.capacity_hint = capacity_hint
        dev = (C.c_int32 * len(devices))(*devices)
        self._h = C.c_void_p()
        self._L = L
        check(L.rsx_scs_create_layout(C.byref(p), dev, len(devices), query_groups, {"peer": 0, "rccl": 1}[exchange], C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_scs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_shards(self):
        return self._L.rsx_scs_num_shards(self._h)

    @property
    def num_query_groups(self):
        return self._L.rsx_scs_num_query_groups(self._h)

    def __len__(self):
        n = C.c_int64()
        check(self._L.rsx_scs_size(self._h, C.byref(n)))
        return n.value

    def makeAndSaveScancontextAndKeys(self, scan_down):
        pts = np.ascontiguousarray(scan_down, dtype=np.float32)
        idx = C.c_int32()
        check(self._L.rsx_scs_add_points(self._h, pts.ctypes.data, pts.shape[0], pts.shape[1] * 4, C.byref(idx)))
        return idx.value

    def add_descriptors_f32(self, descs):
        d = np.ascontiguousarray(descs, dtype=np.float32).reshape(-1, 1200)
        check(self._L.rsx_scs_add_descriptors_f32(self._h, d.ctypes.data, d.shape[0]))

    def descriptor(self, i):
        out = np.empty(1200, dtype=np.float64)
        check(self._L.rsx_scs_get_descriptor(self._h, i, out.ctypes.data))
        return out

    def query(self, q_descs, k=1, n_eligible=-1):
        q = np.ascontiguousarray(q_descs, dtype=np.float32).reshape(-1, 1200)
        out = np.zeros((q.shape[0], k), dtype=HIT_DTYPE)
        check(self._L.rsx_scs_query(self._h, q.ctypes.data, q.shape[0], k, n_eligible, out.ctypes.data))
        return out

    def detect(self):
        r = _rsx.ScDetection()
        check(self._L.rsx_scs_detect_loop_closure(self._h, C.byref(r)))
        return r


def merge_topk(parts, k=None):
    """Host merge of per-shard lists: parts (nparts, nq, k) HIT_DTYPE -> (nq, k)."""
    parts = np.ascontiguousarray(parts, dtype=HIT_DTYPE)
    nparts, nq, kk = parts.shape
    out = np.zeros((nq, kk), dtype=HIT_DTYPE)
    check(lib().rsx_sc_merge_topk(parts.ctypes.data, nparts, nq, kk, out.ctypes.data))
    return out
