"""ctypes loader for the CPU oracle (oracle/liboracle.so, oracle/_ref/libref_kdtree.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NR, NS, DS = 20, 60, 1200


class Hit(C.Structure):
    _fields_ = [("dist", C.c_double), ("index", C.c_int32), ("shift", C.c_int32)]


HIT_DTYPE = np.dtype([("dist", "<f8"), ("index", "<i4"), ("shift", "<i4")])


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference exists, _ref/libref_kdtree.so."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_sos = [os.path.join(_HERE, "_ref", f) for f in ("libref_kdtree.so", "libref_sc_seq.so", "libref_sc_sse2.so", "libref_sc_avxfma.so", "libref_sc_avxfma34.so")]
    if os.path.isdir("/root/reference/pgo/SC-A-LOAM/include/scancontext"):
        ref_srcs = [os.path.join(_HERE, "ref_kdtree.cpp"), os.path.join(_HERE, "ref_sc.cpp"), os.path.join(_HERE, "standin", "Eigen", "Dense")]
        stale_ref = any((not os.path.exists(so)) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in ref_srcs) for so in ref_sos)
        if force or stale_ref:
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.scref_xy2theta.restype = C.c_float
        L.scref_xy2theta.argtypes = [C.c_float, C.c_float]
        L.scref_deg2rad_f.restype = C.c_float
        L.scref_deg2rad_f.argtypes = [C.c_float]
        L.scref_dist_direct.restype = C.c_double
        L.scref_fast_align.restype = C.c_int
        L.scref_ringkey_l2.restype = C.c_float
        L.scref_create.restype = C.c_void_p
        L.scref_destroy.argtypes = [C.c_void_p]
        L.scref_set_dist_thres.argtypes = [C.c_void_p, C.c_double]
        L.scref_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]
        L.scref_size.restype = C.c_int64
        L.scref_size.argtypes = [C.c_void_p]
        L.scref_tree_size.restype = C.c_int64
        L.scref_tree_size.argtypes = [C.c_void_p]
        L.scref_add_points.restype = C.c_int64
        L.scref_add_points.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.scref_add_descriptor.restype = C.c_int64
        L.scref_add_descriptor.argtypes = [C.c_void_p, C.c_void_p]
        L.scref_get_descriptor.restype = C.POINTER(C.c_double)
        L.scref_get_descriptor.argtypes = [C.c_void_p, C.c_int64]
        L.scref_get_ringkey_f32.restype = C.POINTER(C.c_float)
        L.scref_get_ringkey_f32.argtypes = [C.c_void_p, C.c_int64]
        L.scref_get_sectorkey.restype = C.POINTER(C.c_double)
        L.scref_get_sectorkey.argtypes = [C.c_void_p, C.c_int64]
        L.scref_knn.restype = C.c_int
        L.scref_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.scref_detect_loop_closure.restype = C.c_int
        L.scref_detect_loop_closure.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.scref_detect_between_session.restype = C.c_int
        L.scref_detect_between_session.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.scref_exhaustive.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int]
        L.scref_exhaustive_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int]
        L.scref_pair_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.scref_merge_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.scref_make_scancontext.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_void_p]
        L.scref_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.scref_distance_literal.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.scref_circshift.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.scref_ringkey.argtypes = [C.c_void_p, C.c_void_p]
        L.scref_sectorkey.argtypes = [C.c_void_p, C.c_void_p]
        L.scref_ringkey_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.scref_dist_direct.argtypes = [C.c_void_p, C.c_void_p]
        L.scref_fast_align.argtypes = [C.c_void_p, C.c_void_p]
        L.scref_ringkey_l2.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.scref_set_knn_mode.argtypes = [C.c_void_p, C.c_int]
        L.kdref_build.restype = C.c_void_p
        L.kdref_build.argtypes = [C.c_void_p, C.c_int64]
        L.kdref_free.argtypes = [C.c_void_p]
        L.kdref_vind.argtypes = [C.c_void_p, C.c_void_p]
        L.kdref_knn.restype = C.c_int
        L.kdref_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own nanoflann kd-tree (oracle/_ref); None when it was never built."""
    global _ref
    if _ref is None:
        build()
        p = os.path.join(_HERE, "_ref", "libref_kdtree.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_kdtree_build.restype = C.c_void_p
        R.ref_kdtree_build.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        R.ref_kdtree_knn.restype = C.c_int
        R.ref_kdtree_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_kdtree_free.argtypes = [C.c_void_p]
        _ref = R
    return _ref


# ------------------------------------------------------------------------------------------
# numpy-level helpers
# ------------------------------------------------------------------------------------------

def xy2theta(x, y):
    return lib().scref_xy2theta(float(np.float32(x)), float(np.float32(y)))


def make_scancontext(pts, lidar_height=2.0, max_radius=80.0):
    """pts: (n, >=3) float32 -> (1200,) float64 column-major 20x60."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty(DS, dtype=np.float64)
    lib().scref_make_scancontext(pts.ctypes.data, pts.shape[0], pts.shape[1], lidar_height, max_radius, out.ctypes.data)
    return out


def ringkey(desc):
    desc = np.ascontiguousarray(desc, dtype=np.float64)
    out = np.empty(NR, dtype=np.float64)
    lib().scref_ringkey(desc.ctypes.data, out.ctypes.data)
    return out


def ringkey_f32(desc):
    desc = np.ascontiguousarray(desc, dtype=np.float64)
    out = np.empty(NR, dtype=np.float32)
    lib().scref_ringkey_f32(desc.ctypes.data, out.ctypes.data)
    return out


def sectorkey(desc):
    desc = np.ascontiguousarray(desc, dtype=np.float64)
    out = np.empty(NS, dtype=np.float64)
    lib().scref_sectorkey(desc.ctypes.data, out.ctypes.data)
    return out


def circshift(desc, k):
    desc = np.ascontiguousarray(desc, dtype=np.float64)
    out = np.empty_like(desc)
    lib().scref_circshift(desc.ctypes.data, NR, NS, int(k), out.ctypes.data)
    return out


def dist_direct(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().scref_dist_direct(a.ctypes.data, b.ctypes.data)


def fast_align(v1, v2):
    v1 = np.ascontiguousarray(v1, dtype=np.float64)
    v2 = np.ascontiguousarray(v2, dtype=np.float64)
    return lib().scref_fast_align(v1.ctypes.data, v2.ctypes.data)


def distance(a, b, search_ratio=0.1, literal=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    d = C.c_double()
    s = C.c_int()
    f = lib().scref_distance_literal if literal else lib().scref_distance
    f(a.ctypes.data, b.ctypes.data, search_ratio, C.byref(d), C.byref(s))
    return d.value, s.value


def ringkey_l2(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return lib().scref_ringkey_l2(a.ctypes.data, b.ctypes.data, a.size)


class Manager:
    """Oracle counterpart of the reference SCManager (Scancontext.h:62-122)."""

    def __init__(self, dist_thres=None):
        self._L = lib()
        self._h = self._L.scref_create()
        if dist_thres is not None:
            self.set_dist_thres(dist_thres)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.scref_destroy(self._h)
            self._h = None

    def set_dist_thres(self, t):
        self._L.scref_set_dist_thres(self._h, float(t))

    def set_knn_mode(self, tree_order=True):
        """Candidate stage: True (default) = nanoflann's tree restated (ties in the reference's order),
        False = brute force, lower index first among equal distances."""
        self._L.scref_set_knn_mode(self._h, 1 if tree_order else 0)

    def set_params(self, lidar_height=2.0, max_radius=80.0, num_exclude_recent=30, num_candidates=3,
                   tree_making_period=30, search_ratio=0.1):
        self._L.scref_set_params(self._h, lidar_height, max_radius, num_exclude_recent, num_candidates,
                                 tree_making_period, search_ratio)

    def __len__(self):
        return self._L.scref_size(self._h)

    @property
    def tree_size(self):
        return self._L.scref_tree_size(self._h)

    def add_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        return self._L.scref_add_points(self._h, pts.ctypes.data, pts.shape[0], pts.shape[1])

    def add_descriptor(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.float64).reshape(-1)
        assert desc.size == DS
        return self._L.scref_add_descriptor(self._h, desc.ctypes.data)

    def add_descriptors(self, descs):
        for d in descs:
            self.add_descriptor(d)

    def descriptor(self, i):
        return np.ctypeslib.as_array(self._L.scref_get_descriptor(self._h, i), shape=(DS,)).copy()

    def ringkey_f32(self, i):
        return np.ctypeslib.as_array(self._L.scref_get_ringkey_f32(self._h, i), shape=(NR,)).copy()

    def sectorkey(self, i):
        return np.ctypeslib.as_array(self._L.scref_get_sectorkey(self._h, i), shape=(NS,)).copy()

    def knn(self, key, n_search, k=3):
        key = np.ascontiguousarray(key, dtype=np.float32)
        idx = np.zeros(k, dtype=np.int64)
        dist = np.zeros(k, dtype=np.float32)
        found = self._L.scref_knn(self._h, key.ctypes.data, n_search, k, idx.ctypes.data, dist.ctypes.data)
        return found, idx, dist

    def detect_loop_closure(self):
        yaw = C.c_float()
        md = C.c_double()
        nn = C.c_int()
        lid = self._L.scref_detect_loop_closure(self._h, C.byref(yaw), C.byref(md), C.byref(nn))
        return lid, yaw.value, md.value, nn.value

    def detect_between_session(self, key, desc):
        key = np.ascontiguousarray(key, dtype=np.float32)
        desc = np.ascontiguousarray(desc, dtype=np.float64)
        yaw = C.c_float()
        md = C.c_double()
        nn = C.c_int()
        lid = self._L.scref_detect_between_session(self._h, key.ctypes.data, desc.ctypes.data,
                                                   C.byref(yaw), C.byref(md), C.byref(nn))
        return lid, yaw.value, md.value, nn.value

    def exhaustive(self, qdesc, n_eligible=None, k=1, nthreads=1):
        qdesc = np.ascontiguousarray(qdesc, dtype=np.float64)
        if n_eligible is None:
            n_eligible = len(self)
        out = np.zeros(k, dtype=HIT_DTYPE)
        self._L.scref_exhaustive(self._h, qdesc.ctypes.data, n_eligible, k, out.ctypes.data, nthreads)
        return out

    def exhaustive_batch(self, qdescs, n_eligible=None, k=1, nthreads=1):
        q = np.ascontiguousarray(qdescs, dtype=np.float64).reshape(-1, DS)
        if n_eligible is None:
            n_eligible = len(self)
        out = np.zeros((q.shape[0], k), dtype=HIT_DTYPE)
        self._L.scref_exhaustive_batch(self._h, q.ctypes.data, q.shape[0], n_eligible, k, out.ctypes.data, nthreads)
        return out

    def pair_distances(self, qdesc, first=0, count=None, nthreads=1):
        qdesc = np.ascontiguousarray(qdesc, dtype=np.float64)
        if count is None:
            count = len(self) - first
        dist = np.empty(count, dtype=np.float64)
        shift = np.empty(count, dtype=np.int32)
        self._L.scref_pair_distances(self._h, qdesc.ctypes.data, first, count, dist.ctypes.data,
                                     shift.ctypes.data, nthreads)
        return dist, shift


def merge_topk(parts, k):
    """parts: (nparts, k) HIT_DTYPE -> (k,) HIT_DTYPE."""
    parts = np.ascontiguousarray(parts, dtype=HIT_DTYPE)
    out = np.zeros(k, dtype=HIT_DTYPE)
    lib().scref_merge_topk(parts.ctypes.data, parts.shape[0], k, out.ctypes.data)
    return out


class KdTree:
    """oracle/kdtree_ref.c: nanoflann's tree and search restated (what the oracle's detectors use)."""

    def __init__(self, keys):
        self._L = lib()
        self._keys = np.ascontiguousarray(keys, dtype=np.float32)  # must outlive the tree
        assert self._keys.ndim == 2 and self._keys.shape[1] == NR
        self._h = self._L.kdref_build(self._keys.ctypes.data, self._keys.shape[0])

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.kdref_free(self._h)
            self._h = None

    def knn(self, q, k=3):
        q = np.ascontiguousarray(q, dtype=np.float32)
        idx = np.zeros(k, dtype=np.int64)
        dist = np.zeros(k, dtype=np.float32)
        n = self._L.kdref_knn(self._h, q.ctypes.data, k, idx.ctypes.data, dist.ctypes.data)
        return n, idx, dist

    def vind(self):
        out = np.zeros(self._keys.shape[0], dtype=np.int64)
        self._L.kdref_vind(self._h, out.ctypes.data)
        return out


class RefKdTree:
    """The reference's InvKeyTree (nanoflann) compiled from /root/reference (oracle/_ref)."""

    def __init__(self, keys):
        self._R = ref_lib()
        if self._R is None:
            raise RuntimeError("oracle/_ref/libref_kdtree.so not built")
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        self._h = self._R.ref_kdtree_build(keys.ctypes.data, keys.shape[0], keys.shape[1])

    def __del__(self):
        if getattr(self, "_h", None):
            self._R.ref_kdtree_free(self._h)
            self._h = None

    def knn(self, q, k=3):
        q = np.ascontiguousarray(q, dtype=np.float32)
        idx = np.zeros(k, dtype=np.uint64)
        dist = np.zeros(k, dtype=np.float32)
        n = self._R.ref_kdtree_knn(self._h, q.ctypes.data, k, idx.ctypes.data, dist.ctypes.data)
        return n, idx.astype(np.int64), dist


# ------------------------------------------------------------------------------------------
# ORORA (oracle/orora_ref.c) -- PARITY UNPINNED, see the header of orora_ref.h
# ------------------------------------------------------------------------------------------
class OroraParams(C.Structure):
    _fields_ = [("tim_noise_bound", C.c_double), ("noise_bound_radial", C.c_double),
                ("noise_bound_tangential", C.c_double), ("gnc_factor", C.c_double),
                ("cost_threshold", C.c_double), ("max_iterations", C.c_int32), ("flags", C.c_int32)]


ORORA_RESULT_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("yaw", "<f8"), ("iterations", "<i4"),
                               ("rot_inliers", "<i4"), ("trans_inliers", "<i4"), ("status", "<i4")])


def orora_default_params():
    p = OroraParams()
    lib().ororaref_default_params(C.byref(p))
    return p


def orora_register_batch(src, dst, offsets, params=None, nthreads=1):
    L = lib()
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.ascontiguousarray(dst, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if params is None:
        params = orora_default_params()
    n = offsets.size - 1
    out = np.zeros(n, dtype=ORORA_RESULT_DTYPE)
    L.ororaref_register_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int]
    L.ororaref_register_batch(src.ctypes.data, dst.ctypes.data, offsets.ctypes.data, n, C.byref(params),
                              out.ctypes.data, nthreads)
    return out


def orora_scalar_tls(x, beta):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    L.ororaref_scalar_tls.restype = C.c_double
    L.ororaref_scalar_tls.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    n_in = C.c_int32()
    est = L.ororaref_scalar_tls(x.ctypes.data, beta.ctypes.data, x.size, C.byref(n_in))
    return est, n_in.value


# ------------------------------------------------------------------------------------------
# max-clique inlier selection before the solver (oracle/pmc_ref.c) -- PARITY UNPINNED, see pmc_ref.h
# ------------------------------------------------------------------------------------------
PMC_INFO_DTYPE = np.dtype([("size", "<i4"), ("max_core", "<i4"), ("seeds", "<i4"), ("flags", "<i4")])
PMC_PROVEN, PMC_PASSTHROUGH, PMC_MAX_K = 1, 2, 2048


def pmc_adjacency(src, dst, tau):
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.ascontiguousarray(dst, dtype=np.float32)
    k = len(src)
    adj = np.zeros((k, k), dtype=np.uint8)
    L = lib()
    L.pmcref_adjacency.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p]
    L.pmcref_adjacency.restype = None
    L.pmcref_adjacency(src.ctypes.data, dst.ctypes.data, k, float(tau), adj.ctypes.data)
    return adj


def pmc_core_numbers(adj):
    adj = np.ascontiguousarray(adj, dtype=np.uint8)
    k = len(adj)
    core = np.zeros(k, dtype=np.int32)
    L = lib()
    L.pmcref_core_numbers.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.pmcref_core_numbers.restype = None
    L.pmcref_core_numbers(adj.ctypes.data, k, core.ctypes.data)
    return core


def pmc_select_batch(src, dst, offsets, tau, nthreads=1):
    """-> (member uint8[M] concatenated like the matches, info[n_pairs])"""
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.ascontiguousarray(dst, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    member = np.zeros(int(offsets[-1]), dtype=np.uint8)
    info = np.zeros(n, dtype=PMC_INFO_DTYPE)
    L = lib()
    L.pmcref_select_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    L.pmcref_select_batch.restype = None
    L.pmcref_select_batch(src.ctypes.data, dst.ctypes.data, offsets.ctypes.data, n, float(tau), member.ctypes.data, info.ctypes.data, nthreads)
    return member, info


def pmc_exact_size(adj, lb=0, max_nodes=2_000_000):
    adj = np.ascontiguousarray(adj, dtype=np.uint8)
    L = lib()
    L.pmcref_exact_size.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64]
    L.pmcref_exact_size.restype = C.c_int32
    return int(L.pmcref_exact_size(adj.ctypes.data, len(adj), int(lb), int(max_nodes)))


def pmc_compact(src, dst, offsets, member):
    """the matches a selection keeps, in their original order, with their offsets: what the solver is fed"""
    keep = member.astype(bool)
    offsets = np.asarray(offsets, dtype=np.int64)
    cnt = np.add.reduceat(keep.astype(np.int64), offsets[:-1]) if len(offsets) > 1 else np.zeros(0, dtype=np.int64)
    cnt[np.diff(offsets) == 0] = 0
    new_off = np.zeros(len(offsets), dtype=np.int64)
    new_off[1:] = np.cumsum(cnt)
    return np.ascontiguousarray(src[keep]), np.ascontiguousarray(dst[keep]), new_off


# ------------------------------------------------------------------------------------------
# cen2019 (oracle/cen2019_ref.c) -- PARITY UNPINNED, see the header of cen2019_ref.c
# ------------------------------------------------------------------------------------------
def cen2019_extract(img, cols=None, col_offset=11, max_points=10000, min_range=58, max_targets=200000, debug=False):
    """img: (rows, row_stride) uint8 -> (n,2) int32 (azimuth idx, range idx) [+ debug dict]."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, stride = img.shape
    if cols is None:
        cols = stride - col_offset
    out = np.zeros((max_targets, 2), dtype=np.int32)
    L.cen2019ref_extract.restype = C.c_int32
    L.cen2019ref_extract.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    h = np.empty((rows, cols), dtype=np.float32) if debug else None
    mean_h, ncand, jstar = C.c_float(), C.c_int64(), C.c_int64()
    n = L.cen2019ref_extract(img.ctypes.data, rows, cols, stride, col_offset, max_points, min_range, out.ctypes.data,
                             max_targets, h.ctypes.data if debug else None, C.byref(mean_h), C.byref(ncand), C.byref(jstar))
    res = out[:min(n, max_targets)].copy()
    if debug:
        return res, {"h": h, "mean_h": mean_h.value, "ncand": ncand.value, "jstar": jstar.value, "count": n}
    return res


def cen2019_to_cartesian(targets, azimuths, resolution):
    L = lib()
    t = np.ascontiguousarray(targets, dtype=np.int32)
    az = np.ascontiguousarray(azimuths, dtype=np.float32)
    out = np.empty((t.shape[0], 2), dtype=np.float32)
    L.cen2019ref_to_cartesian.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p]
    L.cen2019ref_to_cartesian(t.ctypes.data, t.shape[0], az.ctypes.data, resolution, out.ctypes.data)
    return out


# ---------------------------------------------------------------------------------------------
# VoxelGrid downsample (oracle/voxelgrid_ref.c) -- PARITY UNPINNED, see the header of that file
# ---------------------------------------------------------------------------------------------
def voxelgrid_filter(pts, leaf=0.4, intensity_col=3):
    """pts: (n, >=3) float32 rows x,y,z[,intensity at column intensity_col].  -> ((m,4) float32, overflow)."""
    L = lib()
    p = np.ascontiguousarray(pts, dtype=np.float32)
    n = p.shape[0]
    out = np.zeros((max(n, 1), 4), dtype=np.float32)
    ov = C.c_int32()
    L.vgref_filter.restype = C.c_int64
    L.vgref_filter.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
    ioff = 4 * intensity_col if (intensity_col is not None and p.shape[1] > intensity_col) else -1
    m = L.vgref_filter(p.ctypes.data, n, p.shape[1] * 4, ioff, leaf, out.ctypes.data, out.shape[0], C.byref(ov))
    return out[:m].copy(), bool(ov.value)


# ---------------------------------------------------------------------------------------------
# ICP loop verification (oracle/icp_ref.c) -- PARITY UNPINNED, see the header of that file
# ---------------------------------------------------------------------------------------------
ICP_SUM_SEQUENTIAL_FLOAT, ICP_SUM_TREE = 0, 1


class IcpRefParams(C.Structure):
    _fields_ = [("max_corr_dist", C.c_double), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double), ("max_iterations", C.c_int32), ("sum_order", C.c_int32)]


class IcpRefResult(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("fitness", C.c_double), ("iterations", C.c_int32),
                ("converged", C.c_int32), ("state", C.c_int32), ("reserved", C.c_int32)]


def icp_rotation_from_covariance(H):
    L = lib()
    h = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    r = np.zeros(9, dtype=np.float64)
    L.icpref_rotation_from_covariance(_dp(h), _dp(r))
    return r.reshape(3, 3)


def icp_align(source, target, max_corr_dist=150.0, transformation_epsilon=1e-6, euclidean_fitness_epsilon=1e-6,
              max_iterations=100, guess=None, sum_order=0):
    """sum_order: ICP_SUM_SEQUENTIAL_FLOAT (0) or ICP_SUM_TREE (1, the order the device kernel adds in: icp_ref.c)"""
    L = lib()
    s = np.ascontiguousarray(np.asarray(source, dtype=np.float32)[:, :3])
    t = np.ascontiguousarray(np.asarray(target, dtype=np.float32)[:, :3])
    p = IcpRefParams(max_corr_dist, transformation_epsilon, euclidean_fitness_epsilon, max_iterations, sum_order)
    g = np.ascontiguousarray(guess, dtype=np.float32).reshape(16) if guess is not None else None
    r = IcpRefResult()
    L.icpref_align.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(IcpRefParams), C.c_void_p, C.POINTER(IcpRefResult)]
    L.icpref_align(s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], C.byref(p), g.ctypes.data if g is not None else None, C.byref(r))
    return {"transform": np.array(r.transform, dtype=np.float32).reshape(4, 4), "fitness": r.fitness,
            "iterations": r.iterations, "converged": bool(r.converged), "state": r.state}


# ---------------------------------------------------------------------------------------------
# Loop verification chain + map assembly (oracle/loopverify_ref.c): PGO.cpp:199-220, 329-406, 631-655
# ---------------------------------------------------------------------------------------------
class LvRefResult(C.Structure):
    _fields_ = [("accepted", C.c_int32), ("converged", C.c_int32), ("iterations", C.c_int32), ("state", C.c_int32),
                ("fitness", C.c_double), ("transform", C.c_float * 16),
                ("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("roll", C.c_float), ("pitch", C.c_float), ("yaw", C.c_float),
                ("relative", C.c_double * 16), ("n_source", C.c_int64), ("n_target", C.c_int64)]


def _kf_pack(clouds):
    """list of (n_i, 4) float32 keyframe clouds -> (packed (N, 4) float32, offsets int64[nkf + 1])"""
    off = np.zeros(len(clouds) + 1, dtype=np.int64)
    for i, c in enumerate(clouds):
        off[i + 1] = off[i] + len(c)
    allp = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1, 4) for c in clouds]), dtype=np.float32) \
        if len(clouds) and off[-1] else np.zeros((1, 4), dtype=np.float32)
    return allp, off


def pose_matrix(pose6):
    """pcl::getTransformation(x, y, z, roll, pitch, yaw) as the reference calls it (float): (4, 4) float32"""
    L = lib()
    p = np.ascontiguousarray(pose6, dtype=np.float64).reshape(6)
    t = np.zeros(16, dtype=np.float32)
    L.lvref_pose_matrix.argtypes = [C.c_void_p, C.c_void_p]
    L.lvref_pose_matrix(p.ctypes.data, t.ctypes.data)
    return t.reshape(4, 4)


def loop_submap(clouds, key, submap_size, root_pose, leaf=0.4):
    """loopFindNearKeyframesCloud (PGO.cpp:329-352) -> (m, 4) float32"""
    L = lib()
    allp, off = _kf_pack(clouds)
    out = np.zeros((max(int(off[-1]), 1), 4), dtype=np.float32)
    rp = np.ascontiguousarray(root_pose, dtype=np.float64).reshape(6)
    L.lvref_submap.restype = C.c_int64
    L.lvref_submap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_int64]
    m = L.lvref_submap(allp.ctypes.data, off.ctypes.data, len(clouds), key, submap_size, rp.ctypes.data, leaf, out.ctypes.data, out.shape[0])
    return out[:m].copy()


def loop_verify(clouds, loop_idx, curr_idx, root_pose, history_num=25, leaf=0.4, fitness_threshold=0.3,
                max_corr_dist=150.0, transformation_epsilon=1e-6, euclidean_fitness_epsilon=1e-6, max_iterations=100, sum_order=0):
    """doICPVirtualRelative (PGO.cpp:355-406) -> dict"""
    L = lib()
    allp, off = _kf_pack(clouds)
    rp = np.ascontiguousarray(root_pose, dtype=np.float64).reshape(6)
    p = IcpRefParams(max_corr_dist, transformation_epsilon, euclidean_fitness_epsilon, max_iterations, sum_order)
    r = LvRefResult()
    L.lvref_verify.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                               C.POINTER(IcpRefParams), C.c_double, C.POINTER(LvRefResult)]
    L.lvref_verify(allp.ctypes.data, off.ctypes.data, len(clouds), loop_idx, curr_idx, rp.ctypes.data, history_num, leaf, C.byref(p),
                   fitness_threshold, C.byref(r))
    return {"accepted": bool(r.accepted), "converged": bool(r.converged), "iterations": r.iterations, "state": r.state,
            "fitness": r.fitness, "transform": np.array(r.transform, dtype=np.float32).reshape(4, 4),
            "xyz_rpy": np.array([r.x, r.y, r.z, r.roll, r.pitch, r.yaw], dtype=np.float32),
            "relative": np.array(r.relative, dtype=np.float64).reshape(4, 4), "n_source": r.n_source, "n_target": r.n_target}


def map_build(clouds, poses, skip=2, leaf=0.4):
    """pubMap's cloud (PGO.cpp:631-655) -> (m, 4) float32"""
    L = lib()
    allp, off = _kf_pack(clouds)
    ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(len(clouds), 6)
    out = np.zeros((max(int(off[-1]), 1), 4), dtype=np.float32)
    L.lvref_map.restype = C.c_int64
    L.lvref_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int64]
    m = L.lvref_map(allp.ctypes.data, off.ctypes.data, len(clouds), ps.ctypes.data, skip, leaf, out.ctypes.data, out.shape[0])
    return out[:m].copy()


# ------------------------------------------------------------------------------------------
# The reference's own Scancontext.cpp, compiled unmodified against oracle/standin (oracle/ref_sc.cpp)
# ------------------------------------------------------------------------------------------
ORDER_SEQ, ORDER_EIGEN_SSE2, ORDER_EIGEN_AVX_FMA, ORDER_EIGEN34_AVX_FMA = 0, 1, 2, 3
REF_SC_VARIANTS = {ORDER_SEQ: "seq", ORDER_EIGEN_SSE2: "sse2", ORDER_EIGEN_AVX_FMA: "avxfma", ORDER_EIGEN34_AVX_FMA: "avxfma34"}


def set_sum_order(order):
    """Summation order of the oracle's Eigen-style reductions (sc_ref.h SCREF_ORDER_*)."""
    lib().scref_set_sum_order(int(order))


def get_sum_order():
    return lib().scref_get_sum_order()


class RefSC:
    """ctypes view of oracle/_ref/libref_sc_<variant>.so = /root/reference's Scancontext.cpp itself."""

    def __init__(self, order=ORDER_EIGEN_SSE2):
        build()
        p = os.path.join(_HERE, "_ref", f"libref_sc_{REF_SC_VARIANTS[order]}.so")
        if not os.path.exists(p):
            raise FileNotFoundError(p)
        if order in (ORDER_EIGEN_AVX_FMA, ORDER_EIGEN34_AVX_FMA) and " fma " not in open("/proc/cpuinfo").read():
            raise FileNotFoundError("host CPU has no FMA: cannot run " + p)
        R = C.CDLL(p)
        R.ref_sc_build_info.restype = C.c_char_p
        R.ref_sc_xy2theta.restype = C.c_float
        R.ref_sc_xy2theta.argtypes = [C.c_float, C.c_float]
        R.ref_sc_make_scancontext.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        R.ref_sc_ringkey.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_ringkey_f32.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_sectorkey.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_dist_direct.restype = C.c_double
        R.ref_sc_dist_direct.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_fast_align.restype = C.c_int
        R.ref_sc_fast_align.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_sc_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        R.ref_sc_distances_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        R.ref_sc_circshift.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        R.ref_sc_create.restype = C.c_void_p
        R.ref_sc_destroy.argtypes = [C.c_void_p]
        R.ref_sc_set_dist_thres.argtypes = [C.c_void_p, C.c_double]
        R.ref_sc_size.restype = C.c_int64
        R.ref_sc_size.argtypes = [C.c_void_p]
        R.ref_sc_add_points.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        R.ref_sc_add_descriptor.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_get.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_sc_detect_loop_closure.restype = C.c_int
        R.ref_sc_detect_loop_closure.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_sc_detect_between_session.restype = C.c_int
        R.ref_sc_detect_between_session.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_sc_last_log.restype = C.c_char_p
        self.R = R
        self.order = order

    def build_info(self):
        return self.R.ref_sc_build_info().decode()

    def distances_batch(self, queries, descs, nthreads=1):
        """every (query, entry) pair through the reference's distanceBtnScanContext, OpenMP over (query, entry block)
        -> (dist (nq, n) float64, shift (nq, n) int32)"""
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, DS)
        d = np.ascontiguousarray(descs, dtype=np.float64).reshape(-1, DS)
        dist = np.empty((q.shape[0], d.shape[0]), dtype=np.float64)
        shift = np.empty((q.shape[0], d.shape[0]), dtype=np.int32)
        self.R.ref_sc_distances_batch(q.ctypes.data, q.shape[0], d.ctypes.data, d.shape[0], dist.ctypes.data, shift.ctypes.data, int(nthreads))
        return dist, shift

    def xy2theta(self, x, y):
        return self.R.ref_sc_xy2theta(float(np.float32(x)), float(np.float32(y)))

    def make_scancontext(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.empty(DS, dtype=np.float64)
        self.R.ref_sc_make_scancontext(pts.ctypes.data, pts.shape[0], pts.shape[1], out.ctypes.data)
        return out

    def ringkey(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.float64)
        out = np.empty(NR, dtype=np.float64)
        self.R.ref_sc_ringkey(desc.ctypes.data, out.ctypes.data)
        return out

    def ringkey_f32(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.float64)
        out = np.empty(NR, dtype=np.float32)
        self.R.ref_sc_ringkey_f32(desc.ctypes.data, out.ctypes.data)
        return out

    def sectorkey(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.float64)
        out = np.empty(NS, dtype=np.float64)
        self.R.ref_sc_sectorkey(desc.ctypes.data, out.ctypes.data)
        return out

    def dist_direct(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return self.R.ref_sc_dist_direct(a.ctypes.data, b.ctypes.data)

    def fast_align(self, v1, v2):
        v1 = np.ascontiguousarray(v1, dtype=np.float64)
        v2 = np.ascontiguousarray(v2, dtype=np.float64)
        return self.R.ref_sc_fast_align(v1.ctypes.data, v2.ctypes.data)

    def circshift(self, desc, k):
        desc = np.ascontiguousarray(desc, dtype=np.float64)
        out = np.empty_like(desc)
        self.R.ref_sc_circshift(desc.ctypes.data, NR, NS, int(k), out.ctypes.data)
        return out

    def distance(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        d, s = C.c_double(), C.c_int()
        self.R.ref_sc_distance(a.ctypes.data, b.ctypes.data, C.byref(d), C.byref(s))
        return d.value, s.value

    def distances(self, q, descs):
        q = np.ascontiguousarray(q, dtype=np.float64)
        descs = np.ascontiguousarray(descs, dtype=np.float64).reshape(-1, DS)
        dist = np.empty(descs.shape[0], dtype=np.float64)
        shift = np.empty(descs.shape[0], dtype=np.int32)
        self.R.ref_sc_distances(q.ctypes.data, descs.ctypes.data, descs.shape[0], dist.ctypes.data, shift.ctypes.data)
        return dist, shift


class RefManager:
    """The reference's SCManager object itself (Scancontext.h:57-122) behind ref_sc.cpp."""

    def __init__(self, order=ORDER_EIGEN_SSE2, dist_thres=None):
        self._sc = RefSC(order)
        self._R = self._sc.R
        self._h = self._R.ref_sc_create()
        if dist_thres is not None:
            self._R.ref_sc_set_dist_thres(self._h, float(dist_thres))

    def __del__(self):
        if getattr(self, "_h", None):
            self._R.ref_sc_destroy(self._h)
            self._h = None

    def __len__(self):
        return self._R.ref_sc_size(self._h)

    def add_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        self._R.ref_sc_add_points(self._h, pts.ctypes.data, pts.shape[0], pts.shape[1])

    def add_descriptor(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.float64).reshape(-1)
        self._R.ref_sc_add_descriptor(self._h, desc.ctypes.data)

    def get(self, i):
        d, rk, sk = np.empty(DS), np.empty(NR, dtype=np.float32), np.empty(NS)
        self._R.ref_sc_get(self._h, i, d.ctypes.data, rk.ctypes.data, sk.ctypes.data)
        return d, rk, sk

    def detect_loop_closure(self):
        yaw = C.c_float()
        lid = self._R.ref_sc_detect_loop_closure(self._h, C.byref(yaw))
        return lid, yaw.value

    def last_log(self):
        """stdout of the last detect call (Scancontext.cpp:406,412)."""
        return self._R.ref_sc_last_log().decode()

    def detect_between_session(self, key, desc):
        key = np.ascontiguousarray(key, dtype=np.float32)
        desc = np.ascontiguousarray(desc, dtype=np.float64)
        yaw = C.c_float()
        lid = self._R.ref_sc_detect_between_session(self._h, key.ctypes.data, desc.ctypes.data, C.byref(yaw))
        return lid, yaw.value


# ------------------------------------------------------------------------------------------
# ORORA front end (oracle/frontend_ref.c) -- PARITY UNPINNED, see the header of frontend_ref.c
# ------------------------------------------------------------------------------------------
class FrontendRef:
    """Polar -> Cartesian remap, ORB-style descriptors and BF-Hamming knnMatch + ratio on the CPU."""

    def __init__(self, rows=400, cols=3360, W=964, cart_res=0.2592):
        self.L = lib()
        # the C-ABI carries cart_resolution / radar resolution as fp32: the oracle starts from the same fp32 values
        self.rows, self.cols, self.W, self.cart_res = rows, cols, W, float(np.float32(cart_res))
        self.gauss = np.zeros(7, dtype=np.float32)
        self.dirs = np.zeros(60, dtype=np.float32)
        self.pairs = np.zeros(30 * 256 * 4, dtype=np.int8)
        self.L.feref_tables(self.gauss.ctypes.data_as(C.c_void_p), self.dirs.ctypes.data_as(C.c_void_p),
                            self.pairs.ctypes.data_as(C.c_void_p))
        self._map_key = None

    def cartesian(self, img, azimuths, resolution, col_offset=11):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        az = np.asarray(azimuths, dtype=np.float32)
        key = (float(np.float32(resolution)), float(az[0]), float(az[1]) - float(az[0]))
        W = self.W
        if key != self._map_key:
            self.map_rb = np.empty(W * W, dtype=np.float32)
            self.map_ab = np.empty(W * W, dtype=np.float32)
            self.L.feref_cart_map.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
            self.L.feref_cart_map(W, self.cart_res, key[0], key[1], key[2], self.rows, self.map_rb.ctypes.data, self.map_ab.ctypes.data)
            self._map_key = key
        self.cart = np.empty((W, W), dtype=np.float32)
        self.L.feref_cart_remap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.feref_cart_remap(img.ctypes.data, self.rows, self.cols, img.shape[1], col_offset, W, self.map_rb.ctypes.data,
                                self.map_ab.ctypes.data, self.cart.ctypes.data)
        tmp = np.empty((W, W), dtype=np.float32)
        self.blur = np.empty((W, W), dtype=np.float32)
        self.L.feref_blur.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.feref_blur(self.cart.ctypes.data, W, self.gauss.ctypes.data, tmp.ctypes.data, self.blur.ctypes.data)
        return self.cart

    def describe(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        n = xy.shape[0]
        desc = np.zeros((n, 32), dtype=np.uint8)
        valid = np.zeros(n, dtype=np.uint8)
        self.L.feref_describe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        self.L.feref_describe(self.cart.ctypes.data, self.blur.ctypes.data, self.W, self.cart_res, xy.ctypes.data, n,
                              self.dirs.ctypes.data, self.pairs.ctypes.data, desc.ctypes.data, valid.ctypes.data)
        return desc, valid

    def match(self, q, qv, t, tv, ratio=0.8):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        qv = np.ascontiguousarray(qv, dtype=np.uint8)
        tv = np.ascontiguousarray(tv, dtype=np.uint8)
        idx = np.empty(len(q), dtype=np.int32)
        d1 = np.empty(len(q), dtype=np.int32)
        d2 = np.empty(len(q), dtype=np.int32)
        self.L.feref_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        self.L.feref_match(q.ctypes.data, qv.ctypes.data, len(q), t.ctypes.data, tv.ctypes.data, len(t), float(ratio),
                           idx.ctypes.data, d1.ctypes.data, d2.ctypes.data)
        return idx, d1, d2
