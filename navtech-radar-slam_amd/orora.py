"""Host-side wrapper of the ORORA registration entry points of librsx.so (include/rsx.h).

Mirrors the solver call of the upstream file-based odometry entry: matched 2-D feature points of
two consecutive radar scans in, SE(2) pose out.  Computation happens in orora.hip on the GPU."""
import ctypes as C

import numpy as np

from . import _rsx
from ._rsx import ORORA_PMC, ORORA_RESULT_DTYPE, PMC_INFO_DTYPE, OroraParams, check, lib  # noqa: F401


def default_params():
    p = OroraParams()
    check(lib().rsx_orora_default_params(C.byref(p)))
    return p


def max_correspondences():
    return lib().rsx_orora_max_correspondences()


class Orora:
    def __init__(self, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.rsx_orora_create(device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_orora_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def register_batch(self, src_xy, dst_xy, offsets, params=None):
        """src_xy, dst_xy: (M,2) float32; offsets: (n_pairs+1,) int64 -> (n_pairs,) ORORA_RESULT_DTYPE."""
        src = np.ascontiguousarray(src_xy, dtype=np.float32)
        dst = np.ascontiguousarray(dst_xy, dtype=np.float32)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n = off.size - 1
        out = np.zeros(n, dtype=ORORA_RESULT_DTYPE)
        pp = C.byref(params) if params is not None else None
        check(self._L.rsx_orora_register_batch(self._h, src.ctypes.data, dst.ctypes.data, off.ctypes.data, n, pp,
                                               out.ctypes.data))
        return out

    def register(self, src_xy, dst_xy, params=None):
        src = np.ascontiguousarray(src_xy, dtype=np.float32)
        return self.register_batch(src, dst_xy, np.array([0, src.shape[0]], dtype=np.int64), params)[0]

    def register_batch_device(self, src_ptr, dst_ptr, off_ptr, n_pairs, out_ptr, params=None, stream=0):
        pp = C.byref(params) if params is not None else None
        check(self._L.rsx_orora_register_batch_device(self._h, src_ptr, dst_ptr, off_ptr, n_pairs, pp, out_ptr, stream))

    def reserve(self, max_total_matches):
        """sizes the workspaces of the RSX_ORORA_PMC stage for the asynchronous device entry"""
        check(self._L.rsx_orora_reserve(self._h, int(max_total_matches)))

    def max_clique_batch(self, src_xy, dst_xy, offsets, params=None):
        """the max-clique inlier selection on its own -> (member uint8 (M,), info (n_pairs,) PMC_INFO_DTYPE)"""
        src = np.ascontiguousarray(src_xy, dtype=np.float32)
        dst = np.ascontiguousarray(dst_xy, dtype=np.float32)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n = off.size - 1
        member = np.zeros(int(off[-1]), dtype=np.uint8)
        info = np.zeros(n, dtype=PMC_INFO_DTYPE)
        pp = C.byref(params) if params is not None else None
        check(self._L.rsx_orora_max_clique_batch(self._h, src.ctypes.data, dst.ctypes.data, off.ctypes.data, n, pp, member.ctypes.data, info.ctypes.data))
        return member, info

    def last_pmc_info(self, n_pairs):
        info = np.zeros(n_pairs, dtype=PMC_INFO_DTYPE)
        check(self._L.rsx_orora_last_pmc_info(self._h, info.ctypes.data, n_pairs))
        return info
