"""Host-side wrapper of the cen2019 keypoint extraction entry points of librsx.so (include/rsx.h):
polar radar power image in, keypoints (azimuth idx, range idx) and Cartesian points out."""
import ctypes as C

import numpy as np

from ._rsx import Cen2019Params, check, lib


def default_params():
    p = Cen2019Params()
    check(lib().rsx_cen2019_default_params(C.byref(p)))
    return p


class Cen2019:
    def __init__(self, rows=400, cols=3360, device=0):
        self._L = lib()
        self.rows, self.cols = rows, cols
        self._h = C.c_void_p()
        check(self._L.rsx_cen2019_create(device, rows, cols, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_cen2019_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, img, col_offset=11, max_points=10000, min_range=58, azimuths=None, resolution=0.0595,
                max_targets=200000):
        """img: (rows, row_stride) uint8.  -> targets (n,2) int32 [, xy (n,2) float32 if azimuths]."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.shape[0] == self.rows
        p = Cen2019Params(max_points, min_range)
        out = np.zeros((max_targets, 2), dtype=np.int32)
        xy = np.zeros((max_targets, 2), dtype=np.float32) if azimuths is not None else None
        az = np.ascontiguousarray(azimuths, dtype=np.float32) if azimuths is not None else None
        n = C.c_int32()
        check(self._L.rsx_cen2019_extract(self._h, img.ctypes.data, img.shape[1], col_offset, C.byref(p),
                                          az.ctypes.data if az is not None else None, resolution, out.ctypes.data,
                                          xy.ctypes.data if xy is not None else None, max_targets, C.byref(n)))
        k = min(n.value, max_targets)
        if xy is not None:
            return out[:k].copy(), xy[:k].copy()
        return out[:k].copy()

    def extract_batch(self, imgs, col_offset=11, max_points=10000, min_range=58, azimuths=None, resolution=0.0595,
                      max_targets=20000):
        """imgs: (n, rows, row_stride) uint8 -> list of targets (k_i, 2) int32 [, list of xy (k_i, 2) float32]; one chain
        of launches for the whole batch (rsx_cen2019_extract_batch).  azimuths: (rows,) shared or (n, rows)."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        n = imgs.shape[0]
        assert imgs.shape[1] == self.rows
        p = Cen2019Params(max_points, min_range)
        out = np.zeros((n, max_targets, 2), dtype=np.int32)
        az = np.ascontiguousarray(azimuths, dtype=np.float32) if azimuths is not None else None
        xy = np.zeros((n, max_targets, 2), dtype=np.float32) if az is not None else None
        counts = np.zeros(n, dtype=np.int32)
        check(self._L.rsx_cen2019_extract_batch(self._h, imgs.ctypes.data, n, imgs.strides[0], imgs.shape[2], col_offset,
                                                C.byref(p), az.ctypes.data if az is not None else None,
                                                1 if (az is not None and az.ndim == 2) else 0, resolution, out.ctypes.data,
                                                xy.ctypes.data if xy is not None else None, max_targets, counts.ctypes.data))
        ks = np.minimum(counts, max_targets)
        tg = [out[i, :ks[i]].copy() for i in range(n)]
        if xy is not None:
            return tg, [xy[i, :ks[i]].copy() for i in range(n)]
        return tg
