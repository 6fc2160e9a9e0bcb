"""Host-side wrapper of the file-based odometry pipeline of librsx.so (include/rsx.h: rsx_odometry_*): windows of
consecutive polar scans in, one relative motion per scan out.  Test / bench harness; the product entry is host/odometry.cpp."""
import ctypes as C

import numpy as np

from ._rsx import ODOMETRY_SCAN_DTYPE, OdometryParams, check, lib


def default_params():
    p = OdometryParams()
    check(lib().rsx_odometry_default_params(C.byref(p)))
    return p


class Odometry:
    def __init__(self, rows=400, cols=3360, params=None, device=0):
        self._L = lib()
        self.rows, self.cols = rows, cols
        self.params = params if params is not None else default_params()
        self.params.device = device
        self._h = C.c_void_p()
        check(self._L.rsx_odometry_create(C.byref(self.params), rows, cols, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_odometry_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self._L.rsx_odometry_reset(self._h))

    def push(self, imgs, azimuths, want_xy=False, max_xy=None, device_ptr=None):
        """imgs: (n, rows, row_stride) uint8 host array (or, with device_ptr, only its shape / strides are used and the
        bytes are read from that HBM address).  -> structured array (n,) of ODOMETRY_SCAN_DTYPE [, list of xy (k_i, 2)]."""
        n, rows, row_stride = imgs.shape
        assert rows == self.rows
        az = np.ascontiguousarray(azimuths, dtype=np.float32)
        out = np.zeros(n, dtype=ODOMETRY_SCAN_DTYPE)
        mx = (max_xy or self.params.max_keypoints) if want_xy else 0
        xy = np.zeros((n, mx, 2), dtype=np.float32) if want_xy else None
        if device_ptr is None:
            imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
            fn, src = self._L.rsx_odometry_push, imgs.ctypes.data
        else:
            fn, src = self._L.rsx_odometry_push_device, device_ptr
        check(fn(self._h, src, n, imgs.strides[0], row_stride, az.ctypes.data, 1 if az.ndim == 2 else 0, out.ctypes.data,
                 xy.ctypes.data if xy is not None else None, mx))
        if want_xy:
            k = np.minimum(np.minimum(out["n_keypoints"], self.params.max_keypoints), mx)
            return out, [xy[i, :k[i]].copy() for i in range(n)]
        return out
