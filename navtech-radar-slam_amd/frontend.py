"""ctypes harness of the ORORA front end (rsx_frontend_*: polar -> Cartesian, ORB-style descriptors, BF-Hamming
knnMatch + ratio).  Everything is computed by frontend.hip on the GPU; this only marshals buffers."""
import ctypes as C

import numpy as np

from ._rsx import FrontendParams, check, lib


def default_params():
    p = FrontendParams()
    check(lib().rsx_frontend_default_params(C.byref(p)))
    return p


class Frontend:
    def __init__(self, rows=400, cols=3360, device=0, params=None):
        self._L = lib()
        self.params = params if params is not None else default_params()
        self.rows, self.cols, self.W = rows, cols, self.params.cart_pixel_width
        self._h = C.c_void_p()
        check(self._L.rsx_frontend_create(device, rows, cols, C.byref(self.params), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_frontend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cartesian(self, img, azimuths, resolution, col_offset=11, want_image=True):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        az = np.ascontiguousarray(azimuths, dtype=np.float32)
        out = np.empty((self.W, self.W), dtype=np.float32) if want_image else None
        check(self._L.rsx_frontend_cartesian(self._h, img.ctypes.data, img.shape[1], col_offset, az.ctypes.data, float(resolution),
                                             out.ctypes.data if want_image else None))
        return out

    def cartesian_batch_device(self, d_imgs, n, image_stride, row_stride, d_az, az_stride, resolution, col_offset=11, stream=None):
        """n images resident in HBM (d_imgs: device address) with per-image azimuth grids on the device (d_az, az_stride floats apart;
        0: one grid): asynchronous on `stream` (None: the handle's)."""
        check(self._L.rsx_frontend_cartesian_batch_device_az(self._h, d_imgs, n, image_stride, row_stride, col_offset, d_az, az_stride,
                                                             float(resolution), stream))

    def describe_batch_device(self, d_xy, d_counts, n_images, max_targets, d_desc, d_valid, stream=None):
        """descriptors of the keypoints of every image of the last cartesian_batch_device call (device addresses; asynchronous)"""
        check(self._L.rsx_frontend_describe_batch_device(self._h, d_xy, d_counts, n_images, max_targets, d_desc, d_valid, stream))

    def match_consecutive_device(self, d_desc, d_valid, d_counts, max_targets, first_slot, n_pairs, ratio, d_fwd, d_bwd, stream=None):
        """knnMatch(2) + ratio between consecutive keypoint sets, both directions (device addresses; asynchronous)"""
        check(self._L.rsx_frontend_match_consecutive_device(self._h, d_desc, d_valid, d_counts, max_targets, first_slot, n_pairs,
                                                            float(ratio), d_fwd, d_bwd, stream))

    def read_images(self, image=0):
        """(Cartesian image, smoothed copy) of slot `image` of the last cartesian call (rsx_diag.h parity helper)."""
        cart = np.empty((self.W, self.W), dtype=np.float32)
        blur = np.empty((self.W, self.W), dtype=np.float32)
        check(self._L.rsx_frontend_read_images(self._h, image, cart.ctypes.data, blur.ctypes.data))
        return cart, blur

    def describe(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        n = xy.shape[0]
        desc = np.zeros((n, 32), dtype=np.uint8)
        valid = np.zeros(n, dtype=np.uint8)
        check(self._L.rsx_frontend_describe(self._h, xy.ctypes.data, n, desc.ctypes.data, valid.ctypes.data))
        return desc, valid

    def match(self, q_desc, q_valid, t_desc, t_valid, ratio=None):
        q = np.ascontiguousarray(q_desc, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t_desc, dtype=np.uint8).reshape(-1, 32)
        qv = np.ascontiguousarray(q_valid, dtype=np.uint8)
        tv = np.ascontiguousarray(t_valid, dtype=np.uint8)
        idx = np.full(len(q), -1, dtype=np.int32)
        d1 = np.full(len(q), -1, dtype=np.int32)
        d2 = np.full(len(q), -1, dtype=np.int32)
        check(self._L.rsx_frontend_match(self._h, q.ctypes.data, qv.ctypes.data, len(q), t.ctypes.data, tv.ctypes.data, len(t),
                                         float(self.params.ratio if ratio is None else ratio), idx.ctypes.data, d1.ctypes.data, d2.ctypes.data))
        return idx, d1, d2
