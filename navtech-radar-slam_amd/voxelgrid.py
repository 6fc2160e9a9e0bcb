"""Host-side wrapper of the VoxelGrid entry points of librsx.so (include/rsx.h): the reference's
`pcl::VoxelGrid<PointType> downSizeFilterScancontext` (laserPosegraphOptimization.cpp:98,482-484)."""
import ctypes as C

import numpy as np

from ._rsx import check, lib


class VoxelGrid:
    def __init__(self, leaf=0.4, device=0):
        self._L = lib()
        self.leaf = float(leaf)  # setLeafSize(0.4, 0.4, 0.4), laserPosegraphOptimization.cpp:687-688
        self._h = C.c_void_p()
        check(self._L.rsx_voxelgrid_create(device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_voxelgrid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def filter(self, pts, intensity_col=3):
        """pts: (n, >=3) float32 rows x,y,z[,intensity].  -> (m, 4) float32 centroids x,y,z,intensity."""
        p = np.ascontiguousarray(pts, dtype=np.float32)
        n = p.shape[0]
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        cnt = C.c_int64()
        ioff = 4 * intensity_col if (intensity_col is not None and p.shape[1] > intensity_col) else -1
        check(self._L.rsx_voxelgrid_filter(self._h, p.ctypes.data, n, p.shape[1] * 4, ioff, self.leaf, out.ctypes.data,
                                           out.shape[0], C.byref(cnt)))
        return out[:cnt.value].copy()
