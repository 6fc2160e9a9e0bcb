"""Host-side wrapper of the ICP loop-verification entry points of librsx.so (include/rsx.h): the
reference's doICPVirtualRelative (laserPosegraphOptimization.cpp:357-403)."""
import ctypes as C

import numpy as np

from ._rsx import IcpParams, IcpResult, check, lib

LOOP_FITNESS_SCORE_THRESHOLD = 0.3  # laserPosegraphOptimization.cpp:384


class Icp:
    def __init__(self, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.rsx_icp_create(device, C.byref(self._h)))
        self.params = IcpParams()
        check(self._L.rsx_icp_default_params(C.byref(self.params)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_icp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def align(self, source, target, guess=None):
        """source, target: (n, >=3) float32 x,y,z.  -> dict(transform (4,4), fitness, iterations, converged, state)."""
        s = np.ascontiguousarray(source, dtype=np.float32)
        t = np.ascontiguousarray(target, dtype=np.float32)
        g = np.ascontiguousarray(guess, dtype=np.float32).reshape(16) if guess is not None else None
        r = IcpResult()
        check(self._L.rsx_icp_align(self._h, s.ctypes.data, s.shape[0], s.shape[1] * 4, t.ctypes.data, t.shape[0], t.shape[1] * 4,
                                    C.byref(self.params), g.ctypes.data if g is not None else None, C.byref(r)))
        return {"transform": np.array(r.transform, dtype=np.float32).reshape(4, 4), "fitness": r.fitness,
                "iterations": r.iterations, "converged": bool(r.converged), "state": r.state}

    def accepts(self, result):
        """The loop acceptance test of laserPosegraphOptimization.cpp:385."""
        return result["converged"] and result["fitness"] <= LOOP_FITNESS_SCORE_THRESHOLD
