"""Host-side wrapper of the keyframe-cloud store of librsx.so (include/rsx.h): what the reference's pose-graph node does with
keyframeLaserClouds -- loopFindNearKeyframesCloud / doICPVirtualRelative / pubMap (laserPosegraphOptimization.cpp:329-406,
631-655).  Test / bench harness only: the product is the C-ABI."""
import ctypes as C

import numpy as np

from ._rsx import LoopVerifyParams, LoopVerifyResult, check, lib


class KeyframeStore:
    def __init__(self, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.rsx_kfstore_create(device, C.byref(self._h)))
        self.params = LoopVerifyParams()
        check(self._L.rsx_loop_verify_default_params(C.byref(self.params)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rsx_kfstore_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def add(self, cloud):
        """cloud: (n, 4) float32 x, y, z, intensity (or (n, 3): intensity 0) -> keyframe index"""
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        idx = C.c_int32(-1)
        check(self._L.rsx_kfstore_add(self._h, c.ctypes.data, C.c_size_t(c.shape[0]), C.c_size_t(c.shape[1] * 4),
                                      12 if c.shape[1] >= 4 else -1, C.byref(idx)))
        return idx.value

    def size(self):
        nk, npnt = C.c_int64(), C.c_int64()
        check(self._L.rsx_kfstore_size(self._h, C.byref(nk), C.byref(npnt)))
        return nk.value, npnt.value

    def get(self, index):
        n = C.c_int64()
        check(self._L.rsx_kfstore_get(self._h, index, None, C.c_int64(0), C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        check(self._L.rsx_kfstore_get(self._h, index, out.ctypes.data, C.c_int64(out.shape[0]), C.byref(n)))
        return out[:n.value].copy()

    def submap(self, key, submap_size, root_pose, leaf=0.4):
        rp = np.ascontiguousarray(root_pose, dtype=np.float64).reshape(6)
        _, npnt = self.size()
        out = np.zeros((max(npnt, 1), 4), dtype=np.float32)
        n = C.c_int64()
        check(self._L.rsx_loop_submap(self._h, key, submap_size, rp.ctypes.data, C.c_float(leaf), out.ctypes.data,
                                      C.c_int64(out.shape[0]), C.byref(n)))
        return out[:n.value].copy()

    def verify(self, loop_idx, curr_idx, root_pose):
        """doICPVirtualRelative(loop_idx, curr_idx); root_pose = keyframePosesUpdated[loop_idx] (x, y, z, roll, pitch, yaw)"""
        rp = np.ascontiguousarray(root_pose, dtype=np.float64).reshape(6)
        r = LoopVerifyResult()
        check(self._L.rsx_loop_verify(self._h, loop_idx, curr_idx, rp.ctypes.data, C.byref(self.params), C.byref(r)))
        return {"accepted": bool(r.accepted), "converged": bool(r.converged), "iterations": r.iterations, "state": r.state,
                "fitness": r.fitness, "transform": np.array(r.transform, dtype=np.float32).reshape(4, 4),
                "xyz_rpy": np.array([r.x, r.y, r.z, r.roll, r.pitch, r.yaw], dtype=np.float32),
                "relative": np.array(r.relative, dtype=np.float64).reshape(4, 4), "n_source": r.n_source, "n_target": r.n_target}

    def build_map(self, poses, skip=2, leaf=0.4):
        ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
        _, npnt = self.size()
        out = np.zeros((max(npnt, 1), 4), dtype=np.float32)
        n = C.c_int64()
        check(self._L.rsx_kfstore_build_map(self._h, ps.ctypes.data, C.c_int64(ps.shape[0]), skip, C.c_float(leaf), out.ctypes.data,
                                            C.c_int64(out.shape[0]), C.byref(n)))
        return out[:n.value].copy()
