"""ctypes binding of librsx.so (include/rsx.h).  Thin: argument marshalling and status -> exception.

The library is the product; this module never computes anything itself and there is no CPU
fallback: if librsx.so is missing or no GPU is visible, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSX_LIB_PATH") or os.path.join(_HERE, "librsx.so")  # override: kernel experiments

NUM_RING, NUM_SECTOR, DESC_SIZE, MAX_TOPK = 20, 60, 1200, 32
MODE_CANDIDATE, MODE_EXHAUSTIVE = 0, 1
FILTER_AUTO, FILTER_OFF, FILTER_FORCE, FILTER_Q1 = 0, 1, 2, 3   # rsx_sc_params.filter_mode
KIND_AUTO, KIND_DIRECT, KIND_SPECTRAL, KIND_SPECTRAL2 = 0, 1, 2, 3
SUM_EIGEN_SSE2, SUM_SEQ, SUM_EIGEN_AVX_FMA, SUM_EIGEN34_AVX_FMA = 0, 1, 2, 3   # rsx_sc_params.sum_order (how the reference was built)
default_filter_kind = KIND_AUTO   # what SCManager(filter_kind=KIND_AUTO) passes on (tests flip it to cover both forms)

HIT_DTYPE = np.dtype([("dist", "<f8"), ("index", "<i4"), ("shift", "<i4")])


class RsxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rsx status {status}: {msg}")
        self.status = status


class ScParams(C.Structure):
    _fields_ = [
        ("lidar_height", C.c_double),
        ("max_radius", C.c_double),
        ("num_exclude_recent", C.c_int32),
        ("num_candidates", C.c_int32),
        ("search_ratio", C.c_double),
        ("dist_thres", C.c_double),
        ("tree_making_period", C.c_int32),
        ("device", C.c_int32),
        ("shard_rank", C.c_int32),
        ("shard_world", C.c_int32),
        ("capacity_hint", C.c_int64),
        ("filter_mode", C.c_int32),
        ("filter_kind", C.c_int32),
        ("sum_order", C.c_int32),
    ]


class RescoringStats(C.Structure):  # rsx_sc_rescoring_stats (include/rsx_diag.h)
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("candidates", C.c_int64), ("exact_evals", C.c_int64),
                ("queries_rescored", C.c_int64), ("window_previews", C.c_int64), ("valu_previews", C.c_int64),
                ("exact_window_shifts", C.c_int64)]


ORORA_PMC = 4  # rsx_orora_params.flags: max-clique inlier selection before the solver
ORORA_PMC_PROVEN, ORORA_PMC_PASSTHROUGH, ORORA_PMC_NO_WORKSPACE = 1, 2, 4
PMC_INFO_DTYPE = np.dtype([("size", "<i4"), ("max_core", "<i4"), ("seeds", "<i4"), ("flags", "<i4")])


class ScDetection(C.Structure):
    _fields_ = [("loop_id", C.c_int32), ("yaw_diff_rad", C.c_float), ("min_dist", C.c_double), ("nn_idx", C.c_int32),
                ("query_idx", C.c_int32), ("searched", C.c_int32), ("reserved", C.c_int32), ("dist_thres", C.c_double)]


class OroraParams(C.Structure):
    _fields_ = [("tim_noise_bound", C.c_double), ("noise_bound_radial", C.c_double),
                ("noise_bound_tangential", C.c_double), ("gnc_factor", C.c_double),
                ("cost_threshold", C.c_double), ("max_iterations", C.c_int32), ("flags", C.c_int32)]


class IcpParams(C.Structure):
    _fields_ = [("max_corr_dist", C.c_double), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double), ("max_iterations", C.c_int32), ("flags", C.c_int32)]


class IcpResult(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("fitness", C.c_double), ("iterations", C.c_int32),
                ("converged", C.c_int32), ("state", C.c_int32), ("reserved", C.c_int32)]


class LoopVerifyParams(C.Structure):
    _fields_ = [("history_keyframe_search_num", C.c_int32), ("leaf", C.c_float), ("fitness_threshold", C.c_double), ("icp", IcpParams)]


class LoopVerifyResult(C.Structure):
    _fields_ = [("accepted", C.c_int32), ("converged", C.c_int32), ("iterations", C.c_int32), ("state", C.c_int32),
                ("fitness", C.c_double), ("transform", C.c_float * 16),
                ("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("roll", C.c_float), ("pitch", C.c_float), ("yaw", C.c_float),
                ("relative", C.c_double * 16), ("n_source", C.c_int64), ("n_target", C.c_int64)]


class FrontendParams(C.Structure):
    _fields_ = [("cart_pixel_width", C.c_int32), ("cart_resolution", C.c_float), ("ratio", C.c_float), ("flags", C.c_int32)]


class Cen2019Params(C.Structure):
    _fields_ = [("max_points", C.c_int32), ("min_range", C.c_int32)]


class OroraResult(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("yaw", C.c_double), ("iterations", C.c_int32), ("rot_inliers", C.c_int32),
                ("trans_inliers", C.c_int32), ("status", C.c_int32)]


class OdometryParams(C.Structure):
    _fields_ = [("cen", Cen2019Params), ("frontend", FrontendParams), ("orora", OroraParams), ("radar_resolution", C.c_float),
                ("col_offset", C.c_int32), ("max_keypoints", C.c_int32), ("device", C.c_int32)]


ODOMETRY_SCAN_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("yaw", "<f8"), ("iterations", "<i4"), ("rot_inliers", "<i4"),
                                ("trans_inliers", "<i4"), ("status", "<i4"), ("n_keypoints", "<i4"), ("n_matches", "<i4")])


ORORA_RESULT_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("yaw", "<f8"), ("iterations", "<i4"),
                               ("rot_inliers", "<i4"), ("trans_inliers", "<i4"), ("status", "<i4")])

_lib = None

# every symbol include/rsx.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "rsx_last_error_string", "rsx_version", "rsx_device_count", "rsx_selftest_firewall",
    "rsx_sc_filter_range_device", "rsx_sc_query_bounds_device",
    "rsx_sc_default_params", "rsx_sc_create", "rsx_sc_destroy", "rsx_sc_set_dist_thres", "rsx_sc_size",
    "rsx_sc_local_size", "rsx_sc_add_points", "rsx_sc_add_descriptor", "rsx_sc_add_descriptors_f32",
    "rsx_sc_add_descriptors_f32_device", "rsx_sc_add_descriptor_rounded", "rsx_sc_export_descriptors_f32",
    "rsx_sc_save", "rsx_sc_load", "rsx_sc_get_descriptor", "rsx_sc_get_ringkey",
    "rsx_sc_get_sectorkey", "rsx_sc_detect_loop_closure", "rsx_sc_detect_loop_closure_ex", "rsx_sc_make_scancontext",
    "rsx_sc_make_keys", "rsx_sc_dist_direct", "rsx_sc_fast_align", "rsx_sc_distance", "rsx_sc_detect_between_session",
    "rsx_sc_tree_size", "rsx_sc_ringkey_tree_layout", "rsx_sc_query", "rsx_sc_query_device", "rsx_sc_query_stage1_device",
    "rsx_sc_query_stage1_elig_device",
    "rsx_sc_query_stage2_device", "rsx_sc_query_self_device",
    "rsx_sc_pair_distances", "rsx_sc_filter_bounds", "rsx_sc_filter_eps", "rsx_sc_profiled_kernel_name",
    "rsx_sc_merge_topk", "rsx_sc_merge_topk_device", "rsx_sc_hit_to_loop",
    "rsx_sc_dominant_kernel_name", "rsx_sc_profile_enable", "rsx_sc_profile_read", "rsx_sc_profile_read_rescoring", "rsx_sc_window_previews",
    "rsx_scs_create", "rsx_scs_create_layout", "rsx_scs_num_query_groups", "rsx_scs_destroy", "rsx_scs_num_shards", "rsx_scs_set_dist_thres", "rsx_scs_size",
    "rsx_scs_add_points", "rsx_scs_add_descriptors_f32", "rsx_scs_get_descriptor", "rsx_scs_query",
    "rsx_scs_detect_loop_closure",
    "rsx_orora_default_params", "rsx_orora_max_correspondences", "rsx_orora_create", "rsx_orora_destroy",
    "rsx_orora_register_batch", "rsx_orora_register_batch_device", "rsx_orora_max_clique_matches", "rsx_orora_reserve",
    "rsx_orora_max_clique_batch", "rsx_orora_max_clique_batch_device", "rsx_orora_last_pmc_info",
    "rsx_cen2019_default_params", "rsx_cen2019_create", "rsx_cen2019_destroy", "rsx_cen2019_extract",
    "rsx_cen2019_extract_batch", "rsx_cen2019_extract_batch_device",
    "rsx_frontend_default_params", "rsx_frontend_create", "rsx_frontend_destroy", "rsx_frontend_cartesian",
    "rsx_frontend_describe", "rsx_frontend_match",
    "rsx_frontend_cartesian_batch_device", "rsx_frontend_cartesian_batch_device_az", "rsx_frontend_describe_batch_device", "rsx_frontend_match_consecutive_device",
    "rsx_frontend_read_images",
    "rsx_odometry_default_params", "rsx_odometry_create", "rsx_odometry_destroy", "rsx_odometry_reset", "rsx_odometry_window",
    "rsx_odometry_push", "rsx_odometry_push_device", "rsx_host_alloc_pinned", "rsx_host_free_pinned",
    "rsx_voxelgrid_create", "rsx_voxelgrid_destroy", "rsx_voxelgrid_filter", "rsx_sc_add_points_downsampled",
    "rsx_icp_default_params", "rsx_icp_create", "rsx_icp_destroy", "rsx_icp_align",
    "rsx_kfstore_create", "rsx_kfstore_destroy", "rsx_kfstore_add", "rsx_kfstore_add_device", "rsx_kfstore_size", "rsx_kfstore_get",
    "rsx_loop_verify_default_params", "rsx_loop_submap", "rsx_loop_verify", "rsx_kfstore_build_map", "rsx_sc_add_keyframe",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsxError(-7, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 (same SONAME
        # as /opt/rocm's).  If librsx pulled in the system runtime first, a later `import torch`
        # would load a second runtime and find no GPU; loading torch first makes librsx bind to
        # the runtime torch already loaded.  (A C++/ROS host has only the system runtime.)
        if os.environ.get("RSX_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        L.rsx_last_error_string.restype = C.c_char_p
        L.rsx_version.restype = C.c_char_p
        L.rsx_sc_dominant_kernel_name.restype = C.c_char_p
        L.rsx_sc_profiled_kernel_name.restype = C.c_char_p
        L.rsx_sc_profiled_kernel_name.argtypes = [C.c_void_p]
        L.rsx_sc_filter_eps.restype = C.c_double
        L.rsx_sc_filter_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.rsx_sc_default_params.argtypes = [C.POINTER(ScParams)]
        L.rsx_sc_create.argtypes = [C.POINTER(ScParams), C.POINTER(vp)]
        L.rsx_sc_destroy.argtypes = [vp]
        L.rsx_sc_set_dist_thres.argtypes = [vp, dbl]
        L.rsx_sc_size.argtypes = [vp, C.POINTER(i64)]
        L.rsx_sc_local_size.argtypes = [vp, C.POINTER(i64)]
        L.rsx_sc_tree_size.argtypes = [vp, C.POINTER(i64)]
        L.rsx_sc_ringkey_tree_layout.argtypes = [vp, i64, vp, vp, vp]
        L.rsx_sc_add_points.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(i32)]
        L.rsx_sc_add_descriptor.argtypes = [vp, vp, C.POINTER(i32)]
        L.rsx_sc_add_descriptors_f32.argtypes = [vp, vp, i64]
        L.rsx_sc_add_descriptors_f32_device.argtypes = [vp, vp, i64, vp]
        L.rsx_sc_add_descriptor_rounded.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(dbl)]
        L.rsx_sc_export_descriptors_f32.argtypes = [vp, i64, i64, vp]
        L.rsx_sc_save.argtypes = [vp, C.c_char_p]
        L.rsx_sc_load.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
        L.rsx_sc_get_descriptor.argtypes = [vp, i64, vp]
        L.rsx_sc_get_ringkey.argtypes = [vp, i64, vp]
        L.rsx_sc_get_sectorkey.argtypes = [vp, i64, vp]
        L.rsx_sc_detect_loop_closure.argtypes = [vp, C.c_int, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(dbl), C.POINTER(i32)]
        L.rsx_sc_detect_loop_closure_ex.argtypes = [vp, C.c_int, C.POINTER(ScDetection)]
        L.rsx_sc_make_scancontext.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
        L.rsx_sc_make_keys.argtypes = [vp, vp, vp, vp]
        L.rsx_sc_dist_direct.argtypes = [vp, vp, vp, C.POINTER(dbl)]
        L.rsx_sc_fast_align.argtypes = [vp, vp, vp, C.POINTER(i32)]
        L.rsx_sc_distance.argtypes = [vp, vp, vp, C.POINTER(dbl), C.POINTER(i32)]
        L.rsx_sc_detect_between_session.argtypes = [vp, vp, vp, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(dbl), C.POINTER(i32)]
        L.rsx_sc_query.argtypes = [vp, vp, i32, i32, i64, vp]
        L.rsx_sc_query_device.argtypes = [vp, vp, i32, i32, i64, vp, vp]
        L.rsx_sc_query_stage1_device.argtypes = [vp, vp, i32, i32, i64, vp, vp]
        L.rsx_sc_filter_range_device.argtypes = [vp, vp, i32, i64, i64, vp, i64, vp]
        L.rsx_sc_query_bounds_device.argtypes = [vp, vp, i32, i32, i64, vp, i32, i64, i64, vp, vp]
        L.rsx_sc_query_stage1_elig_device.argtypes = [vp, vp, i32, i32, i64, vp, i32, vp, vp]
        L.rsx_sc_query_stage2_device.argtypes = [vp, i32, i32, vp, vp, vp]
        L.rsx_sc_query_self_device.argtypes = [vp, i64, i32, i32, i64, i32, vp, vp]
        L.rsx_sc_pair_distances.argtypes = [vp, vp, i64, i64, vp, vp]
        L.rsx_sc_merge_topk.argtypes = [vp, i32, i32, i32, vp]
        L.rsx_sc_merge_topk_device.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.rsx_sc_profile_enable.argtypes = [vp, C.c_int]
        L.rsx_sc_profile_read.argtypes = [vp, C.POINTER(i64), C.POINTER(dbl)]
        L.rsx_sc_profile_read_rescoring.argtypes = [vp, C.POINTER(RescoringStats)]
        L.rsx_sc_window_previews.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp]
        L.rsx_sc_hit_to_loop.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(C.c_float)]
        L.rsx_scs_create.argtypes = [C.POINTER(ScParams), C.POINTER(i32), i32, C.POINTER(vp)]
        L.rsx_scs_create_layout.argtypes = [C.POINTER(ScParams), C.POINTER(i32), i32, i32, i32, C.POINTER(vp)]
        L.rsx_scs_num_query_groups.argtypes = [vp]
        L.rsx_scs_destroy.argtypes = [vp]
        L.rsx_scs_num_shards.argtypes = [vp]
        L.rsx_scs_set_dist_thres.argtypes = [vp, dbl]
        L.rsx_scs_size.argtypes = [vp, C.POINTER(i64)]
        L.rsx_scs_add_points.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(i32)]
        L.rsx_scs_add_descriptors_f32.argtypes = [vp, vp, i64]
        L.rsx_scs_get_descriptor.argtypes = [vp, i64, vp]
        L.rsx_scs_query.argtypes = [vp, vp, i32, i32, i64, vp]
        L.rsx_scs_detect_loop_closure.argtypes = [vp, C.POINTER(ScDetection)]
        L.rsx_orora_default_params.argtypes = [C.POINTER(OroraParams)]
        L.rsx_orora_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rsx_orora_destroy.argtypes = [vp]
        L.rsx_orora_register_batch.argtypes = [vp, vp, vp, vp, i32, C.POINTER(OroraParams), vp]
        L.rsx_orora_register_batch_device.argtypes = [vp, vp, vp, vp, i32, C.POINTER(OroraParams), vp, vp]
        L.rsx_orora_reserve.argtypes = [vp, i64]
        L.rsx_orora_max_clique_batch.argtypes = [vp, vp, vp, vp, i32, C.POINTER(OroraParams), vp, vp]
        L.rsx_orora_max_clique_batch_device.argtypes = [vp, vp, vp, vp, i32, C.POINTER(OroraParams), vp, vp, vp]
        L.rsx_orora_last_pmc_info.argtypes = [vp, vp, i32]
        L.rsx_cen2019_default_params.argtypes = [C.POINTER(Cen2019Params)]
        L.rsx_cen2019_create.argtypes = [C.c_int, i32, i32, C.POINTER(vp)]
        L.rsx_cen2019_destroy.argtypes = [vp]
        L.rsx_cen2019_extract.argtypes = [vp, vp, i32, i32, C.POINTER(Cen2019Params), vp, C.c_float, vp, vp, i32,
                                          C.POINTER(i32)]
        L.rsx_cen2019_extract_batch.argtypes = [vp, vp, i32, i64, i32, i32, C.POINTER(Cen2019Params), vp, i32, C.c_float, vp, vp, i32, vp]
        L.rsx_cen2019_extract_batch_device.argtypes = [vp, vp, i32, i64, i32, i32, C.POINTER(Cen2019Params), vp, i32, C.c_float, vp, vp,
                                                       i32, vp, vp]
        L.rsx_frontend_default_params.argtypes = [C.POINTER(FrontendParams)]
        L.rsx_frontend_create.argtypes = [C.c_int, i32, i32, C.POINTER(FrontendParams), C.POINTER(vp)]
        L.rsx_frontend_destroy.argtypes = [vp]
        L.rsx_frontend_cartesian.argtypes = [vp, vp, i32, i32, vp, C.c_float, vp]
        L.rsx_frontend_describe.argtypes = [vp, vp, i32, vp, vp]
        L.rsx_frontend_match.argtypes = [vp, vp, vp, i32, vp, vp, i32, C.c_float, vp, vp, vp]
        L.rsx_frontend_cartesian_batch_device.argtypes = [vp, vp, i32, i64, i32, i32, vp, C.c_float, vp]
        L.rsx_frontend_cartesian_batch_device_az.argtypes = [vp, vp, i32, i64, i32, i32, vp, i64, C.c_float, vp]
        L.rsx_frontend_read_images.argtypes = [vp, i32, vp, vp]
        L.rsx_frontend_describe_batch_device.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
        L.rsx_frontend_match_consecutive_device.argtypes = [vp, vp, vp, vp, i32, i32, i32, C.c_float, vp, vp, vp]
        L.rsx_odometry_default_params.argtypes = [C.POINTER(OdometryParams)]
        L.rsx_odometry_create.argtypes = [C.POINTER(OdometryParams), i32, i32, C.POINTER(vp)]
        L.rsx_odometry_destroy.argtypes = [vp]
        L.rsx_odometry_reset.argtypes = [vp]
        L.rsx_odometry_push.argtypes = [vp, vp, i32, i64, i32, vp, i32, vp, vp, i32]
        L.rsx_odometry_push_device.argtypes = [vp, vp, i32, i64, i32, vp, i32, vp, vp, i32]
        L.rsx_host_alloc_pinned.argtypes = [C.c_size_t, C.POINTER(vp)]
        L.rsx_host_free_pinned.argtypes = [vp]
        L.rsx_voxelgrid_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rsx_voxelgrid_destroy.argtypes = [vp]
        L.rsx_voxelgrid_filter.argtypes = [vp, vp, C.c_size_t, C.c_size_t, i32, C.c_float, vp, i64, C.POINTER(i64)]
        L.rsx_sc_add_points_downsampled.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_float, C.POINTER(i32)]
        L.rsx_icp_default_params.argtypes = [C.POINTER(IcpParams)]
        L.rsx_icp_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rsx_icp_destroy.argtypes = [vp]
        L.rsx_icp_align.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.POINTER(IcpParams), vp,
                                    C.POINTER(IcpResult)]
        L.rsx_kfstore_create.argtypes = [i32, C.POINTER(vp)]
        L.rsx_kfstore_destroy.argtypes = [vp]
        L.rsx_kfstore_add.argtypes = [vp, vp, C.c_size_t, C.c_size_t, i32, C.POINTER(i32)]
        L.rsx_kfstore_add_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(i32)]
        L.rsx_kfstore_size.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
        L.rsx_kfstore_get.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
        L.rsx_loop_verify_default_params.argtypes = [C.POINTER(LoopVerifyParams)]
        L.rsx_loop_submap.argtypes = [vp, i32, i32, vp, C.c_float, vp, i64, C.POINTER(i64)]
        L.rsx_loop_verify.argtypes = [vp, i32, i32, vp, C.POINTER(LoopVerifyParams), C.POINTER(LoopVerifyResult)]
        L.rsx_kfstore_build_map.argtypes = [vp, vp, i64, i32, C.c_float, vp, i64, C.POINTER(i64)]
        L.rsx_sc_add_keyframe.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, i32, C.c_float, C.POINTER(i32)]
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise RsxError(status, lib().rsx_last_error_string().decode())


def device_count():
    return lib().rsx_device_count()


def version():
    return lib().rsx_version().decode()


class PinnedArray:
    """A numpy array over page-locked host memory (rsx_host_alloc_pinned): what a caller hands the host-buffer entries
    (rsx_sc_query, rsx_odometry_push, rsx_cen2019_extract ...) so that their uploads run asynchronously at full PCIe rate.
    `.a` is the array; close() (or the context manager) frees the memory -- the array must not be used after that."""

    def __init__(self, shape, dtype):
        import numpy as np
        self._np = np
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        self._p = C.c_void_p()
        check(lib().rsx_host_alloc_pinned(max(n, 1), C.byref(self._p)))
        buf = (C.c_char * max(n, 1)).from_address(self._p.value)
        self.a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._p:
            self.a = None
            check(lib().rsx_host_free_pinned(self._p))
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
