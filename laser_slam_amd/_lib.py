"""ctypes loader for liblsgpu_icp.so (C ABI: include/lsgpu_icp.h).

There is no CPU fallback: if the shared library is missing or a HIP call fails the error is raised
to the caller (the reference's own fallback -- keep the odometry guess on ConvergenceError,
laser_slam/src/laser_track.cpp:499-502 -- lives in the LaserTrack mirror, not here).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LSGPU_SO") or os.path.join(_HERE, "liblsgpu_icp.so")   # (LSGPU_SO: another build of the same library)

OK, NO_CONVERGENCE, BAD_CONFIG, HIP_ERROR, BAD_ARG = 0, 1, 2, 3, 4

# every symbol include/lsgpu_icp.h declares (tests/test_abi.py checks the export table against this)
ABI_SYMBOLS = [
    "lsgpu_icp_config_yaml", "lsgpu_icp_config_default", "lsgpu_icp_create", "lsgpu_icp_destroy",
    "lsgpu_icp_set_reference", "lsgpu_icp_align", "lsgpu_icp_align_batch", "lsgpu_icp_get_trace",
    "lsgpu_chain_config_yaml", "lsgpu_chain_config_default", "lsgpu_icp_filter_reference",
    "lsgpu_icp_filter_reading", "lsgpu_icp_compute", "lsgpu_cloud_upload", "lsgpu_cloud_release",
    "lsgpu_cloud_size", "lsgpu_icp_compute_clouds", "lsgpu_icp_compute_clouds_upload", "lsgpu_filter_cylinder", "lsgpu_filter_voxel_grid",
    "lsgpu_icp_get_reference_mean", "lsgpu_icp_get_info", "lsgpu_icp_get_policy_info", "lsgpu_comm_get_unique_id", "lsgpu_icp_comm_init", "lsgpu_knn", "lsgpu_trim_limit", "lsgpu_normal_eq",
    "lsgpu_transform_points", "lsgpu_rotate_descriptors", "lsgpu_filter_random_sampling",
    "lsgpu_filter_sampling_surface_normal", "lsgpu_check_rigid", "lsgpu_correct_rigid", "lsgpu_rotation_distance",
    "lsgpu_strerror", "lsgpu_last_error", "lsgpu_abi_version", "lsgpu_apply_point_filters",
    "lsgpu_cloud_from_pointcloud2", "lsgpu_cloud_to_pointxyz",
]


class ChainCfg(C.Structure):
    _fields_ = [
        ("reading_prob", C.c_float),
        ("ssn_knn", C.c_int),
        ("ssn_ratio", C.c_float),
        ("pad_", C.c_int),
        ("seed", C.c_int64),
    ]


class IcpConfig(C.Structure):
    _fields_ = [
        ("trim_ratio", C.c_float),
        ("max_iterations", C.c_int),
        ("min_diff_rot", C.c_float),
        ("min_diff_trans", C.c_float),
        ("smooth_length", C.c_int),
        ("cell_size", C.c_float),
        ("profile_kernels", C.c_int),
        ("reserved", C.c_int * 8),
    ]


class IcpStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("final_limit", C.c_float),
        ("final_n_used", C.c_int64),
        ("stragglers", C.c_int64),
        ("t_total_ms", C.c_double),
        ("t_knn_ms", C.c_double),
        ("knn_launches", C.c_int),
        ("t_knn_main_ms", C.c_double),
        ("t_knn_fallback_ms", C.c_double),
        ("cap_retries", C.c_int),
        ("pad_", C.c_int),
        ("t_reserved", C.c_double * 1),
        ("t_select_ms", C.c_double),
        ("t_ne_ms", C.c_double),
        ("committed_select_iterations", C.c_int),
        ("spread_tiles", C.c_int),
        ("reference_reused", C.c_int),
        ("comm_calls", C.c_int),
        ("t_comm_ms", C.c_double),
        ("direction_index_launches", C.c_int),
        ("direction_index_occupancy", C.c_float),
        ("direction_index_heavy_share", C.c_float),
    ]


class PolicyInfo(C.Structure):
    """lsgpu_policy_info: what the handle's launch policy remembers across calls."""
    _fields_ = [("index_rest", C.c_int), ("pay_voxel_us", C.c_float), ("pay_index_us", C.c_float),
                ("ssn_sort_fallbacks", C.c_int), ("ssn_calls", C.c_int), ("reserved", C.c_int * 3)]


class IterTrace(C.Structure):
    _fields_ = [
        ("T_iter", C.c_float * 16),
        ("limit", C.c_float),
        ("n_used", C.c_int64),
        ("A", C.c_double * 36),
        ("b", C.c_double * 6),
        ("x", C.c_double * 6),
        ("knn_main_us", C.c_float),
        ("knn_fallback_us", C.c_float),
        ("stragglers", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class IcpInfo(C.Structure):
    _fields_ = [
        ("n_reference", C.c_int64),
        ("bits_per_axis", C.c_int),
        ("fine_bits", C.c_int),
        ("cell_size", C.c_float),
        ("n_chunks", C.c_uint32),
        ("cells", C.c_uint32 * 17),
        ("table_bytes", C.c_uint64),
    ]


class PointFilter(C.Structure):
    """lsgpu_point_filter (include/lsgpu_icp.h): one module of the input filter chain."""
    _fields_ = [("type", C.c_int), ("dim", C.c_int), ("flag", C.c_int), ("pad_", C.c_int), ("v", C.c_float * 6),
                ("state", C.c_double)]


FILTER_MAX_DIST, FILTER_MIN_DIST, FILTER_BOUNDING_BOX, FILTER_FIX_STEP_SAMPLING, FILTER_RANDOM_SAMPLING, FILTER_REMOVE_NAN = 1, 2, 3, 4, 5, 6


class LsgpuError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        super().__init__(f"{what}: {strerror(code)}" + (f" [{detail}]" if detail else ""))


class ConvergenceError(LsgpuError):
    """PointMatcher::ConvergenceError equivalent (LSGPU_NO_CONVERGENCE)."""


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C laser_slam_amd/csrc`.  There is no CPU fallback for the ICP hot path.")
    L = C.CDLL(SO_PATH)
    vp, i64, fp = C.c_void_p, C.c_int64, C.c_void_p  # data pointers passed as raw addresses
    L.lsgpu_icp_config_yaml.argtypes = [C.POINTER(IcpConfig)]
    L.lsgpu_icp_config_yaml.restype = None
    L.lsgpu_icp_config_default.argtypes = [C.POINTER(IcpConfig)]
    L.lsgpu_icp_config_default.restype = None
    L.lsgpu_icp_create.argtypes = [C.POINTER(IcpConfig), C.c_int, C.POINTER(vp)]
    L.lsgpu_icp_destroy.argtypes = [vp]
    L.lsgpu_icp_destroy.restype = None
    L.lsgpu_icp_set_reference.argtypes = [vp, fp, fp, i64]
    L.lsgpu_icp_align.argtypes = [vp, fp, i64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.POINTER(IcpStats)]
    L.lsgpu_icp_align_batch.argtypes = [C.POINTER(vp), C.c_int, i64, C.POINTER(fp), C.POINTER(fp),
                                        C.POINTER(i64), C.POINTER(fp), C.POINTER(i64),
                                        C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(IcpStats), C.POINTER(C.c_int)]
    L.lsgpu_chain_config_yaml.argtypes = [C.POINTER(ChainCfg)]
    L.lsgpu_chain_config_yaml.restype = None
    L.lsgpu_chain_config_default.argtypes = [C.POINTER(ChainCfg)]
    L.lsgpu_chain_config_default.restype = None
    L.lsgpu_icp_filter_reference.argtypes = [vp, fp, i64, C.c_int, C.c_float, i64, fp, fp, C.POINTER(i64)]
    L.lsgpu_icp_filter_reading.argtypes = [vp, fp, i64, C.c_float, i64, fp, C.POINTER(i64)]
    L.lsgpu_icp_compute.argtypes = [vp, fp, i64, fp, i64, C.POINTER(C.c_float), C.POINTER(ChainCfg),
                                    C.POINTER(C.c_float), C.POINTER(IcpStats)]
    L.lsgpu_filter_cylinder.argtypes = [vp, fp, i64, C.POINTER(C.c_float), C.c_double, C.c_double, C.c_int, fp,
                                        C.POINTER(i64)]
    L.lsgpu_filter_voxel_grid.argtypes = [vp, fp, i64, C.POINTER(C.c_float), C.c_int, fp, C.POINTER(i64)]
    L.lsgpu_apply_point_filters.argtypes = [vp, C.POINTER(PointFilter), C.c_int, fp, i64, i64, fp, C.POINTER(i64)]
    L.lsgpu_cloud_from_pointcloud2.argtypes = [vp, fp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp,
                                               C.POINTER(i64)]
    L.lsgpu_cloud_to_pointxyz.argtypes = [vp, fp, i64, fp]
    L.lsgpu_cloud_upload.argtypes = [vp, C.c_int, fp, i64]
    L.lsgpu_cloud_release.argtypes = [vp, C.c_int]
    L.lsgpu_cloud_size.argtypes = [vp, C.c_int, C.POINTER(i64)]
    L.lsgpu_icp_compute_clouds.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                           C.POINTER(C.c_float), C.POINTER(ChainCfg), C.POINTER(C.c_float),
                                           C.POINTER(IcpStats)]
    L.lsgpu_icp_compute_clouds_upload.argtypes = [vp, C.c_int, fp, i64, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                                  C.POINTER(C.c_float), C.POINTER(ChainCfg), C.POINTER(C.c_float),
                                                  C.POINTER(IcpStats)]
    L.lsgpu_icp_get_trace.argtypes = [vp, C.POINTER(IterTrace), C.c_int]
    L.lsgpu_icp_get_reference_mean.argtypes = [vp, C.POINTER(C.c_float)]
    L.lsgpu_icp_get_info.argtypes = [vp, C.POINTER(IcpInfo)]
    L.lsgpu_icp_get_policy_info.argtypes = [vp, C.POINTER(PolicyInfo)]
    L.lsgpu_icp_get_policy_info.restype = C.c_int
    L.lsgpu_comm_get_unique_id.argtypes = [C.c_char_p]
    L.lsgpu_icp_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.lsgpu_knn.argtypes = [vp, fp, i64, C.POINTER(C.c_float), fp, fp]
    L.lsgpu_trim_limit.argtypes = [vp, fp, i64, C.c_float, C.POINTER(C.c_float)]
    L.lsgpu_normal_eq.argtypes = [vp, fp, i64, C.POINTER(C.c_float), fp, fp, C.c_float,
                                  C.POINTER(C.c_double)]
    L.lsgpu_transform_points.argtypes = [vp, C.POINTER(C.c_float), fp, i64, fp]
    L.lsgpu_rotate_descriptors.argtypes = [vp, C.POINTER(C.c_float), fp, i64, fp]
    L.lsgpu_filter_random_sampling.argtypes = [i64, C.c_float, i64, C.POINTER(C.c_int64)]
    L.lsgpu_filter_random_sampling.restype = i64
    L.lsgpu_filter_sampling_surface_normal.argtypes = [fp, i64, C.c_int, C.c_float, i64, fp, fp]
    L.lsgpu_filter_sampling_surface_normal.restype = i64
    L.lsgpu_check_rigid.argtypes = [C.POINTER(C.c_float)]
    L.lsgpu_rotation_distance.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lsgpu_rotation_distance.restype = C.c_float
    L.lsgpu_correct_rigid.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lsgpu_correct_rigid.restype = None
    L.lsgpu_strerror.argtypes = [C.c_int]
    L.lsgpu_strerror.restype = C.c_char_p
    L.lsgpu_last_error.argtypes = [vp]
    L.lsgpu_last_error.restype = C.c_char_p
    L.lsgpu_abi_version.restype = C.c_int
    _lib = L
    return L


def strerror(code: int) -> str:
    return lib().lsgpu_strerror(code).decode()
