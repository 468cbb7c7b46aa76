"""ctypes binding of libmi355_lidar.so (include/mi355_lidar.h).

The library is HIP-only: there is no CPU fallback anywhere in this package.  Loading fails loudly when the
shared object has not been built (run ``python -c "import __graft_entry__ as g; g.build()"``), and
``Context()`` fails loudly when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_LIB: another build of the same library (the A/B scripts under benchmarks/ time an older build beside the current one)
LIB_PATH = os.environ.get("MI355_LIB") or os.path.join(_HERE, "libmi355_lidar.so")
_LIB = None

METHOD_FAST_GICP, METHOD_NDT_OMP, METHOD_ICP, METHOD_GICP = 0, 1, 2, 3
REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)
OPT_LM, OPT_GN = 0, 1
OK, EINVAL, EHIP, ECAPACITY, ESTATE = 0, -1, -2, -3, -4


class Mi355Error(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"mi355 error {code}: {text}")
        self.code = code


class CloudView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_size_t), ("stride", C.c_size_t), ("intensity_offset", C.c_size_t)]


class GicpParams(C.Structure):
    _fields_ = [
        ("k_correspondences", C.c_int),
        ("max_iterations", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("rotation_epsilon", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("regularization_method", C.c_int),
        ("optimizer", C.c_int),
        ("lm_max_iterations", C.c_int),
        ("lm_init_lambda_factor", C.c_double),
        ("force_iterations", C.c_int),
        ("num_threads", C.c_int),
    ]


class NdtParams(C.Structure):
    _fields_ = [
        ("transformation_epsilon", C.c_double),
        ("step_size", C.c_double),
        ("resolution", C.c_double),
        ("max_iterations", C.c_int),
        ("outlier_ratio", C.c_double),
        ("min_points_per_voxel", C.c_int),
        ("min_covar_eigvalue_mult", C.c_double),
        ("legacy_mt_skip", C.c_int),
        ("num_threads", C.c_int),
    ]


class GicpBfgsParams(C.Structure):
    _fields_ = [("max_correspondence_distance", C.c_double), ("max_iterations", C.c_int), ("max_optimizer_iterations", C.c_int),
                ("transformation_epsilon", C.c_double), ("rotation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
                ("k_correspondences", C.c_int), ("ransac_iterations", C.c_int)]


class IcpParams(C.Structure):
    _fields_ = [("max_correspondence_distance", C.c_double), ("max_iterations", C.c_int), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double), ("rotation_epsilon", C.c_double), ("ransac_iterations", C.c_int)]


class PrefilterParams(C.Structure):
    _fields_ = [("min_distance", C.c_double), ("leaf_size", C.c_double), ("mean_k", C.c_int), ("stddev_mul", C.c_double)]


PAIR_OK, PAIR_TOO_FEW_POINTS = 0, 1


class PointCloud2(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("point_step", C.c_uint32), ("row_step", C.c_uint32),
                ("off_x", C.c_int32), ("off_y", C.c_int32), ("off_z", C.c_int32), ("off_intensity", C.c_int32), ("intensity_datatype", C.c_uint8),
                ("is_bigendian", C.c_uint8), ("data_size", C.c_size_t)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("total_ms", C.c_double), ("launches", C.c_int), ("reserved", C.c_int)]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16),
        ("converged", C.c_int),
        ("iterations", C.c_int),
        ("error", C.c_double),
        ("H", C.c_double * 36),
        ("trans_probability", C.c_double),
        ("n_source", C.c_int),
        ("n_target", C.c_int),
        ("n_inliers", C.c_int),
        ("status", C.c_int),
        ("evaluations", C.c_int),
        ("reserved", C.c_int),
    ]

    def to_dict(self):
        return {
            "T": np.array(self.T, dtype=np.float32).reshape(4, 4),
            "converged": bool(self.converged),
            "iterations": int(self.iterations),
            "error": float(self.error),
            "H": np.array(self.H, dtype=np.float64).reshape(6, 6),
            "trans_probability": float(self.trans_probability),
            "n_source": int(self.n_source),
            "n_target": int(self.n_target),
            "n_inliers": int(self.n_inliers),
            "status": int(self.status),
            "evaluations": int(self.evaluations),
        }


# every symbol include/mi355_lidar.h declares (tests/test_abi.py checks the library exports each one)
EXPORTS = [
    "mi355_abi_version", "mi355_gicp_default_params", "mi355_ndt_default_params", "mi355_prefilter_default_params", "mi355_icp_default_params", "mi355_gicp_bfgs_default_params",
    "mi355_create", "mi355_destroy", "mi355_last_error", "mi355_set_stream", "mi355_synchronize", "mi355_set_grid_cell", "mi355_device_info",
    "mi355_distance_filter", "mi355_voxel_grid", "mi355_sor", "mi355_prefilter", "mi355_transform",
    "mi355_knn", "mi355_covariances",
    "mi355_reg_create", "mi355_reg_destroy", "mi355_reg_set_params", "mi355_reg_set_target", "mi355_reg_set_source", "mi355_reg_align",
    "mi355_reg_fitness", "mi355_reg_gicp_linearize", "mi355_reg_gicp_trial_cost", "mi355_reg_ndt_derivatives", "mi355_reg_ndt_num_cells", "mi355_reg_ndt_cells",
    "mi355_cloud_upload", "mi355_cloud_destroy", "mi355_cloud_size", "mi355_cloud_has_covariances", "mi355_cloud_download", "mi355_cloud_transform",
    "mi355_cloud_concat", "mi355_cloud_compute_covariances", "mi355_cloud_prefilter", "mi355_reg_set_source_cloud", "mi355_reg_set_target_cloud",
    "mi355_batch_set_targets_clouds", "mi355_batch_set_sources_clouds", "mi355_batch_set_targets_views", "mi355_batch_set_sources_views",
    "mi355_cloud_from_pointcloud2", "mi355_cloud_to_pointcloud2", "mi355_pcd_write", "mi355_cloud_save_pcd",
    "mi355_batch_create", "mi355_batch_destroy", "mi355_batch_set_targets_dev", "mi355_batch_set_sources_dev", "mi355_batch_set_targets",
    "mi355_batch_set_sources", "mi355_batch_set_stream", "mi355_batch_align", "mi355_batch_results", "mi355_batch_set_timing", "mi355_batch_kernel_times", "mi355_batch_set_side_streams", "mi355_batch_set_live", "mi355_batch_verify_correspondences", "mi355_batch_diagnostics", "mi355_batch_get_cloud",
    "mi355_multi_create", "mi355_multi_destroy", "mi355_multi_last_error", "mi355_multi_device_count", "mi355_multi_context",
    "mi355_multi_shard_by_target", "mi355_multi_batch_register", "mi355_multi_batch_register_clouds", "mi355_multi_lane_allocations",
]
# include/mi355_scan_matcher.h
EXPORTS_SCAN_MATCHER = [
    "mi355_scan_matcher_create", "mi355_scan_matcher_destroy", "mi355_scan_matcher_last_error", "mi355_scan_matcher_callback_cloud",
    "mi355_scan_matcher_target",
]
# include/mi355_loop_closure.h
EXPORTS_LOOP_CLOSURE = [
    "mi355_loop_closure_default_params", "mi355_loop_closure_create", "mi355_loop_closure_create_multi", "mi355_loop_closure_destroy", "mi355_loop_closure_last_error",
    "mi355_loop_closure_add_key_frame", "mi355_loop_closure_set_pose", "mi355_loop_closure_candidate", "mi355_loop_closure_close",
]


# mi355_result as a NumPy record (same layout as `Result`): BatchRegistration.results(raw=True) hands the C array over without
# building a dict per pair
RESULT_DTYPE = np.dtype([("T", np.float32, (4, 4)), ("converged", np.int32), ("iterations", np.int32), ("error", np.float64), ("H", np.float64, (6, 6)),
                         ("trans_probability", np.float64), ("n_source", np.int32), ("n_target", np.int32), ("n_inliers", np.int32), ("status", np.int32), ("evaluations", np.int32), ("reserved", np.int32)],
                        align=True)


class LoopClosureParams(C.Structure):
    _fields_ = [
        ("search_key_frame_num", C.c_int),
        ("score_threshold", C.c_double),
        ("search_for_candidate_threshold", C.c_double),
        ("accumulate_distance_threshold", C.c_double),
        ("submap_leaf", C.c_float),
        ("max_iteration", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("correspondence_randomness", C.c_int),
        ("registration_method", C.c_int),
        ("step_size", C.c_double),
        ("ndt_resolution", C.c_double),
        ("correspondence_distance", C.c_double),
        ("max_optimizer_iteration", C.c_int),
        ("ransac_iteration", C.c_int),
    ]


class LoopResult(C.Structure):
    _fields_ = [
        ("query_id", C.c_int),
        ("candidate_id", C.c_int),
        ("converged", C.c_int),
        ("accepted", C.c_int),
        ("iterations", C.c_int),
        ("fitness", C.c_double),
        ("transform", C.c_float * 16),
        ("between", C.c_double * 16),
        ("target_points", C.c_size_t),
    ]


class ScanMatcherParams(C.Structure):
    _fields_ = [
        ("registration_method", C.c_int),
        ("displacement", C.c_double),
        ("max_scan_accumulate_num", C.c_int),
        ("max_iteration", C.c_int),
        ("omp_num_thread", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("step_size", C.c_double),
        ("ndt_resolution", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("correspondence_randomness", C.c_int),
        ("reuse_key_frame_covariances", C.c_int),
        ("correspondence_distance", C.c_double),
        ("max_optimizer_iteration", C.c_int),
    ]


class ScanOutput(C.Structure):
    _fields_ = [
        ("first_scan", C.c_int),
        ("converged", C.c_int),
        ("new_key_frame", C.c_int),
        ("pose", C.c_float * 16),
        ("key_frame_id", C.c_long),
        ("target_points", C.c_size_t),
        ("iterations", C.c_int),
        ("n_key_frames", C.c_long),
        ("accum_distance", C.c_double),
    ]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        L.mi355_create.restype = C.c_void_p
        L.mi355_create.argtypes = [C.c_int]
        L.mi355_destroy.argtypes = [C.c_void_p]
        L.mi355_destroy.restype = None
        L.mi355_last_error.restype = C.c_char_p
        L.mi355_last_error.argtypes = [C.c_void_p]
        L.mi355_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_synchronize.argtypes = [C.c_void_p]
        L.mi355_set_grid_cell.argtypes = [C.c_void_p, C.c_float]
        L.mi355_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        PV = C.POINTER(CloudView)
        L.mi355_distance_filter.argtypes = [C.c_void_p, PV, C.c_double, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi355_voxel_grid.argtypes = [C.c_void_p, PV, C.c_float, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.mi355_sor.argtypes = [C.c_void_p, PV, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p]
        L.mi355_prefilter.argtypes = [C.c_void_p, PV, C.POINTER(PrefilterParams), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi355_transform.argtypes = [C.c_void_p, PV, C.c_void_p, C.c_void_p]
        L.mi355_knn.argtypes = [C.c_void_p, PV, PV, C.c_int, C.c_void_p, C.c_void_p]
        L.mi355_covariances.argtypes = [C.c_void_p, PV, C.c_int, C.c_int, C.c_void_p]
        L.mi355_reg_create.restype = C.c_void_p
        L.mi355_reg_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mi355_reg_destroy.argtypes = [C.c_void_p]
        L.mi355_reg_destroy.restype = None
        L.mi355_reg_set_params.argtypes = [C.c_void_p, C.c_void_p]
        L.mi355_reg_set_target.argtypes = [C.c_void_p, PV]
        L.mi355_reg_set_source.argtypes = [C.c_void_p, PV]
        L.mi355_reg_align.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Result), C.c_void_p]
        L.mi355_reg_fitness.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double)]
        L.mi355_reg_gicp_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        L.mi355_reg_gicp_trial_cost.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mi355_reg_ndt_derivatives.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_reg_ndt_num_cells.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.mi355_reg_ndt_cells.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.mi355_cloud_upload.restype = C.c_void_p
        L.mi355_cloud_upload.argtypes = [C.c_void_p, PV]
        L.mi355_cloud_destroy.argtypes = [C.c_void_p]
        L.mi355_cloud_destroy.restype = None
        L.mi355_cloud_size.restype = C.c_size_t
        L.mi355_cloud_size.argtypes = [C.c_void_p]
        L.mi355_cloud_has_covariances.argtypes = [C.c_void_p]
        L.mi355_cloud_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mi355_cloud_transform.restype = C.c_void_p
        L.mi355_cloud_transform.argtypes = [C.c_void_p, C.c_void_p]
        L.mi355_cloud_concat.restype = C.c_void_p
        L.mi355_cloud_concat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.mi355_cloud_compute_covariances.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mi355_cloud_prefilter.restype = C.c_void_p
        L.mi355_cloud_prefilter.argtypes = [C.c_void_p, C.POINTER(PrefilterParams)]
        L.mi355_reg_set_source_cloud.argtypes = [C.c_void_p, C.c_void_p]
        L.mi355_reg_set_target_cloud.argtypes = [C.c_void_p, C.c_void_p]
        L.mi355_batch_set_targets_clouds.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PrefilterParams)]
        L.mi355_batch_set_sources_clouds.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PrefilterParams)]
        L.mi355_batch_set_targets_views.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PrefilterParams)]
        L.mi355_batch_set_sources_views.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PrefilterParams)]
        L.mi355_cloud_from_pointcloud2.restype = C.c_void_p
        L.mi355_cloud_from_pointcloud2.argtypes = [C.c_void_p, C.POINTER(PointCloud2)]
        L.mi355_cloud_to_pointcloud2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mi355_pcd_write.argtypes = [C.c_char_p, PV, C.c_int]
        L.mi355_cloud_save_pcd.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.mi355_batch_create.restype = C.c_void_p
        L.mi355_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.mi355_batch_destroy.argtypes = [C.c_void_p]
        L.mi355_batch_destroy.restype = None
        for name in ("mi355_batch_set_targets_dev", "mi355_batch_set_sources_dev", "mi355_batch_set_targets", "mi355_batch_set_sources"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PrefilterParams)]
        L.mi355_batch_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_batch_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_batch_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.mi355_batch_set_side_streams.argtypes = [C.c_void_p, C.c_int]
        L.mi355_batch_kernel_times.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]
        L.mi355_loop_closure_default_params.argtypes = [C.POINTER(LoopClosureParams)]
        L.mi355_loop_closure_default_params.restype = None
        L.mi355_loop_closure_create.restype = C.c_void_p
        L.mi355_loop_closure_create.argtypes = [C.c_int, C.POINTER(LoopClosureParams)]
        L.mi355_loop_closure_create_multi.restype = C.c_void_p
        L.mi355_loop_closure_create_multi.argtypes = [C.c_void_p, C.c_int, C.POINTER(LoopClosureParams)]
        L.mi355_batch_set_live.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mi355_batch_verify_correspondences.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.mi355_batch_diagnostics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_batch_get_cloud.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi355_multi_create.restype = C.c_void_p
        L.mi355_multi_create.argtypes = [C.c_void_p, C.c_int]
        L.mi355_multi_destroy.argtypes = [C.c_void_p]
        L.mi355_multi_destroy.restype = None
        L.mi355_multi_last_error.restype = C.c_char_p
        L.mi355_multi_last_error.argtypes = [C.c_void_p]
        L.mi355_multi_device_count.argtypes = [C.c_void_p]
        L.mi355_multi_context.restype = C.c_void_p
        L.mi355_multi_context.argtypes = [C.c_void_p, C.c_int]
        L.mi355_multi_lane_allocations.argtypes = [C.c_void_p, C.c_int]
        L.mi355_multi_shard_by_target.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mi355_multi_batch_register.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.POINTER(PrefilterParams), C.POINTER(PrefilterParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_multi_batch_register_clouds.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                        C.POINTER(PrefilterParams), C.POINTER(PrefilterParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_loop_closure_destroy.argtypes = [C.c_void_p]
        L.mi355_loop_closure_destroy.restype = None
        L.mi355_loop_closure_last_error.restype = C.c_char_p
        L.mi355_loop_closure_last_error.argtypes = [C.c_void_p]
        L.mi355_loop_closure_add_key_frame.argtypes = [C.c_void_p, PV, C.c_void_p, C.c_double]
        L.mi355_loop_closure_set_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mi355_loop_closure_candidate.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.mi355_loop_closure_close.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(LoopResult)]
        L.mi355_scan_matcher_create.restype = C.c_void_p
        L.mi355_scan_matcher_create.argtypes = [C.c_int, C.POINTER(ScanMatcherParams)]
        L.mi355_scan_matcher_destroy.argtypes = [C.c_void_p]
        L.mi355_scan_matcher_destroy.restype = None
        L.mi355_scan_matcher_last_error.restype = C.c_char_p
        L.mi355_scan_matcher_last_error.argtypes = [C.c_void_p]
        L.mi355_scan_matcher_callback_cloud.argtypes = [C.c_void_p, PV, C.c_void_p, C.POINTER(ScanOutput)]
        L.mi355_scan_matcher_target.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB = L
    return _LIB


def gicp_params(**kw) -> GicpParams:
    p = GicpParams()
    lib().mi355_gicp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def ndt_params(**kw) -> NdtParams:
    p = NdtParams()
    lib().mi355_ndt_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def gicp_bfgs_params(**kw) -> GicpBfgsParams:
    p = GicpBfgsParams()
    lib().mi355_gicp_bfgs_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def icp_params(**kw) -> IcpParams:
    p = IcpParams()
    lib().mi355_icp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def prefilter_params(**kw) -> PrefilterParams:
    p = PrefilterParams()
    lib().mi355_prefilter_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def view(a: np.ndarray):
    """(CloudView, keep-alive array) for a float32 [n,4] packed cloud or a [n,8] pcl::PointXYZI-shaped one."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (4, 8):
        raise ValueError(f"cloud must be [n,4] packed or [n,8] PointXYZI-shaped float32, got {a.shape}")
    v = CloudView(a.ctypes.data if a.size else None, a.shape[0], a.shape[1] * 4, 12 if a.shape[1] == 4 else 16)
    return v, a
