"""ctypes mirror of include/dmsa_hip.h (the C ABI of libdmsa_hip.so).

The structures here are byte-for-byte the PODs declared in the header; the same
classes are reused by the test-only loader of the CPU checker (under oracle/) because the
oracle's C API takes the same problem descriptors.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint32_p = C.POINTER(C.c_uint32)
c_uint64_p = C.POINTER(C.c_uint64)

DMSA_OK = 0
DMSA_ERR_INVALID = -1
DMSA_ERR_NO_DEVICE = -2
DMSA_ERR_HIP = -3
DMSA_ERR_DEPTH = -4
DMSA_ERR_NOMEM = -5

STOP_NUM_ITER, STOP_FEW_GAUSSIANS, STOP_NAN, STOP_NO_IMPROVEMENT, STOP_EPSILON = range(5)

FLAG_POSE_TABLE_HOST = 0x1
FLAG_FIXED_ITERS = 0x2
FLAG_MIRROR_SUMS = 0x4  # accepted and ignored: the reference order is the default
FLAG_STAGE_TIMERS = 0x8


class Settings(C.Structure):
    """dmsa_settings == DmsaOptimSettings (DmsaOptimizer.h:25-39)."""

    _fields_ = [
        ("num_iter", C.c_int32),
        ("epsilon", C.c_double),
        ("use_analytic_jacobi", C.c_int32),
        ("step_length_optim", C.c_double),
        ("max_step", C.c_double),
        ("gauss_split", C.c_int32),
        ("grid_size_1_factor", C.c_float),
        ("grid_size_2_factor", C.c_float),
        ("min_num_points_per_set", C.c_int32),
        ("min_num_gaussians", C.c_int32),
        ("lambda_diag", C.c_float),
        ("use_centralization", C.c_int32),
    ]


class WindowProblem(C.Structure):
    _fields_ = [
        ("num_control_poses", C.c_int32),
        ("rel_orient", c_double_p),
        ("rel_transl", c_double_p),
        ("stamps", c_double_p),
        ("n_total", C.c_int32),
        ("traj_time", c_double_p),
        ("num_points", C.c_int64),
        ("xyz_local", c_float_p),
        ("tform_idx", c_int32_p),
        ("ring_id", c_int32_p),
        ("num_static", C.c_int64),
        ("xyz_static", c_float_p),
        ("ring_id_static", c_int32_p),
        ("min_grid_size", C.c_float),
        ("use_imu", C.c_int32),
        ("dt_res", C.c_double),
        ("balancing_imu", C.c_double),
        ("gravity", C.c_double * 3),
        ("param_indices", c_int32_p),
        ("preint_rot", c_double_p),
        ("preint_pos", c_double_p),
        ("preint_vel", c_double_p),
        ("cov_pvrot_inv", c_double_p),
    ]


class KeyframeProblem(C.Structure):
    _fields_ = [
        ("num_frames", C.c_int32),
        ("rel_orient", c_double_p),
        ("rel_transl", c_double_p),
        ("frame_offset", c_int64_p),
        ("xyz_local", c_float_p),
        ("normal_local", c_float_p),
        ("ring_id", c_int32_p),
        ("min_grid_size", C.c_float),
        ("use_gravity", C.c_int32),
        ("use_odometry", C.c_int32),
        ("gravity", C.c_double * 3),
        ("cov_grav_inv", C.c_double * 9),
        ("balancing_grav", C.c_double),
        ("balancing_odom", C.c_double),
        ("measured_gravity", c_double_p),
        ("gravity_plausible", c_int32_p),
        ("odom_rel_transl", c_double_p),
        ("odom_rel_orient_mat", c_double_p),
        ("odom_transl_cov_inv", C.c_double * 9),
        ("odom_orient_cov_inv", C.c_double * 9),
    ]


class StaticSelectProblem(C.Structure):  # dmsa_static_points.h
    _fields_ = [
        ("num_window", C.c_int64),
        ("window_xyz", c_float_p),
        ("num_keyframes", C.c_int32),
        ("keyframe_ids", c_int32_p),
        ("frame_offset", c_int64_p),
        ("key_xyz", c_float_p),
        ("key_normal", c_float_p),
        ("key_ring", c_int32_p),
        ("cur_pos", C.c_float * 3),
        ("min_grid_size", C.c_float),
    ]


class PreprocessConfig(C.Structure):  # dmsa_preprocess_config
    _fields_ = [("max_num_points_per_scan", C.c_int32), ("min_dist_ds", C.c_float), ("min_dist", C.c_float), ("seed", C.c_uint32),
                ("lidar_to_imu", C.c_float * 16)]


class TrajState(C.Structure):  # dmsa_traj_state (dmsa_window_setup.h)
    _fields_ = [("t0", C.c_double), ("horizon", C.c_double), ("dt_res", C.c_double), ("n_total", C.c_int32), ("num_control_poses", C.c_int32),
                ("stamps", c_double_p), ("traj_time", c_double_p), ("acc_meas", c_double_p), ("ang_vel_meas", c_double_p), ("gravity", C.c_double * 3),
                ("rel_orient", c_double_p), ("rel_transl", c_double_p), ("glob_orient", c_double_p), ("glob_transl", c_double_p)]


class PointCloud2(C.Structure):  # dmsa_pointcloud2 (dmsa_wire_formats.h)
    _fields_ = [("height", C.c_uint32), ("width", C.c_uint32), ("point_step", C.c_uint32), ("num_fields", C.c_uint32), ("field_offsets", c_uint32_p),
                ("data", C.POINTER(C.c_uint8)), ("data_bytes", C.c_uint64), ("stamp_msg", C.c_double), ("delta_t_pcs", C.c_double)]


SENSORS = {"hesai": 0, "ouster": 1, "robosense": 2, "velodyne": 3, "livoxXYZRTLT_s": 4, "livoxXYZRTLT_ns": 5, "sick": 6, "unknown": 7}


class StaticSelectResult(C.Structure):
    _fields_ = [
        ("num_static", C.c_int64),
        ("keyframe_id", C.c_int32),
        ("min_related_key_id", C.c_int32),
        ("max_overlap", C.c_int32),
        ("pad", C.c_int32),
    ]


class DebugOptions(C.Structure):
    """include/dmsa_debug.h: dmsa_debug_options (fill with dmsa_default_debug_options first)."""
    _fields_ = [(n, C.c_int32) for n in ("device_loop", "dual_stream", "serial_streams", "merge_sort", "key_compress", "fused_segments", "sort_prehist",
                                           "overlap_batch", "serial_tree", "host_threads", "solve_threads", "host_timeline", "trace_time", "fused_leaf_scan", "device_sync", "shared_rotations", "eval_skip", "sync_fault", "speculation_fault", "voxel_coherence", "lm_stream", "stream_priority", "gap_stamps", "lattice_hint", "fit_classes", "eigen_l1_bytes", "small_threshold", "skip_stats", "small_voxel", "long_split", "long_log2", "sort_items", "trial_rows_aside")]


class DebugCounters(C.Structure):
    """include/dmsa_debug.h: dmsa_debug_counters."""
    _fields_ = [(n, C.c_int64) for n in ("sync_retries", "speculation_retries", "skip_pairs", "skip_pairs_equal", "skip_mismatches", "split_blocks", "split_blocks_skipped", "voxel_codes_compared", "voxel_codes_changed",
                                           "lattice_hints_held", "lattice_replays", "voxel_lattice_changes", "small_voxel_launches", "small_voxel_fallbacks")]


class AosView(C.Structure):
    """include/dmsa_aos.h: dmsa_aos_view -- one strided cloud as it lies in memory."""
    _fields_ = [("base", C.c_void_p), ("count", C.c_int64), ("stride", C.c_int32), ("xyz_offset", C.c_int32), ("aux_offset", C.c_int32), ("index", C.POINTER(C.c_int32))]


class RawImu(C.Structure):
    """include/dmsa_raw_sequence.h: dmsa_raw_imu."""
    _fields_ = [("stamp", C.c_double), ("ang_vel", C.c_double * 3), ("lin_acc", C.c_double * 3)]


class WindowRingConfig(C.Structure):
    """include/dmsa_window_ring.h: dmsa_window_ring_config."""
    _fields_ = [("num_scans", C.c_int32), ("max_points_per_scan", C.c_int64), ("max_static_points", C.c_int64), ("max_n_total", C.c_int32),
                ("max_control_poses", C.c_int32)]


class Report(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("stop_reason", C.c_int32),
        ("num_gaussians", C.c_int32),
        ("num_gaussians_l1", C.c_int32),
        ("num_memberships", C.c_int64),
        ("error0", C.c_double),
        ("last_step_norm", C.c_double),
        ("last_line_search_k", C.c_int32),
        ("evaluations", C.c_int32),
    ]


class VoxelLevelInfo(C.Structure):
    _fields_ = [
        ("resolution", C.c_double),
        ("min_xyz", C.c_double * 3),
        ("depth", C.c_int32),
        ("num_events", C.c_int32),
        ("num_leaves", C.c_int64),
        ("num_valid", C.c_int64),
    ]


class IterTrace(C.Structure):
    _fields_ = [("M", C.c_int32), ("M1", C.c_int32), ("Mm", C.c_int64), ("error0", C.c_double), ("step_norm", C.c_double),
                ("best_k", C.c_int32), ("pad", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [
        ("residual_kernel_ms", C.c_double),
        ("residual_launches", C.c_int64),
        ("residual_evaluations", C.c_int64),
        ("residual_algorithmic_bytes", C.c_double),
        ("residual_unit_bytes", C.c_double),
        ("voxelize_ms", C.c_double),
        ("gaussian_fit_ms", C.c_double),
        ("pose_table_ms", C.c_double),
        ("normal_eq_ms", C.c_double),
        ("total_ms", C.c_double),
    ]


def ptr(a: np.ndarray | None, ctype):
    """numpy array -> typed ctypes pointer (None -> NULL). The caller keeps `a` alive."""
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.POINTER(ctype))


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdmsa_hip.so")

_lib = None


class DmsaLibraryMissing(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load libdmsa_hip.so (built in-tree by __graft_entry__.build()).  No fallback: a missing
    library is an error — the product path never routes through a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("DMSA_LIB_PATH", LIB_PATH)  # kernel A/B experiments load an alternative BUILD of the same HIP library
    if not os.path.exists(path):
        raise DmsaLibraryMissing(
            f"{path} not found — run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950) first; there is no CPU fallback"
        )
    lib = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "dmsa_create": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(vp)]),
        "dmsa_destroy": (None, [vp]),
        "dmsa_last_error": (C.c_char_p, [vp]),
        "dmsa_default_settings": (None, [C.POINTER(Settings)]),
        "dmsa_optimize_window": (C.c_int, [vp, C.POINTER(WindowProblem), C.POINTER(Settings), C.POINTER(Report)]),
        "dmsa_optimize_keyframes": (C.c_int, [vp, C.POINTER(KeyframeProblem), C.POINTER(Settings), C.POINTER(Report)]),
        "dmsa_get_global_points": (C.c_int, [vp, c_float_p, C.c_int64]),
        "dmsa_optimize_resident": (C.c_int, [vp, C.POINTER(Settings), C.POINTER(Report)]),
        "dmsa_get_poses": (C.c_int, [vp, c_double_p, c_double_p]),
        "dmsa_window_upload": (C.c_int, [vp, C.POINTER(WindowProblem)]),
        "dmsa_keyframes_upload": (C.c_int, [vp, C.POINTER(KeyframeProblem)]),
        "dmsa_centralize": (C.c_int, [vp]),
        "dmsa_decentralize": (C.c_int, [vp]),
        "dmsa_get_params": (C.c_int, [vp, c_double_p, c_int32_p]),
        "dmsa_set_params": (C.c_int, [vp, c_double_p]),
        "dmsa_additional_errors": (C.c_int, [vp, c_double_p, C.c_int32, c_int32_p]),
        "dmsa_pose_tables": (C.c_int, [vp, C.c_int32, c_double_p, c_float_p]),
        "dmsa_set_pose_tables": (C.c_int, [vp, C.c_int32, c_float_p]),
        "dmsa_num_table_rows": (C.c_int, [vp, c_int32_p]),
        "dmsa_transform_points": (C.c_int, [vp, C.c_int32, c_float_p]),
        "dmsa_build_gaussians": (C.c_int, [vp, C.POINTER(Settings), c_int32_p, c_int64_p]),
        "dmsa_eval_residuals": (C.c_int, [vp, c_double_p]),
        "dmsa_normal_equations": (C.c_int, [vp, C.c_int32, C.c_int32, c_double_p, C.c_double, C.c_double, c_double_p, c_double_p]),
        "dmsa_get_voxel_level": (C.c_int, [vp, C.c_int32, C.POINTER(VoxelLevelInfo), c_uint64_p, c_uint32_p, c_int32_p]),
        "dmsa_get_gaussians": (C.c_int, [vp, c_int32_p, c_int32_p, c_float_p, c_float_p]),
        "dmsa_get_timing": (C.c_int, [vp, C.POINTER(Timing), C.c_int32]),
        "dmsa_synchronize": (C.c_int, [vp]),
        "dmsa_get_trace": (C.c_int, [vp, C.POINTER(IterTrace), C.c_int32]),
        "dmsa_detmath_eval": (C.c_int, [vp, C.c_int32, c_double_p, c_double_p, C.c_int64, c_double_p]),
        "dmsa_sort_pairs": (C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "dmsa_leaf_segments": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_int64, C.c_uint32, c_int32_p, c_int32_p, c_int32_p]),
        "dmsa_serial_fallback_sums": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_uint64)]),
        "dmsa_adaptive_step_size": (C.c_int, [vp, c_double_p, c_double_p, C.c_double, c_int32_p]),
        "dmsa_default_debug_options": (None, [C.POINTER(DebugOptions)]),
        "dmsa_create_ex": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(DebugOptions), C.POINTER(vp)]),
        "dmsa_create_ex2": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(DebugOptions), C.c_uint32, C.POINTER(vp)]),
        "dmsa_get_debug_counters": (C.c_int, [vp, C.POINTER(DebugCounters)]),
        "dmsa_debug_pow_minus_one": (C.c_int, [vp, c_int32_p, C.c_int32, c_float_p]),
        "dmsa_debug_limit_covariance": (C.c_int, [vp, c_float_p, C.c_int64, c_float_p, c_float_p, c_float_p, c_int32_p, c_int32_p]),
        "dmsa_sort_pairs64": (C.c_int, [vp, c_uint64_p, c_uint32_p, C.c_int64, C.c_uint32, c_uint64_p, c_uint32_p]),
        "dmsa_scan_i32": (C.c_int, [vp, c_int32_p, C.c_int64, C.c_int32, c_int32_p]),
        # include/dmsa_aos.h
        "dmsa_window_upload_aos": (C.c_int, [vp, C.POINTER(WindowProblem), C.POINTER(AosView), C.c_int32, C.POINTER(AosView)]),
        "dmsa_keyframes_upload_aos": (C.c_int, [vp, C.POINTER(KeyframeProblem), C.POINTER(AosView), C.c_int32]),
        "dmsa_optimize_window_aos": (C.c_int, [vp, C.POINTER(WindowProblem), C.POINTER(AosView), C.c_int32, C.POINTER(AosView), C.POINTER(Settings), C.POINTER(Report)]),
        "dmsa_optimize_keyframes_aos": (C.c_int, [vp, C.POINTER(KeyframeProblem), C.POINTER(AosView), C.c_int32, C.POINTER(Settings), C.POINTER(Report)]),
        "dmsa_get_global_points_aos": (C.c_int, [vp, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
        "dmsa_reserve": (C.c_int, [vp, C.c_int64, C.c_int32, C.c_int32]),
        "dmsa_window_ring_push_aos": (C.c_int, [vp, C.POINTER(AosView), C.c_int32]),
        "dmsa_window_upload_from_ring_aos": (C.c_int, [vp, C.POINTER(WindowProblem), C.c_double, C.POINTER(AosView)]),
        # include/dmsa_window_ring.h
        "dmsa_window_ring_create": (C.c_int, [vp, C.POINTER(WindowRingConfig)]),
        "dmsa_window_ring_push": (C.c_int, [vp, c_float_p, c_double_p, c_int32_p, C.c_int64]),
        "dmsa_window_ring_points": (C.c_int, [vp, c_int32_p, c_int64_p]),
        "dmsa_window_upload_from_ring": (C.c_int, [vp, C.POINTER(WindowProblem), C.c_double]),
        "dmsa_lm_solve": (C.c_int, [c_double_p, c_double_p, C.c_int32, C.c_double, C.c_int32, c_double_p]),
        "dmsa_lm_solve_device": (C.c_int, [vp, c_double_p, c_double_p, C.c_int32, C.c_double, C.c_double, c_double_p, c_int32_p]),
        "dmsa_neighbourhood_ranges": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "dmsa_submap_poses": (C.c_int, [C.c_int32, c_double_p, c_double_p, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p]),
        "dmsa_update_poses_from_submap": (C.c_int, [C.c_int32, c_double_p, c_double_p, C.c_int32, C.c_int32, c_double_p, c_double_p]),
        # include/dmsa_static_points.h
        "dmsa_select_static_points": (C.c_int, [vp, C.POINTER(StaticSelectProblem), c_float_p, c_int32_p, C.c_int64, c_int32_p, C.POINTER(StaticSelectResult)]),
        "dmsa_get_overlap": (C.c_int, [vp, c_float_p, C.c_int64, c_float_p, C.c_int64, C.c_float, c_float_p, c_int64_p]),
        "dmsa_random_grid_downsampling": (C.c_int, [vp, c_float_p, C.c_int64, C.c_float, C.c_uint32, c_int32_p, C.c_int64, c_int64_p]),
        "dmsa_radius_exists": (C.c_int, [vp, c_float_p, C.c_int64, c_float_p, C.c_int64, C.c_float, C.POINTER(C.c_uint8)]),
        "dmsa_preprocess_scan": (C.c_int, [vp, c_float_p, C.c_int64, C.POINTER(PreprocessConfig), c_float_p, c_int32_p, C.c_int64, c_int64_p, c_float_p]),
        # include/dmsa_window_setup.h
        "dmsa_imu_buffer_create": (C.c_int, [C.c_int32, C.POINTER(vp)]),
        "dmsa_imu_buffer_destroy": (None, [vp]),
        "dmsa_imu_buffer_add": (C.c_int, [vp, c_double_p, c_double_p, C.c_double]),
        "dmsa_imu_buffer_closest": (C.c_int, [vp, C.c_double, c_double_p, c_double_p, c_double_p]),
        "dmsa_imu_buffer_state": (C.c_int, [vp, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]),
        "dmsa_traj_dims": (C.c_int, [C.c_double, C.c_double, C.c_double, c_double_p, c_int32_p]),
        "dmsa_traj_grids": (C.c_int, [C.c_double, C.c_double, C.c_int32, C.c_int32, c_double_p, c_double_p, c_int32_p]),
        "dmsa_traj_tform_indices": (C.c_int, [vp, c_double_p, C.c_int64, C.c_double, c_double_p, C.c_int32, c_int32_p]),
        "dmsa_traj_transfer_imu": (C.c_int, [vp, C.c_double, c_double_p, C.c_int32, c_double_p, c_double_p, c_double_p]),
        "dmsa_traj_preint_factors": (C.c_int, [C.c_int32, C.c_int32, c_int32_p, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                               c_double_p, c_double_p, c_double_p]),
        "dmsa_traj_update_initial_guess": (C.c_int, [c_int32_p, C.POINTER(TrajState), C.POINTER(TrajState), C.c_int32]),
        "dmsa_traj_submap_gravity_estimate": (C.c_int, [C.POINTER(TrajState), c_double_p, c_double_p]),
        # include/dmsa_wire_formats.h
        "dmsa_decode_pointcloud2": (C.c_int, [vp, C.POINTER(PointCloud2), C.c_int32, c_float_p, c_double_p, c_int32_p]),
        "dmsa_format_tum_pose": (C.c_int, [C.c_double, c_double_p, c_double_p, C.c_char_p, C.c_int32]),
        "dmsa_compose_nonkeyframe_pose": (C.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
        # include/dmsa_raw_sequence.h
        "dmsa_raw_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "dmsa_raw_close": (None, [vp]),
        "dmsa_raw_next": (C.c_int, [vp, c_int32_p, C.POINTER(PointCloud2), C.POINTER(RawImu)]),
        "dmsa_raw_create": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "dmsa_raw_write_pointcloud2": (C.c_int, [vp, C.POINTER(PointCloud2)]),
        "dmsa_raw_write_imu": (C.c_int, [vp, C.POINTER(RawImu)]),
        "dmsa_raw_finish": (C.c_int, [vp]),
        # include/dmsa_keyframe_cloud.h
        "dmsa_update_normals": (C.c_int, [vp, c_float_p, C.c_int64, C.c_int32, C.c_float, c_float_p, c_float_p, c_int32_p]),
        "dmsa_make_keyframe_cloud": (C.c_int, [vp, c_float_p, c_int32_p, C.c_int64, C.c_float, C.c_uint32, c_double_p, c_double_p, c_float_p, c_float_p, c_int32_p,
                                               c_int32_p, C.c_int64, c_int64_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "dmsa_window_upload_aos dmsa_keyframes_upload_aos dmsa_optimize_window_aos dmsa_optimize_keyframes_aos dmsa_get_global_points_aos dmsa_reserve dmsa_window_ring_push_aos dmsa_window_upload_from_ring_aos "
    "dmsa_create dmsa_create_ex dmsa_create_ex2 dmsa_default_debug_options dmsa_get_debug_counters dmsa_debug_pow_minus_one dmsa_debug_limit_covariance dmsa_sort_pairs64 dmsa_scan_i32 dmsa_destroy dmsa_last_error dmsa_default_settings dmsa_optimize_window dmsa_optimize_keyframes "
    "dmsa_get_global_points dmsa_window_upload dmsa_keyframes_upload dmsa_centralize dmsa_decentralize dmsa_get_params "
    "dmsa_set_params dmsa_additional_errors dmsa_pose_tables dmsa_set_pose_tables dmsa_num_table_rows dmsa_transform_points dmsa_build_gaussians "
    "dmsa_eval_residuals dmsa_normal_equations dmsa_get_voxel_level dmsa_get_gaussians dmsa_get_timing dmsa_synchronize dmsa_get_trace dmsa_detmath_eval dmsa_lm_solve dmsa_lm_solve_device dmsa_serial_fallback_sums dmsa_sort_pairs dmsa_leaf_segments dmsa_neighbourhood_ranges dmsa_submap_poses dmsa_update_poses_from_submap dmsa_optimize_resident dmsa_adaptive_step_size dmsa_get_poses "
    "dmsa_select_static_points dmsa_get_overlap dmsa_random_grid_downsampling dmsa_radius_exists dmsa_preprocess_scan "
    "dmsa_imu_buffer_create dmsa_imu_buffer_destroy dmsa_imu_buffer_add dmsa_imu_buffer_closest dmsa_imu_buffer_state dmsa_traj_dims dmsa_traj_grids "
    "dmsa_traj_tform_indices dmsa_traj_transfer_imu dmsa_traj_preint_factors dmsa_traj_update_initial_guess dmsa_traj_submap_gravity_estimate "
    "dmsa_window_ring_create dmsa_window_ring_push dmsa_window_ring_points dmsa_window_upload_from_ring "
    "dmsa_raw_open dmsa_raw_close dmsa_raw_next dmsa_raw_create dmsa_raw_write_pointcloud2 dmsa_raw_write_imu dmsa_raw_finish "
    "dmsa_decode_pointcloud2 dmsa_format_tum_pose dmsa_compose_nonkeyframe_pose dmsa_update_normals dmsa_make_keyframe_cloud"
).split()
