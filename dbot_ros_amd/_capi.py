"""ctypes binding of librbsensor_mi355x.so (include/rbsensor_mi355x.h).

The library is the product; this module only declares its prototypes.  Loading fails loudly
when the shared object has not been built (``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C dbot_ros_amd/csrc``) -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RBS_LIB_PATH selects an alternative build of the SAME library (kernel tuning experiments)
LIB_PATH = os.environ.get("RBS_LIB_PATH") or os.path.join(_HERE, "lib", "librbsensor_mi355x.so")

RBS_ABI_VERSION = 2
RBS_OK = 0
RBS_ERR_INVALID_ARGUMENT = -1
RBS_ERR_NO_DEVICE = -2
RBS_ERR_OUT_OF_MEMORY = -3
RBS_ERR_HIP = -4
RBS_ERR_UNSUPPORTED = -5
RBS_IPC_BLOB_BYTES = 512
RBS_PRECISION_DEFAULT, RBS_PRECISION_F64, RBS_PRECISION_F32 = 0, 1, 2
RBS_STATE_DEFAULT, RBS_STATE_WINDOWED, RBS_STATE_DENSE = 0, 1, 2
PRECISIONS = {None: 0, "default": 0, "f64": 1, "f32": 2}
LAYOUTS = {None: 0, "default": 0, "window": 1, "windowed": 1, "dense": 2}
RBS_OCC_DEFAULT, RBS_OCC_DEVICE_RULE, RBS_OCC_REFERENCE = 0, 1, 2
OPTIONS = {"shared_trail": 1, "shared_trail_enter": 2, "shared_trail_every": 3, "tracker_split_max": 4, "timing_every": 5}
OCC_MODES = {None: 0, "default": 0, "device": 1, "eager": 1, "reference": 2, "lazy": 2, "exact": 2}

# every symbol include/rbsensor_mi355x.h declares
EXPORTS = (
    "rbs_abi_version", "rbs_device_count", "rbs_create", "rbs_destroy", "rbs_last_error",
    "rbs_reset", "rbs_set_observation", "rbs_set_observation_f32",
    "rbs_set_observation_native_f32", "rbs_set_observation_device", "rbs_get_observation", "rbs_loglikes",
    "rbs_acquire_frame_buffer", "rbs_commit_frame_buffer", "rbs_loglikes_prefetch", "rbs_set_observation_prefetched",
    "rbs_loglikes_deltas", "rbs_get_poses", "rbs_deltas_buffer", "rbs_set_observation_borrowed", "rbs_set_observation_borrowed_f32", "rbs_shared_trail_state",
    "rbs_shared_trail_rebase", "rbs_window_fraction", "rbs_set_option",
    "rbs_loglikes_device", "rbs_synchronize", "rbs_get_occlusion", "rbs_set_occlusion",
    "rbs_occlusion_device_ptr", "rbs_occlusion_next_device_ptr", "rbs_export_plane", "rbs_import_plane",
    "rbs_export_window", "rbs_import_window", "rbs_stream_join", "rbs_ipc_export", "rbs_ipc_attach", "rbs_stage_windows", "rbs_peer_resample",
    "rbs_get_window", "rbs_get_background", "rbs_raster_kernel_ms", "rbs_set_timing_every",
    "rbs_render_depth",
    "rbs_last_kernel_ms", "rbs_timing_summary",
    "rbs_tracker_create", "rbs_tracker_destroy", "rbs_tracker_initialize", "rbs_tracker_track",
    "rbs_tracker_submit", "rbs_tracker_result", "rbs_tracker_track_f64", "rbs_tracker_submit_f64",
    "rbs_tracker_get",
)


class RbsTrackerParams(C.Structure):
    _fields_ = [
        ("linear_sigma", C.c_double * 3),
        ("angular_sigma", C.c_double * 3),
        ("velocity_factor", C.c_double),
        ("max_kl_divergence", C.c_double),
        ("n_particles", C.c_int32),
    ]


class RbsConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("device_id", C.c_int32),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("K", C.c_double * 9),
        ("max_particles", C.c_int32),
        ("n_objects", C.c_int32),
        ("vertices", C.POINTER(C.c_double)),
        ("vertex_counts", C.POINTER(C.c_int32)),
        ("triangles", C.POINTER(C.c_int32)),
        ("triangle_counts", C.POINTER(C.c_int32)),
        ("p_occluded_visible", C.c_double),
        ("p_occluded_occluded", C.c_double),
        ("initial_occlusion_prob", C.c_double),
        ("tail_weight", C.c_double),
        ("model_sigma", C.c_double),
        ("sigma_factor", C.c_double),
        ("delta_time", C.c_double),
        ("likelihood_precision", C.c_int32),
        ("state_layout", C.c_int32),
        ("n_devices", C.c_int32),
        ("device_ids", C.POINTER(C.c_int32)),
        ("state_slab_px", C.c_int32),
        ("occlusion_mode", C.c_int32),
    ]


_lib = None


def load():
    """Return the loaded library (cached). Raises OSError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(make -C dbot_ros_amd/csrc, or __graft_entry__.build()). "
            "dbot_ros_amd has no CPU fallback.")
    # When torch is in the process its bundled libamdhip64.so.7 must be the one HIP runtime;
    # importing it first makes our DT_NEEDED resolve to the already-loaded copy.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the C-ABI
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    H = C.c_void_p
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.rbs_abi_version.restype = C.c_int32
    lib.rbs_abi_version.argtypes = []
    lib.rbs_device_count.restype = C.c_int32
    lib.rbs_device_count.argtypes = []
    lib.rbs_create.restype = C.c_int32
    lib.rbs_create.argtypes = [C.POINTER(RbsConfig), C.POINTER(H)]
    lib.rbs_destroy.restype = None
    lib.rbs_destroy.argtypes = [H]
    lib.rbs_last_error.restype = C.c_char_p
    lib.rbs_last_error.argtypes = [H]
    lib.rbs_reset.restype = C.c_int32
    lib.rbs_reset.argtypes = [H]
    lib.rbs_set_observation.restype = C.c_int32
    lib.rbs_set_observation.argtypes = [H, dp, C.c_size_t]
    lib.rbs_set_observation_borrowed.restype = C.c_int32
    lib.rbs_set_observation_borrowed.argtypes = [H, dp, C.c_size_t]
    lib.rbs_set_observation_borrowed_f32.restype = C.c_int32
    lib.rbs_set_observation_borrowed_f32.argtypes = [H, fp, C.c_size_t]
    lib.rbs_set_observation_f32.restype = C.c_int32
    lib.rbs_set_observation_f32.argtypes = [H, fp, C.c_size_t]
    lib.rbs_set_observation_native_f32.restype = C.c_int32
    lib.rbs_set_observation_native_f32.argtypes = [H, fp, C.c_int32, C.c_int32, C.c_int32]
    lib.rbs_acquire_frame_buffer.restype = C.c_int32
    lib.rbs_acquire_frame_buffer.argtypes = [H, C.POINTER(C.POINTER(C.c_float))]
    lib.rbs_commit_frame_buffer.restype = C.c_int32
    lib.rbs_commit_frame_buffer.argtypes = [H]
    lib.rbs_get_observation.restype = C.c_int32
    lib.rbs_get_observation.argtypes = [H, fp]
    lib.rbs_loglikes.restype = C.c_int32
    lib.rbs_loglikes.argtypes = [H, dp, ip, C.c_int32, C.c_int32, dp]
    lib.rbs_loglikes_deltas.restype = C.c_int32
    lib.rbs_loglikes_deltas.argtypes = [H, dp, dp, C.c_int32, ip, C.c_int32, C.c_int32, dp]
    lib.rbs_deltas_buffer.restype = C.c_int32
    lib.rbs_deltas_buffer.argtypes = [H, C.POINTER(dp)]
    lib.rbs_get_poses.restype = C.c_int32
    lib.rbs_get_poses.argtypes = [H, dp, C.c_int32]
    lib.rbs_loglikes_device.restype = C.c_int32
    lib.rbs_loglikes_device.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p]
    lib.rbs_synchronize.restype = C.c_int32
    lib.rbs_synchronize.argtypes = [H]
    lib.rbs_get_occlusion.restype = C.c_int32
    lib.rbs_get_occlusion.argtypes = [H, C.c_int32, fp]
    lib.rbs_set_occlusion.restype = C.c_int32
    lib.rbs_set_occlusion.argtypes = [H, C.c_int32, fp]
    lib.rbs_occlusion_device_ptr.restype = C.c_int32
    lib.rbs_occlusion_device_ptr.argtypes = [H, C.c_int32, C.POINTER(C.c_void_p)]
    lib.rbs_occlusion_next_device_ptr.restype = C.c_int32
    lib.rbs_occlusion_next_device_ptr.argtypes = [H, C.c_int32, C.POINTER(C.c_void_p)]
    lib.rbs_set_observation_device.restype = C.c_int32
    lib.rbs_set_observation_device.argtypes = [H, C.c_void_p, C.c_void_p]
    lib.rbs_get_window.restype = C.c_int32
    lib.rbs_get_window.argtypes = [H, C.c_int32, C.POINTER(C.c_int32)]
    lib.rbs_shared_trail_state.restype = C.c_int32
    lib.rbs_shared_trail_state.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.rbs_shared_trail_rebase.restype = C.c_int32
    lib.rbs_shared_trail_rebase.argtypes = [H, C.c_int32]
    lib.rbs_set_option.restype = C.c_int32
    lib.rbs_set_option.argtypes = [H, C.c_int32, C.c_double]
    lib.rbs_window_fraction.restype = C.c_int32
    lib.rbs_window_fraction.argtypes = [H, C.POINTER(C.c_double)]
    lib.rbs_get_background.restype = C.c_int32
    lib.rbs_get_background.argtypes = [H, C.POINTER(C.c_float)]
    lib.rbs_raster_kernel_ms.restype = C.c_int32
    lib.rbs_raster_kernel_ms.argtypes = [H, C.c_int32, C.POINTER(C.c_float)]
    lib.rbs_set_timing_every.restype = C.c_int32
    lib.rbs_set_timing_every.argtypes = [H, C.c_int32]
    lib.rbs_export_plane.restype = C.c_int32
    lib.rbs_export_plane.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rbs_import_plane.restype = C.c_int32
    lib.rbs_import_plane.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rbs_loglikes_prefetch.restype = C.c_int32
    lib.rbs_loglikes_prefetch.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                          C.POINTER(C.c_float), C.c_size_t]
    lib.rbs_set_observation_prefetched.restype = C.c_int32
    lib.rbs_set_observation_prefetched.argtypes = [H]
    lib.rbs_export_window.restype = C.c_int32
    lib.rbs_export_window.argtypes = [H, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.rbs_import_window.restype = C.c_int32
    lib.rbs_import_window.argtypes = [H, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    lib.rbs_stream_join.restype = C.c_int32
    lib.rbs_stream_join.argtypes = [H, C.c_void_p]
    lib.rbs_ipc_export.restype = C.c_int32
    lib.rbs_ipc_export.argtypes = [H, C.c_void_p]
    lib.rbs_ipc_attach.restype = C.c_int32
    lib.rbs_ipc_attach.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p]
    lib.rbs_stage_windows.restype = C.c_int32
    lib.rbs_stage_windows.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rbs_peer_resample.restype = C.c_int32
    lib.rbs_peer_resample.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rbs_render_depth.restype = C.c_int32
    lib.rbs_render_depth.argtypes = [H, dp, fp]
    lib.rbs_last_kernel_ms.restype = C.c_int32
    lib.rbs_last_kernel_ms.argtypes = [H, C.POINTER(C.c_float)]
    lib.rbs_timing_summary.restype = C.c_int32
    lib.rbs_timing_summary.argtypes = [H, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.POINTER(C.c_int32)]
    lib.rbs_tracker_create.restype = C.c_int32
    lib.rbs_tracker_create.argtypes = [H, C.POINTER(RbsTrackerParams), C.POINTER(H)]
    lib.rbs_tracker_destroy.restype = None
    lib.rbs_tracker_destroy.argtypes = [H]
    lib.rbs_tracker_initialize.restype = C.c_int32
    lib.rbs_tracker_initialize.argtypes = [H, dp]
    lib.rbs_tracker_track.restype = C.c_int32
    lib.rbs_tracker_track.argtypes = [H, fp, dp, dp, C.c_uint64, dp, ip]
    lib.rbs_tracker_track_f64.restype = C.c_int32
    lib.rbs_tracker_track_f64.argtypes = [H, dp, dp, dp, C.c_uint64, dp, ip]
    lib.rbs_tracker_submit_f64.restype = C.c_int32
    lib.rbs_tracker_submit_f64.argtypes = [H, dp, dp, dp, C.c_uint64]
    lib.rbs_tracker_submit.restype = C.c_int32
    lib.rbs_tracker_submit.argtypes = [H, fp, dp, dp, C.c_uint64]
    lib.rbs_tracker_result.restype = C.c_int32
    lib.rbs_tracker_result.argtypes = [H, dp, ip]
    lib.rbs_tracker_get.restype = C.c_int32
    lib.rbs_tracker_get.argtypes = [H, dp, dp, ip]
    _lib = lib
    return lib
