"""Loads surfelmeshing_amd/libsmx.so (the C-ABI of include/smx.h) through ctypes.

There is no CPU fallback: if the HIP library is missing or does not load, every
entry point of this package raises.  If torch is already imported (bench.py
imports it first for torch.distributed), the library binds to the HIP runtime
torch has loaded, so both share one runtime per process.
"""
import ctypes as C
import os

# One hardware queue per busy stream (smx_buffer.hip: runtime_advice): decided when the HIP runtime initialises, i.e. at the
# process's first HIP call -- possibly torch's -- so this module, the APPLICATION side of the binding, raises the default as
# early as its import.  libsmx.so itself does not touch the environment (smx_runtime_advice reports instead).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_HERE = os.path.dirname(os.path.abspath(__file__))
# (SMX_LIB_PATH: A/B measurements of two builds in one session; the product loads the in-tree library)
SO_PATH = os.environ.get("SMX_LIB_PATH") or os.path.join(_HERE, "libsmx.so")


class SmxError(RuntimeError):
    pass


class BufferDesc(C.Structure):
    """smx_buffer_desc == CUDABuffer_<T> (VIS/cuda/cuda_buffer.cuh:44-119)."""
    _fields_ = [("address", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("pitch", C.c_size_t)]


class IntegrateParams(C.Structure):
    """smx_integrate_params: trailing arguments of CUDASurfelReconstruction::Integrate."""
    _fields_ = [("sensor_noise_factor", C.c_float),
                ("max_surfel_confidence", C.c_float),
                ("regularizer_weight", C.c_float),
                ("regularization_frame_window_size", C.c_int32),
                ("do_blending", C.c_int32),
                ("measurement_blending_radius", C.c_int32),
                ("regularization_iterations_per_integration_iteration", C.c_int32),
                ("radius_factor_for_regularization_neighbors", C.c_float),
                ("normal_compatibility_threshold_deg", C.c_float),
                ("surfel_integration_active_window_size", C.c_int32)]

    @classmethod
    def defaults(cls, **kw):
        p = cls(0.05, 5.0, 10.0, 30, 1, 12, 1, 2.0, 40.0, 2147483647)  # APP/main.cc:323-368
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class SurfelBuffersCPU(C.Structure):
    """smx_surfel_buffers_cpu == CUDASurfelBuffersCPU (APP/cuda_surfels_cpu.h:40-74)."""
    _fields_ = [("frame_index", C.c_uint32), ("surfel_count", C.c_size_t),
                ("surfel_x_buffer", C.c_void_p), ("surfel_y_buffer", C.c_void_p), ("surfel_z_buffer", C.c_void_p),
                ("surfel_radius_squared_buffer", C.c_void_p),
                ("surfel_normal_x_buffer", C.c_void_p), ("surfel_normal_y_buffer", C.c_void_p),
                ("surfel_normal_z_buffer", C.c_void_p), ("surfel_last_update_stamp_buffer", C.c_void_p)]


class SurfelDeltaCPU(C.Structure):
    """smx_surfel_delta_cpu: the changed-surfel delta of smx_recon_transfer_changed_to_cpu."""
    _fields_ = [("capacity", C.c_uint32), ("count", C.c_uint32), ("frame_index", C.c_uint32), ("surfel_count", C.c_uint32),
                ("surfel_index", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p),
                ("radius_squared", C.c_void_p), ("normal_x", C.c_void_p), ("normal_y", C.c_void_p),
                ("normal_z", C.c_void_p), ("last_update_stamp", C.c_void_p)]


class ReconStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("surfels_size", "merge_count", "n_visible", "n_new", "n_merged", "n_recent", "n_edges",
                 "n_integrated", "n_replaced", "n_conflict_hits", "capacity_clamped", "n_window_edges", "n_contributors",
                 "n_segments_skipped", "regularizer_saturated", "n_pairs", "n_overflow_pairs", "max_tile_pairs")]


class NNStats(C.Structure):
    """smx_nn_stats"""
    _fields_ = [("n_points", C.c_uint32), ("n_indexed", C.c_uint32), ("n_bricks", C.c_uint32), ("cell_size", C.c_float),
                ("dim", C.c_int32 * 3), ("key_bits", C.c_int32), ("tiles", C.c_uint64), ("staged_candidates", C.c_uint64),
                ("distance_tests", C.c_uint64), ("results", C.c_uint64)]


# every symbol include/smx.h declares (tests/test_abi.py checks the .so exports them all)
EXPORTS = [
    "smx_last_error", "smx_runtime_advice", "smx_host_is_page_locked", "smx_device_count", "smx_set_device", "smx_device_name",
    "smx_stream_create", "smx_stream_create_with_priority", "smx_stream_create_with_cu_mask", "smx_recon_set_internal_cu_mask", "smx_recon_set_handover_mode", "smx_recon_get_handover_mode", "smx_host_alloc", "smx_host_free", "smx_stream_destroy", "smx_stream_synchronize", "smx_debug_marker", "smx_debug_handover_probe",
    "smx_event_create", "smx_event_create_timed", "smx_event_elapsed_ms", "smx_event_destroy", "smx_event_record", "smx_stream_wait_event",
    "smx_buffer_create", "smx_buffer_destroy", "smx_buffer_get_desc", "smx_buffer_upload", "smx_buffer_upload_by_kernel", "smx_buffer_download",
    "smx_buffer_upload_part", "smx_buffer_download_part", "smx_buffer_clear", "smx_buffer_set_to",
    "smx_bilateral_filtering_and_depth_cutoff", "smx_outlier_depth_map_fusion", "smx_bilateral_outlier_fusion", "smx_erode_depth_map",
    "smx_copy_without_border", "smx_median_filter_and_densify_depth_map", "smx_downscale_using_median_while_excluding", "smx_color_image_pyramid", "smx_compute_normals_and_drop_bad_pixels",
    "smx_compute_point_radii_and_remove_isolated_pixels", "smx_erode_normals_radii", "smx_erode_normals_radii_signal",
    "smx_recon_create", "smx_recon_destroy", "smx_recon_integrate", "smx_recon_regularize",
    "smx_recon_transfer_all_to_cpu", "smx_recon_set_delta_tracking", "smx_recon_transfer_changed_to_cpu", "smx_recon_export_vertices", "smx_recon_get_timings", "smx_recon_get_timings_nowait", "smx_recon_debug_stamp_ring", "smx_recon_debug_internal_stream",
    "smx_recon_build_neighbor_index", "smx_recon_neighbor_candidates", "smx_recon_check_triangles", "smx_recon_deform_by_creation_frame",
    "smx_recon_set_timing_enabled", "smx_recon_counts", "smx_recon_get_stats", "smx_recon_set_stats_enabled",
    "smx_recon_kernel_slot_count", "smx_recon_kernel_slot_name", "smx_recon_get_kernel_timings",
    "smx_recon_profile_begin", "smx_recon_profile_end",
    "smx_recon_debug_download_surfels", "smx_recon_debug_upload_surfels", "smx_recon_debug_download_scratch", "smx_recon_debug_count_skipped_segments",
    "smx_recon_set_scan_mode", "smx_recon_debug_set_skip", "smx_recon_set_overlap", "smx_recon_integrate_hooks", "smx_recon_integrate_inputs_ready",
    "smx_nn_create", "smx_nn_destroy", "smx_nn_build", "smx_nn_query_batch", "smx_nn_query_self", "smx_nn_set_query_mode", "smx_nn_set_stats_enabled", "smx_nn_get_stats",
    "smx_synth_render_room",
]

_lib = None


def load():
    """Returns the ctypes handle of libsmx.so; raises SmxError if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise SmxError("%s is missing: run `python -m surfelmeshing_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback." % SO_PATH)
    try:
        L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise SmxError("cannot load %s: %s" % (SO_PATH, e))
    L.smx_last_error.restype = C.c_char_p
    L.smx_recon_kernel_slot_name.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("smx_last_error", "smx_recon_kernel_slot_name"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().smx_last_error()
        raise SmxError("libsmx error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
    n = C.c_int(0)
    check(load().smx_device_count(C.byref(n)))
    return n.value


def require_gpu():
    if device_count() == 0:
        raise SmxError("no HIP device visible: the surfel-integration path needs an MI355X (no CPU fallback)")
