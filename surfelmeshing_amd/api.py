"""Host-side mirror of the reference's interface for the surfel-integration path.

Same names, argument order and meaning as the reference's C++ API, on top of the
C-ABI of include/smx.h (no torch types anywhere):

  CUDABuffer                     VIS/cuda/cuda_buffer.h:45-129
  BilateralFilteringAndDepthCutoffCUDA, OutlierDepthMapFusionCUDA, ErodeDepthMapCUDA,
  CopyWithoutBorderCUDA, ComputeNormalsAndDropBadPixelsCUDA,
  ComputePointRadiiAndRemoveIsolatedPixelsCUDA
                                 APP/cuda_depth_processing.cuh:43-122
  CUDASurfelReconstruction       APP/cuda_surfel_reconstruction.h:44-176
  CUDASurfelBuffersCPU, CUDASurfelsCPU
                                 APP/cuda_surfels_cpu.h:40-124
  SurfelNeighborIndex            batched FindNearestSurfelsWithinRadius, APP/octree.h:470-477

Errors: the reference aborts through LOG(FATAL); here a non-zero C-ABI status
raises SmxError (there is no CPU fallback).
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import BufferDesc, IntegrateParams, SmxError, SurfelBuffersCPU, ReconStats  # noqa: F401

kInvalidSurfelIndex = 0xFFFFFFFF  # APP/surfel.h (Surfel::kInvalidIndex)
kSurfelAttributeCount = 25        # APP/cuda_surfel_reconstruction_kernels.cuh:76


def _stream(s):
    return C.c_void_p(s) if s else C.c_void_p(0)


class Stream:
    """A HIP stream (cudaStream_t in the reference's signatures)."""

    def __init__(self, priority_class=None, cu_mask=None):
        """priority_class: None = plain stream; -1 / 0 / +1 = lowest / default / highest device priority
        (cudaStreamCreateWithPriority).  cu_mask: list of 32-bit words, bit k = compute unit k may be used
        (smx_stream_create_with_cu_mask; default priority)."""
        self.handle = C.c_void_p()
        if cu_mask:
            arr = (C.c_uint32 * len(cu_mask))(*cu_mask)
            _lib.check(_lib.load().smx_stream_create_with_cu_mask(C.byref(self.handle), arr, C.c_uint32(len(cu_mask))))
        elif priority_class is None:
            _lib.check(_lib.load().smx_stream_create(C.byref(self.handle)))
        else:
            _lib.check(_lib.load().smx_stream_create_with_priority(C.byref(self.handle), C.c_int32(priority_class)))

    def synchronize(self):
        _lib.check(_lib.load().smx_stream_synchronize(self.handle))

    def __int__(self):
        return self.handle.value or 0

    def close(self):
        if self.handle:
            _lib.load().smx_stream_destroy(self.handle)
            self.handle = C.c_void_p()


class _PagelockedBlock:
    """Owner of one smx_host_alloc block; numpy arrays made from it keep it alive (it is their base object)."""

    def __init__(self, nbytes, write_combined):
        self._p = C.c_void_p()
        _lib.check(_lib.load().smx_host_alloc(C.byref(self._p), C.c_size_t(max(nbytes, 1)),
                                              C.c_int32(1 if write_combined else 0)))
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self._p.value, False), "version": 3}

    def __del__(self):
        try:
            if self._p:
                _lib.load().smx_host_free(self._p)
                self._p = C.c_void_p()
        except Exception:
            pass


class PagelockedArray:
    """Page-locked host memory as a numpy array (cudaHostAlloc in the reference's upload staging,
    APP/main.cc:825-829, 917): uploads from `.array` are asynchronous to the host.  The memory lives as long as this
    object or any view of `.array` does; keep one of them until the copies that read it have finished."""

    def __init__(self, shape, dtype, write_combined=False):
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        self.array = np.asarray(_PagelockedBlock(n, write_combined)).view(self.dtype).reshape(shape)

    def close(self):
        """Drop this object's reference (the block is freed once no view is left)."""
        self.array = None


def _sv(stream):
    if stream is None:
        return C.c_void_p(0)
    if isinstance(stream, Stream):
        return stream.handle
    return C.c_void_p(int(stream))


def StreamSynchronize(stream=None):
    _lib.check(_lib.load().smx_stream_synchronize(_sv(stream)))


class CUDABuffer:
    """CUDABuffer<T>(height, width): pitched 2-D device memory.  `dtype` is the numpy scalar type
    and `channels` the number of scalars per element (float2 -> (np.float32, 2), Vec3u8 -> (np.uint8, 3))."""

    def __init__(self, height, width, dtype, channels=1):
        self.dtype = np.dtype(dtype)
        self.channels = int(channels)
        self.elem_bytes = self.dtype.itemsize * self.channels
        self._h = C.c_void_p()
        _lib.check(_lib.load().smx_buffer_create(int(height), int(width), self.elem_bytes, C.byref(self._h)))
        self._desc = BufferDesc()
        _lib.check(_lib.load().smx_buffer_get_desc(self._h, C.byref(self._desc)))

    # -- reference accessors
    def width(self):
        return self._desc.width

    def height(self):
        return self._desc.height

    def Size(self):
        return self._desc.pitch * self._desc.height

    def ToCUDA(self):
        return self._desc

    def _host_shape(self):
        return (self.height(), self.width()) + ((self.channels,) if self.channels > 1 else ())

    def UploadAsync(self, stream, data):
        a = np.ascontiguousarray(data, dtype=self.dtype)
        assert a.shape == self._host_shape(), (a.shape, self._host_shape())
        _lib.check(_lib.load().smx_buffer_upload(self._h, _sv(stream), a.ctypes.data_as(C.c_void_p), C.c_size_t(0)))
        self._keep = a  # keep the host array alive until the caller synchronises

    def UploadByKernelAsync(self, stream, data):
        """The same copy done by a kernel that reads the page-locked source over the bus (smx_buffer_upload_by_kernel); `data`
        must be (a view of) a PagelockedArray of the buffer's shape and dtype -- raises SmxError otherwise."""
        assert data.dtype == self.dtype and data.shape == self._host_shape() and data.flags.c_contiguous
        _lib.check(_lib.load().smx_buffer_upload_by_kernel(self._h, _sv(stream), data.ctypes.data_as(C.c_void_p), C.c_size_t(0), C.c_void_p(0)))
        self._keep = data

    def UploadPitchedAsync(self, stream, pitch, data):
        _lib.check(_lib.load().smx_buffer_upload(self._h, _sv(stream), data.ctypes.data_as(C.c_void_p), C.c_size_t(pitch)))
        self._keep = data

    def UploadPartAsync(self, start, length, stream, data):
        a = np.ascontiguousarray(data)
        _lib.check(_lib.load().smx_buffer_upload_part(self._h, _sv(stream), C.c_size_t(start), C.c_size_t(length),
                                                      a.ctypes.data_as(C.c_void_p)))
        self._keep = a

    def DownloadAsync(self, stream, out=None):
        if out is None:
            out = np.empty(self._host_shape(), self.dtype)
        assert out.flags.c_contiguous and out.dtype == self.dtype and out.shape == self._host_shape()
        _lib.check(_lib.load().smx_buffer_download(self._h, _sv(stream), out.ctypes.data_as(C.c_void_p), C.c_size_t(0)))
        return out

    def DownloadPartAsync(self, start, length, stream, out):
        _lib.check(_lib.load().smx_buffer_download_part(self._h, _sv(stream), C.c_size_t(start), C.c_size_t(length),
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def Download(self, stream=None):
        """DebugDownload: blocking convenience."""
        out = self.DownloadAsync(stream)
        StreamSynchronize(stream)
        return out

    def Upload(self, data, stream=None):
        """DebugUpload: blocking convenience."""
        self.UploadAsync(stream, data)
        StreamSynchronize(stream)

    def Clear(self, value, stream=None):
        pat = np.asarray(value, dtype=self.dtype).reshape(-1)
        if pat.size == 1 and self.channels > 1:
            pat = np.repeat(pat, self.channels)
        assert pat.size == self.channels
        pat = np.ascontiguousarray(pat)
        _lib.check(_lib.load().smx_buffer_clear(self._h, _sv(stream), pat.ctypes.data_as(C.c_void_p)))

    def SetTo(self, other, stream=None):
        _lib.check(_lib.load().smx_buffer_set_to(self._h, other._h, _sv(stream)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().smx_buffer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _d(buf):
    return C.byref(buf.ToCUDA() if isinstance(buf, CUDABuffer) else buf)


# ---- depth preprocessing free functions (APP/cuda_depth_processing.cuh) -----------------------
def BilateralFilteringAndDepthCutoffCUDA(stream, sigma_xy, sigma_value_factor, value_to_ignore, radius_factor,
                                         max_depth, depth_valid_region_radius, input_depth, output_depth):
    _lib.check(_lib.load().smx_bilateral_filtering_and_depth_cutoff(
        _sv(stream), C.c_float(sigma_xy), C.c_float(sigma_value_factor), C.c_uint16(int(value_to_ignore)),
        C.c_float(radius_factor), C.c_uint16(int(max_depth)), C.c_float(depth_valid_region_radius),
        _d(input_depth), _d(output_depth)))


def OutlierDepthMapFusionCUDA(stream, tolerance, input_depth, depth_fx, depth_fy, depth_cx, depth_cy, other_depths,
                              others_TR_reference, output_depth, required_count=-1):
    """Both overloads of OutlierDepthMapFusionCUDA<count,u16>: count-1 == len(other_depths);
    required_count=-1 is the all-must-agree overload (APP/main.cc:1061-1112)."""
    n = len(other_depths)
    descs = (BufferDesc * n)(*[b.ToCUDA() if isinstance(b, CUDABuffer) else b for b in other_depths])
    T = np.ascontiguousarray(np.asarray(others_TR_reference, np.float32).reshape(n, 12))
    _lib.check(_lib.load().smx_outlier_depth_map_fusion(
        _sv(stream), C.c_int32(n), C.c_int32(required_count), C.c_float(tolerance), _d(input_depth),
        C.c_float(depth_fx), C.c_float(depth_fy), C.c_float(depth_cx), C.c_float(depth_cy),
        descs, T.ctypes.data_as(C.c_void_p), _d(output_depth)))


def BilateralFilteringAndOutlierFusionCUDA(stream, sigma_xy, sigma_value_factor, radius_factor, max_depth,
                                           depth_valid_region_radius, input_depth, tolerance, depth_fx, depth_fy, depth_cx,
                                           depth_cy, other_depths, others_TR_reference, scratch_depth, output_depth,
                                           required_count=-1):
    """BilateralFilteringAndDepthCutoffCUDA (value_to_ignore 0) + OutlierDepthMapFusionCUDA as the reference's caller chains
    them (APP/main.cc:1015-1115): one launch where the library can fuse them (smx_bilateral_outlier_fusion), same output."""
    n = len(other_depths)
    descs = (BufferDesc * n)(*[b.ToCUDA() if isinstance(b, CUDABuffer) else b for b in other_depths])
    T = np.ascontiguousarray(np.asarray(others_TR_reference, np.float32).reshape(n, 12))
    _lib.check(_lib.load().smx_bilateral_outlier_fusion(
        _sv(stream), C.c_float(sigma_xy), C.c_float(sigma_value_factor), C.c_float(radius_factor), C.c_uint16(int(max_depth)),
        C.c_float(depth_valid_region_radius), _d(input_depth), C.c_int32(n), C.c_int32(required_count), C.c_float(tolerance),
        C.c_float(depth_fx), C.c_float(depth_fy), C.c_float(depth_cx), C.c_float(depth_cy),
        descs, T.ctypes.data_as(C.c_void_p), _d(scratch_depth), _d(output_depth)))


def ErodeDepthMapCUDA(stream, radius, input_depth, output_depth):
    _lib.check(_lib.load().smx_erode_depth_map(_sv(stream), C.c_int32(radius), _d(input_depth), _d(output_depth)))


def MedianFilterAndDensifyDepthMapCUDA(stream, input_depth, output_depth):
    """MedianFilterAndDensifyDepthMap (APP/main.cc:206-252, a CPU function in the reference) on the GPU."""
    _lib.check(_lib.load().smx_median_filter_and_densify_depth_map(_sv(stream), _d(input_depth), _d(output_depth)))


def DownscaleUsingMedianWhileExcludingCUDA(stream, value_to_ignore, input_depth, output_depth):
    """Image<u16>::DownscaleUsingMedianWhileExcluding (VIS/image.h:1003-1053, the depth half of --pyramid_level)."""
    _lib.check(_lib.load().smx_downscale_using_median_while_excluding(_sv(stream), C.c_uint16(value_to_ignore),
                                                                      _d(input_depth), _d(output_depth)))


def ColorImagePyramidCUDA(stream, pyramid_level, input_color, output_color):
    """ImagePyramid(color_frame, pyramid_level) (VIS/image_cache.h:203-275 over Image<Vec3u8>::DownscaleToHalfSize,
    VIS/image.h:929-948): the colour half of --pyramid_level (APP/main.cc:973-981)."""
    _lib.check(_lib.load().smx_color_image_pyramid(_sv(stream), C.c_int32(pyramid_level), _d(input_color),
                                                   _d(output_color)))


def CopyWithoutBorderCUDA(stream, input_depth, output_depth):
    _lib.check(_lib.load().smx_copy_without_border(_sv(stream), _d(input_depth), _d(output_depth)))


def ComputeNormalsAndDropBadPixelsCUDA(stream, observation_angle_threshold_deg, depth_scaling, depth_fx, depth_fy,
                                       depth_cx, depth_cy, in_depth, out_depth, out_normals):
    _lib.check(_lib.load().smx_compute_normals_and_drop_bad_pixels(
        _sv(stream), C.c_float(observation_angle_threshold_deg), C.c_float(depth_scaling),
        C.c_float(depth_fx), C.c_float(depth_fy), C.c_float(depth_cx), C.c_float(depth_cy),
        _d(in_depth), _d(out_depth), _d(out_normals)))


def ComputePointRadiiAndRemoveIsolatedPixelsCUDA(stream, point_radius_extension_factor, point_radius_clamp_factor,
                                                 depth_scaling, depth_fx, depth_fy, depth_cx, depth_cy,
                                                 depth_buffer, radius_buffer, out_depth):
    _lib.check(_lib.load().smx_compute_point_radii_and_remove_isolated_pixels(
        _sv(stream), C.c_float(point_radius_extension_factor), C.c_float(point_radius_clamp_factor),
        C.c_float(depth_scaling), C.c_float(depth_fx), C.c_float(depth_fy), C.c_float(depth_cx), C.c_float(depth_cy),
        _d(depth_buffer), _d(radius_buffer), _d(out_depth)))


def ErodeNormalsRadiiCUDA(stream, erosion_radius, observation_angle_threshold_deg, point_radius_extension_factor,
                          point_radius_clamp_factor, depth_scaling, depth_fx, depth_fy, depth_cx, depth_cy, in_depth,
                          out_depth, out_normals, radius_buffer):
    """Erosion (radius 0: border copy) + normals + radii as one launch (smx_erode_normals_radii): the final depth, the
    normals and the radii of the three separate calls (APP/main.cc:1128-1191)."""
    _lib.check(_lib.load().smx_erode_normals_radii(
        _sv(stream), C.c_int32(erosion_radius), C.c_float(observation_angle_threshold_deg),
        C.c_float(point_radius_extension_factor), C.c_float(point_radius_clamp_factor), C.c_float(depth_scaling),
        C.c_float(depth_fx), C.c_float(depth_fy), C.c_float(depth_cx), C.c_float(depth_cy), _d(in_depth), _d(out_depth),
        _d(out_normals), _d(radius_buffer)))


def SynthRenderRoom(stream, depth_out, color_out, fx, fy, cx, cy, global_T_frame, seed, frame_index,
                    depth_scaling=5000.0, noise_sigma=0.001, dropout=0.01):
    """Benchmark input generator (smx_synth_render_room): one synthetic room frame into device buffers."""
    T = np.ascontiguousarray(np.asarray(global_T_frame, np.float32).reshape(12))
    _lib.check(_lib.load().smx_synth_render_room(
        _sv(stream), _d(depth_out), _d(color_out), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
        T.ctypes.data_as(C.c_void_p), C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32(frame_index),
        C.c_float(depth_scaling), C.c_float(noise_sigma), C.c_float(dropout)))


# ---- GPU -> CPU hand-off types (APP/cuda_surfels_cpu.h) ---------------------------------------
class CUDASurfelBuffersCPU:
    def __init__(self, max_surfel_count):
        n = int(max_surfel_count)
        self.frame_index = 0
        self.surfel_count = 0
        self.surfel_x_buffer = np.empty(n, np.float32)
        self.surfel_y_buffer = np.empty(n, np.float32)
        self.surfel_z_buffer = np.empty(n, np.float32)
        self.surfel_radius_squared_buffer = np.empty(n, np.float32)
        self.surfel_normal_x_buffer = np.empty(n, np.float32)
        self.surfel_normal_y_buffer = np.empty(n, np.float32)
        self.surfel_normal_z_buffer = np.empty(n, np.float32)
        self.surfel_last_update_stamp_buffer = np.empty(n, np.uint32)

    def _pod(self):
        p = SurfelBuffersCPU()
        for name, _ in SurfelBuffersCPU._fields_[2:]:
            setattr(p, name, getattr(self, name).ctypes.data)
        return p


class CUDASurfelDeltaCPU:
    """Changed surfels: slot indices (ascending) and the eight attributes TransferAllToCPU moves, for those slots."""
    ROWS = (("x", np.float32, "surfel_x_buffer"), ("y", np.float32, "surfel_y_buffer"), ("z", np.float32, "surfel_z_buffer"),
            ("radius_squared", np.float32, "surfel_radius_squared_buffer"),
            ("normal_x", np.float32, "surfel_normal_x_buffer"), ("normal_y", np.float32, "surfel_normal_y_buffer"),
            ("normal_z", np.float32, "surfel_normal_z_buffer"),
            ("last_update_stamp", np.uint32, "surfel_last_update_stamp_buffer"))

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.count = 0
        self.frame_index = 0
        self.surfel_count = 0
        self.surfel_index = np.empty(self.capacity, np.uint32)
        for name, dt, _ in self.ROWS:
            setattr(self, name, np.empty(self.capacity, dt))

    def _pod(self):
        from ._lib import SurfelDeltaCPU
        p = SurfelDeltaCPU()
        p.capacity = self.capacity
        p.surfel_index = self.surfel_index.ctypes.data
        for name, _, _ in self.ROWS:
            setattr(p, name, getattr(self, name).ctypes.data)
        return p

    def ApplyTo(self, buffers):
        """Patch a CUDASurfelBuffersCPU (a previous full transfer + all deltas since) to the current state."""
        idx = self.surfel_index[:self.count]
        for name, _, full in self.ROWS:
            getattr(buffers, full)[idx] = getattr(self, name)[:self.count]
        buffers.frame_index = self.frame_index
        buffers.surfel_count = self.surfel_count


class CUDASurfelsCPU:
    """Mutex-guarded write/read double buffer, APP/cuda_surfels_cpu.h:83-124."""

    def __init__(self, max_surfel_count):
        self._write = CUDASurfelBuffersCPU(max_surfel_count)
        self._read = CUDASurfelBuffersCPU(max_surfel_count)
        self._lock = threading.Lock()
        self._debug_wrote_data = False

    def LockWriteBuffers(self):
        self._lock.acquire()

    def UnlockWriteBuffers(self):
        self._debug_wrote_data = True
        self._lock.release()

    def WaitForLockAndSwapBuffers(self):
        with self._lock:
            if not self._debug_wrote_data:
                # LOG(FATAL) in the reference (:109-111)
                raise SmxError("Trying to swap the CUDASurfelsCPU buffers, but no data was written. "
                               "Possible multi-threading bug!")
            self._write, self._read = self._read, self._write
            self._debug_wrote_data = False

    def write_buffers(self):
        return self._write

    def read_buffers(self):
        return self._read


# ---- CUDASurfelReconstruction (APP/cuda_surfel_reconstruction.h) ------------------------------
class PinholeCamera4f:
    """The accessors of VIS/camera.h's PinholeCamera4f that the hot path uses."""

    def __init__(self, width, height, fx, fy, cx, cy):
        self._w, self._h = int(width), int(height)
        self._p = (float(fx), float(fy), float(cx), float(cy))

    def width(self):
        return self._w

    def height(self):
        return self._h

    def parameters(self):
        return self._p

    def Scaled(self, factor):
        """Camera::Scaled (VIS/camera.h:1564-1574): size = factor * size + 0.5 truncated, the four pinhole parameters
        times factor in float (PinholeProjection::ScaleParameters, camera.h:954-964; origin at the image corner)."""
        f = np.float32(factor)
        return PinholeCamera4f(int(factor * self._w + np.float32(0.5)), int(factor * self._h + np.float32(0.5)),
                               *[float(np.float32(v) * f) for v in self._p])


class CUDASurfelReconstruction:
    def __init__(self, max_surfel_count, depth_camera, vertex_buffer_resource=None,
                 neighbor_index_buffer_resource=None, normal_vertex_buffer_resource=None, render_window=None,
                 device_id=-1):
        """device_id (not in the reference's constructor): the GPU the object lives on, -1 = the current device."""
        _lib.require_gpu()
        self.max_surfel_count = int(max_surfel_count)
        self.depth_camera = depth_camera
        fx, fy, cx, cy = depth_camera.parameters()
        self._h = C.c_void_p()
        _lib.check(_lib.load().smx_recon_create(C.c_uint32(self.max_surfel_count), depth_camera.width(),
                                                depth_camera.height(), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                                C.c_float(cy), C.c_int32(device_id), C.byref(self._h)))
        self._last_stream = None

    def Integrate(self, stream, frame_index, depth_scaling, depth_buffer, normals_buffer, radius_buffer, color_buffer,
                  global_T_local, sensor_noise_factor, max_surfel_confidence, regularizer_weight,
                  regularization_frame_window_size, do_blending, measurement_blending_radius,
                  regularization_iterations_per_integration_iteration, radius_factor_for_regularization_neighbors,
                  normal_compatibility_threshold_deg, surfel_integration_active_window_size):
        p = IntegrateParams(sensor_noise_factor, max_surfel_confidence, regularizer_weight,
                            regularization_frame_window_size, 1 if do_blending else 0, measurement_blending_radius,
                            regularization_iterations_per_integration_iteration,
                            radius_factor_for_regularization_neighbors, normal_compatibility_threshold_deg,
                            surfel_integration_active_window_size)
        self.IntegrateP(stream, frame_index, depth_scaling, depth_buffer, normals_buffer, radius_buffer, color_buffer,
                        global_T_local, p)

    def IntegrateP(self, stream, frame_index, depth_scaling, depth_buffer, normals_buffer, radius_buffer,
                   color_buffer, global_T_local, params):
        T = np.ascontiguousarray(np.asarray(global_T_local, np.float32).reshape(12))
        self._last_stream = stream
        _lib.check(_lib.load().smx_recon_integrate(
            self._h, _sv(stream), C.c_uint32(frame_index), C.c_float(depth_scaling), _d(depth_buffer),
            _d(normals_buffer), _d(radius_buffer), _d(color_buffer), T.ctypes.data_as(C.c_void_p), C.byref(params)))

    def Regularize(self, stream, frame_index, regularizer_weight, radius_factor_for_regularization_neighbors,
                   regularization_frame_window_size):
        _lib.check(_lib.load().smx_recon_regularize(self._h, _sv(stream), C.c_uint32(frame_index),
                                                    C.c_float(regularizer_weight),
                                                    C.c_float(radius_factor_for_regularization_neighbors),
                                                    C.c_int32(regularization_frame_window_size)))

    def TransferAllToCPU(self, stream, frame_index, buffers):
        """Requires the caller to hold buffers.LockWriteBuffers() (APP/main.cc:1261-1264)."""
        wb = buffers.write_buffers()
        pod = wb._pod()
        _lib.check(_lib.load().smx_recon_transfer_all_to_cpu(self._h, _sv(stream), C.c_uint32(frame_index),
                                                             C.byref(pod)))
        wb.frame_index = pod.frame_index
        wb.surfel_count = pod.surfel_count

    def SetDeltaTracking(self, stream, enabled):
        """Not in the reference: mark the slots whose transferred attributes change, for
        TransferChangedToCPU.  Enabling marks every existing slot."""
        _lib.check(_lib.load().smx_recon_set_delta_tracking(self._h, _sv(stream), C.c_int32(1 if enabled else 0)))

    def TransferChangedToCPU(self, stream, frame_index, capacity=None, delta=None):
        """The changed-surfel delta since the previous call (synchronous): a CUDASurfelDeltaCPU (pass `delta` to
        reuse one instead of allocating capacity-sized arrays per call)."""
        cap = int(capacity if capacity is not None else self.max_surfel_count)
        d = delta if delta is not None else CUDASurfelDeltaCPU(cap)
        pod = d._pod()
        rc = _lib.load().smx_recon_transfer_changed_to_cpu(self._h, _sv(stream), C.c_uint32(frame_index), C.byref(pod))
        d.count, d.frame_index, d.surfel_count = pod.count, pod.frame_index, pod.surfel_count
        if rc != 0:
            d.count_needed = pod.count
            d.count = 0
            self.last_failed_delta = d
            _lib.check(rc)
        return d

    def DeformByCreationFrame(self, stream, frame_T, reactivate=None, frame_index=0):
        """The loop-closure hook the reference describes but does not ship (README.md:152-176): surfels created at
        frame c move by the rigid correction frame_T[c] ([n_frames, 3, 4] / [n_frames, 12]); reactivate[c] != 0
        re-stamps them with frame_index."""
        T = np.ascontiguousarray(np.asarray(frame_T, np.float32).reshape(-1, 12))
        ra = np.ascontiguousarray(reactivate, np.uint8) if reactivate is not None else None
        if ra is not None and ra.size != T.shape[0]:
            raise ValueError("reactivate needs one byte per frame")
        _lib.check(_lib.load().smx_recon_deform_by_creation_frame(
            self._h, _sv(stream), T.ctypes.data_as(C.c_void_p), C.c_uint32(T.shape[0]),
            ra.ctypes.data_as(C.c_void_p) if ra is not None else C.c_void_p(0), C.c_uint32(frame_index), C.c_int32(0)))

    def CheckTrianglesForRemeshing(self, stream, triangles, long_edge_total_factor_squared):
        """The per-triangle tests of SurfelMeshing::CheckRemeshing (APP/surfel_meshing.cc:590-650) for triangles
        [T,3] of slot indices against the device-resident map.  Returns flags [T] (bits: see smx.h)."""
        tri = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        flags = np.zeros(tri.shape[0], np.uint8)
        _lib.check(_lib.load().smx_recon_check_triangles(
            self._h, _sv(stream), tri.ctypes.data_as(C.c_void_p), C.c_uint32(tri.shape[0]),
            C.c_float(long_edge_total_factor_squared), flags.ctypes.data_as(C.c_void_p), C.c_int32(0)))
        return flags

    def UpdateVisualizationBuffers(self, *args, **kwargs):
        """Viewer-only in the reference (OpenGL interop); nothing to do without a render window."""

    def ExportVertices(self, stream, position_buffer, color_buffer):
        _lib.check(_lib.load().smx_recon_export_vertices(self._h, _sv(stream), _d(position_buffer), _d(color_buffer)))

    def GetTimings(self):
        """(data_association, surfel_merging, measurement_blending, integration, neighbor_update,
        new_surfel_creation, regularization) in ms."""
        out = (C.c_float * 7)()
        _lib.check(_lib.load().smx_recon_get_timings(self._h, out))
        return tuple(out)

    def debug_stamp_ring(self):
        """(records [8][16] uint64, wall clock kHz): smx_recon_debug_stamp_ring."""
        out = np.zeros((8, 16), np.uint64)
        khz = C.c_int32(0)
        _lib.check(_lib.load().smx_recon_debug_stamp_ring(self._h, out.ctypes.data_as(C.c_void_p), C.c_int32(out.size), C.byref(khz)))
        return out, int(khz.value)

    def GetTimingsNoWait(self):
        """(the seven stage times in ms, call number) of the newest Integrate call that is known to be through, without
        waiting for the last one (smx_recon_get_timings_nowait); call number 0 = none yet."""
        out = (C.c_float * 7)()
        call = C.c_uint64(0)
        _lib.check(_lib.load().smx_recon_get_timings_nowait(self._h, out, C.byref(call)))
        return tuple(out), int(call.value)

    def _counts(self):
        a, b = C.c_uint32(), C.c_uint32()
        _lib.check(_lib.load().smx_recon_counts(self._h, _sv(self._last_stream), C.byref(a), C.byref(b)))
        return a.value, b.value

    def surfel_count(self):
        return self._counts()[0]

    def surfels_size(self):
        return self._counts()[1]

    # -- extras (not in the reference interface)
    def stats(self):
        s = ReconStats()
        _lib.check(_lib.load().smx_recon_get_stats(self._h, _sv(self._last_stream), C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in ReconStats._fields_}

    def set_timing_enabled(self, enabled):
        """False/0 = off, True/1 = the reference's stage events, 3 = stage events + per-kernel events."""
        _lib.check(_lib.load().smx_recon_set_timing_enabled(self._h, C.c_int32(int(enabled))))

    @staticmethod
    def kernel_time_names():
        L = _lib.load()
        return [L.smx_recon_kernel_slot_name(i).decode() for i in range(L.smx_recon_kernel_slot_count())]

    def kernel_times_ms(self):
        n = _lib.load().smx_recon_kernel_slot_count()
        out = (C.c_float * n)()
        _lib.check(_lib.load().smx_recon_get_kernel_timings(self._h, out, C.c_int32(n)))
        return list(out)

    def profile_begin(self, kernel_name, max_frames):
        _lib.check(_lib.load().smx_recon_profile_begin(self._h, C.c_int32(self.kernel_time_names().index(kernel_name)),
                                                       C.c_int32(max_frames)))

    def profile_end(self):
        ms, n = C.c_float(), C.c_int32()
        _lib.check(_lib.load().smx_recon_profile_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_stats_enabled(self, enabled):
        _lib.check(_lib.load().smx_recon_set_stats_enabled(self._h, C.c_int32(1 if enabled else 0)))

    def set_scan_mode(self, mode):
        _lib.check(_lib.load().smx_recon_set_scan_mode(self._h, C.c_int32(mode)))

    def set_handover_mode(self, mode):
        """smx_recon_set_handover_mode: 1 = device word + gate kernel (default), 0 = event"""
        _lib.check(_lib.load().smx_recon_set_handover_mode(self._h, C.c_int32(int(mode))))

    def handover_mode(self):
        m = C.c_int32(-1)
        _lib.check(_lib.load().smx_recon_get_handover_mode(self._h, C.byref(m)))
        return int(m.value)

    def set_internal_cu_mask(self, mask_words):
        """experiment (smx_recon_set_internal_cu_mask): the internal stream on the compute units of the mask (empty = all)"""
        arr = (C.c_uint32 * max(1, len(mask_words)))(*mask_words)
        _lib.check(_lib.load().smx_recon_set_internal_cu_mask(self._h, arr, C.c_uint32(len(mask_words))))

    def debug_set_skip(self, mask):
        """TIMING ONLY (smx_recon_debug_set_skip): bit 0 = no regulariser, bit 1 = front of the frame only."""
        _lib.check(_lib.load().smx_recon_debug_set_skip(self._h, C.c_int32(mask)))

    def set_overlap(self, enabled):
        """Frame pipelining on/off (regulariser of frame f beside the first kernels of frame f+1)."""
        _lib.check(_lib.load().smx_recon_set_overlap(self._h, C.c_int32(1 if enabled else 0)))

    def debug_download_surfels(self, count=None):
        n = self.surfels_size() if count is None else int(count)
        rows = np.zeros((kSurfelAttributeCount, n), np.float32)
        _lib.check(_lib.load().smx_recon_debug_download_surfels(self._h, _sv(self._last_stream),
                                                                rows.ctypes.data_as(C.c_void_p), C.c_uint32(n)))
        return rows

    def debug_upload_surfels(self, rows, merge_count=0):
        rows = np.ascontiguousarray(rows, np.float32)
        assert rows.ndim == 2 and rows.shape[0] == kSurfelAttributeCount
        _lib.check(_lib.load().smx_recon_debug_upload_surfels(self._h, _sv(self._last_stream),
                                                              rows.ctypes.data_as(C.c_void_p),
                                                              C.c_uint32(rows.shape[1]), C.c_uint32(merge_count)))

    _SCRATCH = {"supporting": (0, np.uint32), "support_counts": (1, np.uint32), "depth_sums_q": (2, np.int64),
                "conflicting": (3, np.uint32), "first_depth": (4, np.float32), "new_flags": (5, np.uint8),
                "new_indices": (6, np.uint32)}

    def debug_count_skipped_segments(self):
        out = C.c_uint32(0)
        _lib.check(_lib.load().smx_recon_debug_count_skipped_segments(self._h, _sv(self._last_stream), C.byref(out)))
        return int(out.value)

    def debug_download_scratch(self, name):
        which, dt = self._SCRATCH[name]
        out = np.empty((self.depth_camera.height(), self.depth_camera.width()), dt)
        _lib.check(_lib.load().smx_recon_debug_download_scratch(self._h, _sv(self._last_stream), C.c_int32(which),
                                                                out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().smx_recon_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- radius-neighbor search -------------------------------------------------------------------
class SurfelNeighborIndex:
    """Batched replacement of CompressedOctree::FindNearestSurfelsWithinRadius (APP/octree.h:470-477):
    uniform-grid index rebuilt from the surfel position rows, queried for many positions at once."""

    def __init__(self, device_id=-1):
        _lib.require_gpu()
        self._h = C.c_void_p()
        _lib.check(_lib.load().smx_nn_create(C.c_int32(device_id), C.byref(self._h)))

    def Build(self, x, y, z, cell_size, stream=None):
        x, y, z = (np.ascontiguousarray(a, np.float32) for a in (x, y, z))
        self._keep = (x, y, z)
        _lib.check(_lib.load().smx_nn_build(self._h, _sv(stream), x.ctypes.data_as(C.c_void_p),
                                            y.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                                            C.c_uint32(x.size), C.c_float(cell_size), C.c_int32(0)))

    def FindNearestSurfelsWithinRadius(self, positions, radius_squared, max_result_count, state=None, skip_mask=0,
                                       stream=None):
        """positions [nq,3]; radius_squared scalar or [nq].  Returns (counts [nq], dist2 [nq,K], indices [nq,K])."""
        q = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        nq = q.shape[0]
        qx, qy, qz = (np.ascontiguousarray(q[:, i]) for i in range(3))
        r2 = np.ascontiguousarray(np.broadcast_to(np.asarray(radius_squared, np.float32), (nq,)))
        k = int(max_result_count)
        idx = np.zeros((nq, k), np.uint32)
        d2 = np.zeros((nq, k), np.float32)
        cnt = np.zeros(nq, np.int32)
        st = np.ascontiguousarray(state, np.uint8) if state is not None else None
        _lib.check(_lib.load().smx_nn_query_batch(
            self._h, _sv(stream), C.c_uint32(nq), qx.ctypes.data_as(C.c_void_p), qy.ctypes.data_as(C.c_void_p),
            qz.ctypes.data_as(C.c_void_p), r2.ctypes.data_as(C.c_void_p), C.c_int32(k),
            st.ctypes.data_as(C.c_void_p) if st is not None else C.c_void_p(0), C.c_uint8(skip_mask), C.c_int32(0),
            idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
            C.c_int32(0)))
        return cnt, d2, idx

    def BuildFromReconstruction(self, reconstruction, cell_size, stream=None):
        """Index over the smooth positions of all slots of the device-resident map, merged slots left out; no host
        round trip.  A snapshot: rebuild after Integrate / Regularize."""
        self._keep = None
        _lib.check(_lib.load().smx_recon_build_neighbor_index(reconstruction._h, _sv(stream), self._h,
                                                              C.c_float(cell_size)))

    def FindNeighborCandidates(self, reconstruction, surfel_indices, radius_factor_squared, max_result_count,
                               state=None, skip_mask=0, stream=None):
        """Candidate lists of SurfelMeshing::TriangulateSurfel (APP/surfel_meshing.cc:417-425) for a batch of slots:
        ball = radius_factor_squared * radius_squared of the slot around its smooth position, read on the device.
        Returns (counts [n], dist2 [n,K], indices [n,K])."""
        sl = np.ascontiguousarray(surfel_indices, np.uint32).reshape(-1)
        n, k = sl.size, int(max_result_count)
        idx = np.zeros((n, k), np.uint32)
        d2 = np.zeros((n, k), np.float32)
        cnt = np.zeros(n, np.int32)
        st = np.ascontiguousarray(state, np.uint8) if state is not None else None
        _lib.check(_lib.load().smx_recon_neighbor_candidates(
            reconstruction._h, _sv(stream), self._h, sl.ctypes.data_as(C.c_void_p), C.c_uint32(n),
            C.c_float(radius_factor_squared), C.c_int32(k),
            st.ctypes.data_as(C.c_void_p) if st is not None else C.c_void_p(0), C.c_uint8(skip_mask), C.c_int32(0),
            idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
            C.c_int32(0)))
        return cnt, d2, idx

    def FindNearestOfIndexedPoints(self, n_points, max_result_count, radius_squared=None, factor=1.0, state=None,
                                   skip_mask=0, stream=None):
        """Every indexed point queries its own neighbourhood (smx_nn_query_self): r^2 = factor * radius_squared[i], or
        r^2 = factor for all if radius_squared is None.  n_points (optional check) = the number of points given to Build.  Returns
        (counts [n], dist2 [n,K], indices [n,K]); device staging is allocated here (the C entry point takes device
        pointers only)."""
        n, k = int(self.stats()["n_points"]), int(max_result_count)   # (the C entry point writes one row per point of the BUILD)
        if n_points is not None and int(n_points) != n:
            raise ValueError("n_points = %d, but the index was built over %d points" % (int(n_points), n))
        didx, dd2, dcnt = CUDABuffer(1, n * k, np.uint32), CUDABuffer(1, n * k, np.float32), CUDABuffer(1, n, np.int32)
        dr2 = dst = None
        if radius_squared is not None:
            dr2 = CUDABuffer(1, n, np.float32)
            dr2.UploadAsync(stream, np.ascontiguousarray(radius_squared, np.float32).reshape(1, n))
        if state is not None:
            dst = CUDABuffer(1, n, np.uint8)
            dst.UploadAsync(stream, np.ascontiguousarray(state, np.uint8).reshape(1, n))
        _lib.check(_lib.load().smx_nn_query_self(
            self._h, _sv(stream), C.c_void_p(dr2.ToCUDA().address if dr2 else 0), C.c_float(factor), C.c_int32(k),
            C.c_void_p(dst.ToCUDA().address if dst else 0), C.c_uint8(skip_mask), C.c_void_p(didx.ToCUDA().address),
            C.c_void_p(dd2.ToCUDA().address), C.c_void_p(dcnt.ToCUDA().address)))
        cnt = dcnt.Download(stream)[0].copy()
        d2 = dd2.Download(stream)[0].reshape(n, k).copy()
        idx = didx.Download(stream)[0].reshape(n, k).copy()
        for b in (didx, dd2, dcnt, dr2, dst):
            if b is not None:
                b.close()
        return cnt, d2, idx

    def set_query_mode(self, mode):
        """A/B switch (results identical): 0 = LDS-staged brick tiles, 1 = one wavefront per query through L1 / L2."""
        _lib.check(_lib.load().smx_nn_set_query_mode(self._h, C.c_int32(mode)))

    def set_stats_enabled(self, enabled, stream=None):
        _lib.check(_lib.load().smx_nn_set_stats_enabled(self._h, _sv(stream), C.c_int32(1 if enabled else 0)))

    def stats(self, stream=None):
        """Index geometry and, while enabled, the tile / candidate / test / result counters of the queries (smx_nn_stats)."""
        st = _lib.NNStats()
        _lib.check(_lib.load().smx_nn_get_stats(self._h, _sv(stream), C.byref(st)))
        d = {n: getattr(st, n) for n, _ in _lib.NNStats._fields_ if n != "dim"}
        d["dim"] = [int(v) for v in st.dim]
        return d

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().smx_nn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
