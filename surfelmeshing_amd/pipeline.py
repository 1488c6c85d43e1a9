"""Per-frame driver that mimics the reference's caller (APP/main.cc:1015-1223):
bilateral filter -> multi-frame outlier cull -> erosion -> normals -> radii -> Integrate,
with the same ping-pong of the two filtered depth buffers (A, B).

This is host plumbing around the C-ABI calls; all computing happens in libsmx.so.
"""
from dataclasses import dataclass

import numpy as np

from . import api
from ._lib import IntegrateParams


@dataclass
class PreprocessParams:
    """Defaults: APP/main.cc:279, 415-475."""
    depth_scaling: float = 5000.0
    max_depth: float = 3.0
    depth_valid_region_radius: float = 333.0
    observation_angle_threshold_deg: float = 85.0
    depth_erosion_radius: int = 2
    outlier_filtering_frame_count: int = 8
    outlier_filtering_required_inliers: int = -1
    bilateral_filter_sigma_xy: float = 3.0
    bilateral_filter_radius_factor: float = 2.0
    bilateral_filter_sigma_depth_factor: float = 0.05
    outlier_filtering_depth_tolerance_factor: float = 0.02
    point_radius_extension_factor: float = 1.5
    point_radius_clamp_factor: float = float("inf")
    median_filter_and_densify_iterations: int = 0   # main.cc:929-939 (on the CPU there, on the GPU here)
    pyramid_level: int = 0                          # main.cc:299-303, 751, 941-981: work at 1 / 2^level of the input size

    def max_depth_u16(self):
        return int(np.uint16(min(self.depth_scaling * self.max_depth, 65535.0)))  # main.cc:1021


def others_TR_reference(global_T_reference, global_T_others, depth_scaling):
    """The relative poses of the outlier-cull neighbours as APP/main.cc:1037-1059 forms them: translations scaled to
    depth units, (reference_scaled_frame_T_global * global_T_other_scaled)^-1, one 3x4 per neighbour (float32).
    Inputs: 3x4 global_T_frame matrices."""
    G = np.asarray(global_T_reference, np.float64).reshape(3, 4)
    Rr, tr = G[:, :3], G[:, 3]
    out = []
    for O in global_T_others:
        O = np.asarray(O, np.float64).reshape(3, 4)
        Ro, to = O[:, :3], O[:, 3]
        R = Rr.T @ Ro
        t = Rr.T @ (to * depth_scaling) - Rr.T @ (tr * depth_scaling)
        Ri = R.T
        out.append(np.concatenate([Ri, (-Ri @ t)[:, None]], axis=1))
    return np.asarray(out, np.float32).reshape(len(out), 3, 4)


class FramePipeline:
    def __init__(self, width, height, fx, fy, cx, cy, max_surfel_count, pre=None, params=None, stream=None):
        """width, height and the intrinsics describe the input frames; with pre.pyramid_level > 0 the pipeline works
        on the scaled camera (main.cc:751) and upload() reduces the frames on the GPU."""
        self.pre = pre or PreprocessParams()
        self.in_w, self.in_h = width, height
        self.camera = api.PinholeCamera4f(width, height, fx, fy, cx, cy)
        if self.pre.pyramid_level > 0:
            if self.pre.median_filter_and_densify_iterations > 0:
                raise ValueError("Simultaneous downscaling and median filtering of depth maps is not implemented "
                                 "(main.cc:944-947)")
            if width % (1 << self.pre.pyramid_level) or height % (1 << self.pre.pyramid_level):
                raise ValueError("the input size must be divisible by 2^pyramid_level (VIS/image.h:930-931)")
            self.camera = self.camera.Scaled(1.0 / 2 ** self.pre.pyramid_level)
        width, height = self.camera.width(), self.camera.height()
        fx, fy, cx, cy = self.camera.parameters()
        self.w, self.h = width, height
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.params = params or IntegrateParams.defaults()
        self.stream = stream
        self.reconstruction = api.CUDASurfelReconstruction(max_surfel_count, self.camera)
        self.filtered_A = api.CUDABuffer(height, width, np.uint16)
        self.filtered_B = api.CUDABuffer(height, width, np.uint16)
        self.normals = api.CUDABuffer(height, width, np.float32, 2)
        self.radius = api.CUDABuffer(height, width, np.float32)
        self.radius.Clear(0.0, stream)
        self.raw_depth = {}   # frame index -> CUDABuffer<u16>, raw uploads (main.cc:905-968)
        self.color = {}       # frame index -> CUDABuffer<Vec3u8>

    def upload(self, frame_index, depth, color):
        d = api.CUDABuffer(self.in_h, self.in_w, np.uint16)
        d.UploadAsync(self.stream, depth)
        c = api.CUDABuffer(self.in_h, self.in_w, np.uint8, 3)
        c.UploadAsync(self.stream, color)
        if self.pre.median_filter_and_densify_iterations > 0:   # main.cc:929-939, per frame at load time
            t = api.CUDABuffer(self.h, self.w, np.uint16)
            for _ in range(self.pre.median_filter_and_densify_iterations):
                api.MedianFilterAndDensifyDepthMapCUDA(self.stream, d, t)
                d, t = t, d
            api.StreamSynchronize(self.stream)
            t.close()
        if self.pre.pyramid_level > 0:   # main.cc:941-962 (depth), 973-981 (colour); CPU loops in the reference
            ds, cs = api.CUDABuffer(self.h, self.w, np.uint16), api.CUDABuffer(self.h, self.w, np.uint8, 3)
            api.DownscaleUsingMedianWhileExcludingCUDA(self.stream, 0, d, ds)
            api.ColorImagePyramidCUDA(self.stream, self.pre.pyramid_level, c, cs)
            api.StreamSynchronize(self.stream)
            d.close()
            c.close()
            d, c = ds, cs
        self.raw_depth[frame_index] = d
        self.color[frame_index] = c

    def release(self, frame_index):
        for m in (self.raw_depth, self.color):
            b = m.pop(frame_index, None)
            if b is not None:
                b.close()

    def preprocess(self, frame_index, other_frames, others_TR_reference):
        """The five preprocessing calls, APP/main.cc:1015-1191.  Result: depth in filtered_A,
        normals, radius."""
        s, p = self.stream, self.pre
        api.BilateralFilteringAndDepthCutoffCUDA(
            s, p.bilateral_filter_sigma_xy, p.bilateral_filter_sigma_depth_factor, 0,
            p.bilateral_filter_radius_factor, p.max_depth_u16(), p.depth_valid_region_radius,
            self.raw_depth[frame_index], self.filtered_A)
        if other_frames:
            api.OutlierDepthMapFusionCUDA(
                s, p.outlier_filtering_depth_tolerance_factor, self.filtered_A, self.fx, self.fy, self.cx, self.cy,
                [self.raw_depth[g] for g in other_frames], others_TR_reference, self.filtered_B,
                required_count=(-1 if p.outlier_filtering_required_inliers in (-1, len(other_frames))
                                else p.outlier_filtering_required_inliers))
            src, dst = self.filtered_B, self.filtered_A
        else:
            src, dst = self.filtered_A, self.filtered_B
        if p.depth_erosion_radius > 0:
            api.ErodeDepthMapCUDA(s, p.depth_erosion_radius, src, dst)
        else:
            api.CopyWithoutBorderCUDA(s, src, dst)
        src, dst = dst, src
        api.ComputeNormalsAndDropBadPixelsCUDA(s, p.observation_angle_threshold_deg, p.depth_scaling, self.fx, self.fy,
                                               self.cx, self.cy, src, dst, self.normals)
        src, dst = dst, src
        api.ComputePointRadiiAndRemoveIsolatedPixelsCUDA(s, p.point_radius_extension_factor,
                                                         p.point_radius_clamp_factor, p.depth_scaling, self.fx,
                                                         self.fy, self.cx, self.cy, src, self.radius, dst)
        self.depth_final = dst
        return dst

    def integrate(self, frame_index, global_T_frame):
        self.reconstruction.IntegrateP(self.stream, frame_index, self.pre.depth_scaling, self.depth_final,
                                       self.normals, self.radius, self.color[frame_index], global_T_frame, self.params)

    def integrate_as(self, frame_index, data_frame, global_T_frame):
        """Integrate the preprocessed images of `data_frame` under the frame index (stamp) `frame_index`
        (revisiting a pose with a stored frame)."""
        self.reconstruction.IntegrateP(self.stream, frame_index, self.pre.depth_scaling, self.depth_final,
                                       self.normals, self.radius, self.color[data_frame], global_T_frame, self.params)

    def process(self, frame_index, other_frames, others_TR_reference, global_T_frame):
        self.preprocess(frame_index, other_frames, others_TR_reference)
        self.integrate(frame_index, global_T_frame)


# ---- native driver (include/smx_driver.h): the same per-frame sequence in C++ -----------------------
import ctypes as _C

from . import _lib as _smxlib


class DriverConfig(_C.Structure):
    _fields_ = [("width", _C.c_int32), ("height", _C.c_int32),
                ("fx", _C.c_float), ("fy", _C.c_float), ("cx", _C.c_float), ("cy", _C.c_float),
                ("max_surfel_count", _C.c_uint32),
                ("depth_scaling", _C.c_float), ("max_depth", _C.c_float), ("depth_valid_region_radius", _C.c_float),
                ("observation_angle_threshold_deg", _C.c_float), ("depth_erosion_radius", _C.c_int32),
                ("outlier_filtering_required_inliers", _C.c_int32),
                ("bilateral_filter_sigma_xy", _C.c_float), ("bilateral_filter_radius_factor", _C.c_float),
                ("bilateral_filter_sigma_depth_factor", _C.c_float),
                ("outlier_filtering_depth_tolerance_factor", _C.c_float),
                ("point_radius_extension_factor", _C.c_float), ("point_radius_clamp_factor", _C.c_float),
                ("integrate", IntegrateParams)]


class DriverStep(_C.Structure):
    _fields_ = [("frame_index", _C.c_uint32), ("other_count", _C.c_int32), ("other_frames", _C.c_uint32 * 8),
                ("others_TR_reference", (_C.c_float * 12) * 8), ("global_T_frame", _C.c_float * 12)]


DRIVER_EXPORTS = ["smx_driver_create", "smx_driver_destroy", "smx_driver_recon", "smx_driver_upload_frame",
                  "smx_driver_render_frame", "smx_driver_release_frame", "smx_driver_frame_descs", "smx_driver_run",
                  "smx_driver_work_descs", "smx_driver_download_frame", "smx_driver_download_work", "smx_driver_set_overlap", "smx_driver_set_fused_tail", "smx_driver_set_fused_head", "smx_driver_set_run_ahead", "smx_driver_set_split_preprocessing", "smx_driver_set_pre_cu_mask",
                  "smx_driver_run_streamed", "smx_driver_debug_streams", "smx_driver_set_staged_uploads", "smx_driver_upload_counts", "smx_driver_set_read_timings", "smx_driver_timing_sums", "smx_driver_debug_prepare", "smx_driver_profile_begin", "smx_driver_profile_end"]


class DriverHostFrame(_C.Structure):
    _fields_ = [("frame_index", _C.c_uint32), ("depth", _C.c_void_p), ("color", _C.c_void_p)]


class _BorrowedRecon(api.CUDASurfelReconstruction):
    """CUDASurfelReconstruction view of the object the native driver owns."""

    def __init__(self, handle, camera, max_surfel_count):  # noqa: super().__init__ would create a new object
        self._h = handle
        self.depth_camera = camera
        self.max_surfel_count = int(max_surfel_count)
        self._last_stream = None

    def close(self):
        self._h = _C.c_void_p()


class NativeFramePipeline:
    """FramePipeline twin whose frame loop runs in C++ (smx_driver_run): one ctypes call enqueues many frames."""

    def __init__(self, width, height, fx, fy, cx, cy, max_surfel_count, pre=None, params=None, stream=None):
        L = _smxlib.load()
        _smxlib.require_gpu()
        for n in DRIVER_EXPORTS:
            getattr(L, n).restype = _C.c_int
        self.w, self.h, self.fx, self.fy, self.cx, self.cy = width, height, fx, fy, cx, cy
        self.pre = pre or PreprocessParams()
        self.params = params or IntegrateParams.defaults()
        self.stream = stream
        p = self.pre
        cfg = DriverConfig(width, height, fx, fy, cx, cy, max_surfel_count, p.depth_scaling, p.max_depth,
                           p.depth_valid_region_radius, p.observation_angle_threshold_deg, p.depth_erosion_radius,
                           p.outlier_filtering_required_inliers, p.bilateral_filter_sigma_xy,
                           p.bilateral_filter_radius_factor, p.bilateral_filter_sigma_depth_factor,
                           p.outlier_filtering_depth_tolerance_factor, p.point_radius_extension_factor,
                           p.point_radius_clamp_factor, self.params)
        self._d = _C.c_void_p()
        _smxlib.check(L.smx_driver_create(_C.byref(cfg), _C.byref(self._d)))
        rh = _C.c_void_p()
        _smxlib.check(L.smx_driver_recon(self._d, _C.byref(rh)))
        self.reconstruction = _BorrowedRecon(rh, api.PinholeCamera4f(width, height, fx, fy, cx, cy), max_surfel_count)
        self.resident = set()

    def set_overlap(self, enabled):
        _smxlib.check(_smxlib.load().smx_driver_set_overlap(self._d, _C.c_int32(1 if enabled else 0)))

    def set_fused_head(self, enabled):
        _smxlib.check(_smxlib.load().smx_driver_set_fused_head(self._d, _C.c_int32(1 if enabled else 0)))

    def set_split_preprocessing(self, enabled):
        _smxlib.check(_smxlib.load().smx_driver_set_split_preprocessing(self._d, _C.c_int32(1 if enabled else 0)))

    def set_pre_cu_mask(self, mask_words):
        """experiment: the preprocessing queues on the compute units of the mask (list of 32-bit words; empty = all)"""
        arr = (_C.c_uint32 * max(1, len(mask_words)))(*mask_words)
        _smxlib.check(_smxlib.load().smx_driver_set_pre_cu_mask(self._d, arr, _C.c_uint32(len(mask_words))))

    def set_fused_tail(self, enabled):
        _smxlib.check(_smxlib.load().smx_driver_set_fused_tail(self._d, _C.c_int32(1 if enabled else 0)))

    def set_staged_uploads(self, enabled):
        _smxlib.check(_smxlib.load().smx_driver_set_staged_uploads(self._d, _C.c_int32(1 if enabled else 0)))

    def upload_counts(self, reset=False):
        """(frames of run_streamed that took the staged route, frames that took the copy engine)"""
        a, b = _C.c_uint64(0), _C.c_uint64(0)
        _smxlib.check(_smxlib.load().smx_driver_upload_counts(self._d, _C.byref(a), _C.byref(b), _C.c_int32(1 if reset else 0)))
        return int(a.value), int(b.value)

    def set_read_timings(self, mode):
        """0 = off, 1 = GetTimingsNoWait after every Integrate of the native loop, 2 = the blocking GetTimings (APP/main.cc:1511)."""
        _smxlib.check(_smxlib.load().smx_driver_set_read_timings(self._d, _C.c_int32(int(mode))))

    def timing_sums(self, reset=True):
        sums = (_C.c_double * 7)()
        calls = _C.c_uint64(0)
        _smxlib.check(_smxlib.load().smx_driver_timing_sums(self._d, sums, _C.byref(calls), _C.c_int32(1 if reset else 0)))
        return [float(x) for x in sums], int(calls.value)

    def set_run_ahead(self, enabled):
        _smxlib.check(_smxlib.load().smx_driver_set_run_ahead(self._d, _C.c_int32(1 if enabled else 0)))

    def _s(self):
        return api._sv(self.stream)

    def upload(self, frame_index, depth, color):
        d = np.ascontiguousarray(depth, np.uint16)
        c = np.ascontiguousarray(color, np.uint8)
        _smxlib.check(_smxlib.load().smx_driver_upload_frame(self._d, self._s(), _C.c_uint32(frame_index),
                                                            d.ctypes.data_as(_C.c_void_p), c.ctypes.data_as(_C.c_void_p)))
        api.StreamSynchronize(self.stream)
        self.resident.add(frame_index)

    def render(self, frame_index, global_T_frame, seed, noise_sigma=0.001, dropout=0.01):
        T = np.ascontiguousarray(np.asarray(global_T_frame, np.float32).reshape(12))
        _smxlib.check(_smxlib.load().smx_driver_render_frame(self._d, self._s(), _C.c_uint32(frame_index),
                                                            T.ctypes.data_as(_C.c_void_p), _C.c_uint32(seed & 0xFFFFFFFF),
                                                            _C.c_float(noise_sigma), _C.c_float(dropout)))
        self.resident.add(frame_index)

    def release(self, frame_index):
        if frame_index in self.resident:
            _smxlib.check(_smxlib.load().smx_driver_release_frame(self._d, _C.c_uint32(frame_index)))
            self.resident.discard(frame_index)

    @staticmethod
    def make_step(frame_index, other_frames, others_TR_reference, global_T_frame):
        st = DriverStep()
        st.frame_index = frame_index
        st.other_count = len(other_frames)
        T = np.asarray(others_TR_reference, np.float32).reshape(len(other_frames), 12) if other_frames else None
        for i, g in enumerate(other_frames):
            st.other_frames[i] = g
            for k in range(12):
                st.others_TR_reference[i][k] = float(T[i, k])
        G = np.asarray(global_T_frame, np.float32).reshape(12)
        for k in range(12):
            st.global_T_frame[k] = float(G[k])
        return st

    def run(self, steps):
        """Enqueue a list of DriverStep (no synchronisation)."""
        arr = (DriverStep * len(steps))(*steps)
        _smxlib.check(_smxlib.load().smx_driver_run(self._d, self._s(), arr, _C.c_int32(len(steps))))

    def run_array(self, arr, n):
        _smxlib.check(_smxlib.load().smx_driver_run(self._d, self._s(), arr, _C.c_int32(n)))

    def prepare_array(self, arr, n):
        """(measurement) smx_driver_debug_prepare: the next run_array calls find these steps preprocessed."""
        _smxlib.check(_smxlib.load().smx_driver_debug_prepare(self._d, self._s(), arr, _C.c_int32(n)))

    def profile_begin(self, stage, max_frames):
        """(measurement) time stamps around preprocessing stage 0 / 1 / 2 (bilateral, outlier cull, erode + normals + radii)."""
        _smxlib.check(_smxlib.load().smx_driver_profile_begin(self._d, _C.c_int32(stage), _C.c_int32(max_frames)))

    def profile_end(self):
        ms, n = _C.c_float(), _C.c_int32()
        _smxlib.check(_smxlib.load().smx_driver_profile_end(self._d, _C.byref(ms), _C.byref(n)))
        return ms.value, n.value

    def run_streamed(self, steps, uploads):
        """smx_driver_run_streamed: steps (list of DriverStep), uploads = per step None or (frame_index, depth, color)
        host arrays -- page-locked (api.PagelockedArray) for copies that overlap the kernels -- which must stay alive
        until the streams are synchronised."""
        arr = (DriverStep * len(steps))(*steps)
        up = (DriverHostFrame * len(steps))()
        for i, u in enumerate(uploads):
            if u is None:
                continue
            f, d, c = u
            assert d.dtype == np.uint16 and d.shape == (self.h, self.w) and d.flags.c_contiguous
            assert c.dtype == np.uint8 and c.shape == (self.h, self.w, 3) and c.flags.c_contiguous
            up[i].frame_index, up[i].depth, up[i].color = f, d.ctypes.data, c.ctypes.data
            self.resident.add(f)
        _smxlib.check(_smxlib.load().smx_driver_run_streamed(self._d, self._s(), arr, up,
                                                            _C.c_int32(len(steps))))

    def process(self, frame_index, other_frames, others_TR_reference, global_T_frame):
        self.run([self.make_step(frame_index, other_frames, others_TR_reference, global_T_frame)])

    def download_frame(self, frame_index):
        d = np.empty((self.h, self.w), np.uint16)
        c = np.empty((self.h, self.w, 3), np.uint8)
        _smxlib.check(_smxlib.load().smx_driver_download_frame(self._d, self._s(), _C.c_uint32(frame_index),
                                                              d.ctypes.data_as(_C.c_void_p), c.ctypes.data_as(_C.c_void_p)))
        return d, c

    def download_work(self):
        """(final depth, normals, radius) after the last processed frame."""
        d = np.empty((self.h, self.w), np.uint16)
        n = np.empty((self.h, self.w, 2), np.float32)
        r = np.empty((self.h, self.w), np.float32)
        _smxlib.check(_smxlib.load().smx_driver_download_work(self._d, self._s(), d.ctypes.data_as(_C.c_void_p),
                                                             n.ctypes.data_as(_C.c_void_p), r.ctypes.data_as(_C.c_void_p)))
        return d, n, r

    def close(self):
        if getattr(self, "_d", None):
            self.reconstruction.close()
            _smxlib.load().smx_driver_destroy(self._d)
            self._d = _C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
