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

    def max_depth_u16(self):
        return int(np.uint16(min(self.depth_scaling * self.max_depth, 65535.0)))  # main.cc:1021


class FramePipeline:
    def __init__(self, width, height, fx, fy, cx, cy, max_surfel_count, pre=None, params=None, stream=None):
        self.w, self.h = width, height
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.pre = pre or PreprocessParams()
        self.params = params or IntegrateParams.defaults()
        self.stream = stream
        self.camera = api.PinholeCamera4f(width, height, fx, fy, cx, cy)
        self.reconstruction = api.CUDASurfelReconstruction(max_surfel_count, self.camera)
        self.filtered_A = api.CUDABuffer(height, width, np.uint16)
        self.filtered_B = api.CUDABuffer(height, width, np.uint16)
        self.normals = api.CUDABuffer(height, width, np.float32, 2)
        self.radius = api.CUDABuffer(height, width, np.float32)
        self.radius.Clear(0.0, stream)
        self.raw_depth = {}   # frame index -> CUDABuffer<u16>, raw uploads (main.cc:905-968)
        self.color = {}       # frame index -> CUDABuffer<Vec3u8>

    def upload(self, frame_index, depth, color):
        d = api.CUDABuffer(self.h, self.w, np.uint16)
        d.UploadAsync(self.stream, depth)
        c = api.CUDABuffer(self.h, self.w, np.uint8, 3)
        c.UploadAsync(self.stream, color)
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

    def process(self, frame_index, other_frames, others_TR_reference, global_T_frame):
        self.preprocess(frame_index, other_frames, others_TR_reference)
        self.integrate(frame_index, global_T_frame)
