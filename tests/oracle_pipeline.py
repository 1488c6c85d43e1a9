"""Oracle-side twin of surfelmeshing_amd/pipeline.py (reference call order, APP/main.cc:1015-1223),
on numpy arrays through the CPU oracle.  Test infrastructure."""
import numpy as np

import oracle as orc


class OraclePipeline:
    def __init__(self, width, height, fx, fy, cx, cy, max_surfel_count, pre, params=None, sum_mode=orc.SUM_EXACT):
        self.level = getattr(pre, "pyramid_level", 0)
        if self.level > 0:   # Camera::Scaled(1 / 2^level), main.cc:751, VIS/camera.h:954-964, 1564-1574
            f = np.float32(1.0 / 2 ** self.level)
            width, height = int(float(f) * width + np.float32(0.5)), int(float(f) * height + np.float32(0.5))
            fx, fy, cx, cy = (float(np.float32(v) * f) for v in (fx, fy, cx, cy))
        self.w, self.h, self.fx, self.fy, self.cx, self.cy = width, height, fx, fy, cx, cy
        self.pre = pre
        self.params = params or orc.IntegrateParams.defaults()
        self.recon = orc.Recon(max_surfel_count, width, height, fx, fy, cx, cy, sum_mode)
        self.raw_depth, self.color = {}, {}
        self.radius = np.zeros((height, width), np.float32)
        self.stages = {}

    def upload(self, f, depth, color):
        self.raw_depth[f] = np.ascontiguousarray(depth, np.uint16)
        if getattr(self.pre, "median_filter_and_densify_iterations", 0) > 0:   # main.cc:929-939
            self.raw_depth[f] = orc.median_filter_and_densify(self.raw_depth[f], self.pre.median_filter_and_densify_iterations)
        self.color[f] = np.ascontiguousarray(color, np.uint8)
        if self.level > 0:                                                     # main.cc:941-962, 973-981
            self.raw_depth[f] = orc.downscale_using_median_while_excluding(self.raw_depth[f], self.w, self.h, 0)
            self.color[f] = orc.color_image_pyramid(self.color[f], self.level)

    def release(self, f):
        self.raw_depth.pop(f, None)
        self.color.pop(f, None)

    def preprocess(self, f, other_frames, others_TR_reference):
        p = self.pre
        a = orc.bilateral_filter_and_cutoff(self.raw_depth[f], p.bilateral_filter_sigma_xy,
                                            p.bilateral_filter_sigma_depth_factor, 0, p.bilateral_filter_radius_factor,
                                            p.max_depth_u16(), p.depth_valid_region_radius)
        self.stages["bilateral"] = a
        if other_frames:
            req = -1 if p.outlier_filtering_required_inliers in (-1, len(other_frames)) else p.outlier_filtering_required_inliers
            a = orc.outlier_depth_map_fusion(a, [self.raw_depth[g] for g in other_frames], others_TR_reference,
                                             self.fx, self.fy, self.cx, self.cy,
                                             p.outlier_filtering_depth_tolerance_factor, req)
            self.stages["outlier"] = a
        a = orc.erode_depth_map(a, p.depth_erosion_radius)
        self.stages["erode"] = a
        a, normals = orc.compute_normals_and_drop_bad_pixels(a, self.fx, self.fy, self.cx, self.cy,
                                                             p.observation_angle_threshold_deg, p.depth_scaling)
        self.stages["normals_depth"] = a
        a, self.radius = orc.compute_point_radii_and_remove_isolated_pixels(
            a, self.fx, self.fy, self.cx, self.cy, p.point_radius_extension_factor, p.point_radius_clamp_factor,
            p.depth_scaling, radius_init=self.radius)
        self.normals = normals
        self.depth_final = a
        return a

    def integrate(self, f, global_T_frame):
        self.recon.integrate(f, self.pre.depth_scaling, self.depth_final, self.normals, self.radius, self.color[f],
                             global_T_frame, self.params)

    def integrate_as(self, frame_index, data_frame, global_T_frame):
        """Integrate the preprocessed images of `data_frame` under the frame index (stamp) `frame_index`."""
        self.recon.integrate(frame_index, self.pre.depth_scaling, self.depth_final, self.normals, self.radius,
                             self.color[data_frame], global_T_frame, self.params)

    def process(self, f, other_frames, others_TR_reference, global_T_frame):
        self.preprocess(f, other_frames, others_TR_reference)
        self.integrate(f, global_T_frame)
