"""CPU oracle vs the reference's own kernels (oracle/_ref/libsmx_ref.so) on the same inputs, FREE-RUNNING: the depth
stages agree bit for bit; Integrate agrees until the first pixel with two supporting candidates, then the reference's
races (first atomicCAS wins) and the oracle's fixed rule (lowest index) pick different -- equally legal -- surfels and
the two maps drift apart.  tools/ref_compare_synced.py and tests/test_gpu_reference_pin.py compare frame by frame from
a common state with the reference's race outcomes imposed on the oracle."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
import oracle as orc
from oracle import ref_binding as ref
from common import small_stream, small_pre
from oracle_pipeline import OraclePipeline

W, H = 160, 120
s = small_stream(W, H, obstacle_until=8)
pre = small_pre(W)
f = 6
d, c = s.frame(f)
# --- depth stages
a_o = orc.bilateral_filter_and_cutoff(d, pre.bilateral_filter_sigma_xy, pre.bilateral_filter_sigma_depth_factor, 0, pre.bilateral_filter_radius_factor, pre.max_depth_u16(), pre.depth_valid_region_radius)
a_r = ref.bilateral_filter_and_cutoff(d, pre.bilateral_filter_sigma_xy, pre.bilateral_filter_sigma_depth_factor, 0, pre.bilateral_filter_radius_factor, pre.max_depth_u16(), pre.depth_valid_region_radius)
diff = a_o.astype(int) - a_r.astype(int)
print('bilateral: px', a_o.size, 'valid', (a_o > 0).sum(), 'differ', (diff != 0).sum(), 'max|d|', np.abs(diff).max(), 'zero-mismatch', ((a_o == 0) != (a_r == 0)).sum())
others = [s.frame(g)[0] for g in s.outlier_frames(f)]
T = s.others_TR_reference(f)
b_o = orc.outlier_depth_map_fusion(a_o, others, T, s.fx, s.fy, s.cx, s.cy, pre.outlier_filtering_depth_tolerance_factor, -1)
b_r = ref.outlier_depth_map_fusion(a_o, others, T, s.fx, s.fy, s.cx, s.cy, pre.outlier_filtering_depth_tolerance_factor, -1)
print('outlier(all): differ', (b_o != b_r).sum(), 'kept', (b_o > 0).sum())
b_o6 = orc.outlier_depth_map_fusion(a_o, others, T, s.fx, s.fy, s.cx, s.cy, pre.outlier_filtering_depth_tolerance_factor, 6)
b_r6 = ref.outlier_depth_map_fusion(a_o, others, T, s.fx, s.fy, s.cx, s.cy, pre.outlier_filtering_depth_tolerance_factor, 6)
print('outlier(6 of 8): differ', (b_o6 != b_r6).sum(), 'kept', (b_o6 > 0).sum())
for r in (0, 1, 2, 3):
    e_o, e_r = orc.erode_depth_map(b_o, r), ref.erode_depth_map(b_o, r)
    print('erode r=%d: differ' % r, (e_o != e_r).sum(), 'kept', (e_o > 0).sum())
e_o = orc.erode_depth_map(b_o, pre.depth_erosion_radius)
n_od, n_on = orc.compute_normals_and_drop_bad_pixels(e_o, s.fx, s.fy, s.cx, s.cy, pre.observation_angle_threshold_deg, pre.depth_scaling)
n_rd, n_rn = ref.compute_normals_and_drop_bad_pixels(e_o, s.fx, s.fy, s.cx, s.cy, pre.observation_angle_threshold_deg, pre.depth_scaling)
m = (n_od > 0) & (n_rd > 0)
print('normals: depth differ', (n_od != n_rd).sum(), 'valid', m.sum(), 'max abs normal diff', np.abs(n_on[m] - n_rn[m]).max() if m.any() else None, 'bit-equal frac', (n_on[m].view(np.uint32) == n_rn[m].view(np.uint32)).mean())
r_od, r_or = orc.compute_point_radii_and_remove_isolated_pixels(n_od, s.fx, s.fy, s.cx, s.cy, pre.point_radius_extension_factor, pre.point_radius_clamp_factor, pre.depth_scaling)
r_rd, r_rr = ref.compute_point_radii_and_remove_isolated_pixels(n_od, s.fx, s.fy, s.cx, s.cy, pre.point_radius_extension_factor, pre.point_radius_clamp_factor, pre.depth_scaling)
m = (r_od > 0) & (r_rd > 0)
print('radii: depth differ', (r_od != r_rd).sum(), 'valid', m.sum(), 'max rel radius diff', (np.abs(r_or[m] - r_rr[m]) / r_or[m]).max() if m.any() else None, 'bit-equal frac', (r_or[m].view(np.uint32) == r_rr[m].view(np.uint32)).mean())

# --- integrate sequence: both sides get the ORACLE's preprocessed images
po = OraclePipeline(W, H, s.fx, s.fy, s.cx, s.cy, 60000, pre)
rr = ref.Recon(60000, W, H, s.fx, s.fy, s.cx, s.cy)
for g in range(0, 24):
    dd, cc = s.frame(g)
    po.upload(g, dd, cc)
params = orc.IntegrateParams.defaults()
for g in range(4, 16):
    po.preprocess(g, s.outlier_frames(g), s.others_TR_reference(g))
    depth_r = po.depth_final.copy()
    rr.integrate(g, pre.depth_scaling, depth_r, po.normals, po.radius, po.color[g], s.pose(g), params)
    po.integrate(g, s.pose(g))
    co, cr = po.recon.stats(), rr.counts()
    n = min(po.recon.surfels_size, cr['surfels_size'])
    So, Sr = po.recon.surfels()[:, :n], rr.surfels(n)
    so, sr = po.recon.scratch(), rr.scratch()
    sup_diff = (so['supporting'] != sr['supporting']).sum()
    cnt_diff = (so['support_counts'] != sr['support_counts']).sum()
    fd = np.abs(np.where(np.isfinite(so['first_depth']), so['first_depth'], 0) - np.where(np.isfinite(sr['first_depth']), sr['first_depth'], 0)).max()
    pos_err = np.abs(So[0:3] - Sr[0:3]).max() if n else 0
    smooth_err = np.abs(So[3:6] - Sr[3:6]).max() if n else 0
    nb_diff = (So[19:23].view(np.uint32) != Sr[19:23].view(np.uint32)).mean() if n else 0
    stamp_diff = (So[18].view(np.uint32) != Sr[18].view(np.uint32)).sum() if n else 0
    print('frame %2d: size %6d / %6d  merges %d / %d  new %d / %d | supporting px differ %d, counts differ %d, first_depth max|d| %.2e | blended depth differ %d | pos max|d| %.2e smooth %.2e nb-rows differ %.4f stamps differ %d' % (
        g, po.recon.surfels_size, cr['surfels_size'], po.recon.merge_count, cr['merge_count'], co['n_new'], cr['n_new'], sup_diff, cnt_diff, fd, (po.depth_final != depth_r).sum(), pos_err, smooth_err, nb_diff, stamp_diff))
