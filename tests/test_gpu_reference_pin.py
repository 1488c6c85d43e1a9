"""Pins the CPU oracle against the REFERENCE'S OWN KERNELS.

oracle/_ref/libsmx_ref.so (oracle/ref_build.py) is the reference's cuda_depth_processing.cu and
cuda_surfel_reconstruction_kernels.cu, compiled by hipcc for gfx950 from where they lie under /root/reference, plus a
host harness that restates the call sequence of CUDASurfelReconstruction::Integrate around the reference's launcher
functions.  These tests run those kernels on the MI355X and compare them with the oracle on the same inputs:

 * the depth-processing kernels are bit-identical (the bilateral filter up to 1 depth unit on a handful of pixels: its
   `expf` is the device library's, the oracle's is the self-contained one the product shares);
 * for Integrate, the reference decides the supporting and the conflicting surfel of a pixel by races; the oracle
   replaces each race by a fixed legal rule (oracle/smx_oracle.h).  Per frame, both sides start from the same surfel
   state, the reference's kernels run, and the oracle is then run with the race outcomes of that very run imposed --
   accepted only where the oracle itself saw the surfel qualify (so an illegal outcome would be counted, none is).
   Everything else must then agree: counts, merges, association images, blended depth, every integer row, and every
   float row bit for bit except where the reference sums floats with atomicAdd in scheduling order (depth sums ->
   a stray blended-depth LSB; regulariser gradients -> the smooth positions, compared within 5e-6 m; observed: up to 1.2e-6 m).

The library is built in the authoring container (the GPU box has no /root/reference) and travels with the repository
snapshot; without it the tests are skipped.
"""
import os

import numpy as np
import pytest

import oracle as orc
from common import FLOAT_ROWS, INT_ROWS, small_pre, small_stream
from oracle_pipeline import OraclePipeline

pytestmark = pytest.mark.gpu

SMOOTH_ROWS = (3, 4, 5)


@pytest.fixture(scope="module")
def ref():
    import torch  # noqa: F401  (the HIP runtime PyTorch ships must be loaded first)
    from oracle import ref_binding
    if not ref_binding.available():
        pytest.skip("oracle/_ref/libsmx_ref.so not built (needs /root/reference: python oracle/ref_build.py)")
    return ref_binding


@pytest.mark.parametrize("w,h", [(160, 120), (200, 77)])
def test_depth_kernels_match_the_reference(ref, w, h):
    s = small_stream(w, h, obstacle_until=8)
    pre = small_pre(w)
    for f in (5, 9):
        d, _ = s.frame(f)
        args = (pre.bilateral_filter_sigma_xy, pre.bilateral_filter_sigma_depth_factor, 0,
                pre.bilateral_filter_radius_factor, pre.max_depth_u16(), pre.depth_valid_region_radius)
        a_o, a_r = orc.bilateral_filter_and_cutoff(d, *args), ref.bilateral_filter_and_cutoff(d, *args)
        diff = a_o.astype(np.int32) - a_r.astype(np.int32)
        assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) <= max(3, a_o.size // 2000), np.count_nonzero(diff)
        assert np.array_equal(a_o == 0, a_r == 0)
        others = [s.frame(g)[0] for g in s.outlier_frames(f)]
        T = s.others_TR_reference(f)
        cam = (s.fx, s.fy, s.cx, s.cy)
        for required in (-1, 6):
            b_o = orc.outlier_depth_map_fusion(a_o, others, T, *cam, pre.outlier_filtering_depth_tolerance_factor, required)
            b_r = ref.outlier_depth_map_fusion(a_o, others, T, *cam, pre.outlier_filtering_depth_tolerance_factor, required)
            assert np.array_equal(b_o, b_r) and (b_o > 0).sum() > 1000
        for count in (2, 4, 6):
            b2_o = orc.outlier_depth_map_fusion(a_o, others[:count], T[:count], *cam, 0.02, -1)
            b2_r = ref.outlier_depth_map_fusion(a_o, others[:count], T[:count], *cam, 0.02, -1)
            assert np.array_equal(b2_o, b2_r)
        b_o = orc.outlier_depth_map_fusion(a_o, others, T, *cam, pre.outlier_filtering_depth_tolerance_factor, -1)
        for radius in (0, 1, 2, 3):
            assert np.array_equal(orc.erode_depth_map(b_o, radius), ref.erode_depth_map(b_o, radius))
        e_o = orc.erode_depth_map(b_o, pre.depth_erosion_radius)
        n_od, n_on = orc.compute_normals_and_drop_bad_pixels(e_o, *cam, pre.observation_angle_threshold_deg, pre.depth_scaling)
        n_rd, n_rn = ref.compute_normals_and_drop_bad_pixels(e_o, *cam, pre.observation_angle_threshold_deg, pre.depth_scaling)
        assert np.array_equal(n_od, n_rd) and (n_od > 0).sum() > 500
        m = n_od > 0
        assert np.array_equal(n_on[m].view(np.uint32), n_rn[m].view(np.uint32))
        r_od, r_or = orc.compute_point_radii_and_remove_isolated_pixels(
            n_od, *cam, pre.point_radius_extension_factor, pre.point_radius_clamp_factor, pre.depth_scaling)
        r_rd, r_rr = ref.compute_point_radii_and_remove_isolated_pixels(
            n_od, *cam, pre.point_radius_extension_factor, pre.point_radius_clamp_factor, pre.depth_scaling)
        assert np.array_equal(r_od, r_rd)
        m = r_od > 0
        assert m.sum() > 300 and np.array_equal(r_or[m].view(np.uint32), r_rr[m].view(np.uint32))


@pytest.mark.parametrize("kw", [
    dict(),
    dict(regularization_iterations_per_integration_iteration=3),
    dict(regularization_frame_window_size=3, surfel_integration_active_window_size=6, measurement_blending_radius=5,
         sensor_noise_factor=0.02, regularizer_weight=4.0, radius_factor_for_regularization_neighbors=1.5),
    dict(do_blending=0, regularization_iterations_per_integration_iteration=0),
])
def test_integrate_matches_the_reference_kernels_frame_by_frame(ref, kw):
    _integrate_pin(ref, kw, 160, 120, list(range(4, 24)), 60000)


def test_integrate_matches_the_reference_kernels_at_640x480(ref):
    """The same pin at the bench's resolution (the 5 M-slot version is tests/tools/ref_pin_fullsize.py, whose result
    is kept under profiles/)."""
    _integrate_pin(ref, {}, 640, 480, list(range(4, 12)), 700000)


def _integrate_pin(ref, kw, w, h, frames, cap):
    scale = (w * h) // (160 * 120)                # pixel count relative to the small case: bounds on counts scale with it
    s = small_stream(w, h, obstacle_until=10)     # vanishing obstacle: conflicts, replacements, merges
    pre = small_pre(w)
    params = orc.IntegrateParams.defaults(**kw)
    po = OraclePipeline(w, h, s.fx, s.fy, s.cx, s.cy, cap, pre, params)
    rr = ref.Recon(cap, w, h, s.fx, s.fy, s.cx, s.cy)
    for g in range(0, frames[-1] + 5):
        d, c = s.frame(g)
        po.upload(g, d, c)
    applied = merges = replaced = stray_depth = 0
    for g in frames:
        po.preprocess(g, s.outlier_frames(g), s.others_TR_reference(g))
        n0 = po.recon.surfels_size
        # same state, same preprocessed frame on both sides
        rr.upload_surfels(po.recon.surfels()[:, :n0].copy(), po.recon.merge_count)
        depth_r = po.depth_final.copy()
        rr.integrate(g, pre.depth_scaling, depth_r, po.normals, po.radius, po.color[g], s.pose(g), params)
        sr = rr.scratch()
        sup_r, conf_r = np.ascontiguousarray(sr["supporting"]), np.ascontiguousarray(sr["conflicting"])
        orc.set_race_overrides(sup_r, conf_r)
        try:
            po.integrate(g, s.pose(g))
            ovr = orc.race_override_stats()
        finally:
            orc.set_race_overrides(None, None)
        # every race outcome of the reference run is one the oracle considers legal
        assert ovr["rejected_supporting"] == 0 and ovr["rejected_conflicting"] == 0, (g, ovr)
        applied += ovr["applied_supporting"] + ovr["applied_conflicting"]
        cr, st = rr.counts(), po.recon.stats()
        n = po.recon.surfels_size
        assert (n, po.recon.merge_count, st["n_new"]) == (cr["surfels_size"], cr["merge_count"], cr["n_new"]), g
        merges, replaced = po.recon.merge_count, replaced + st["n_replaced"]
        so = po.recon.scratch()
        assert np.array_equal(so["supporting"], sr["supporting"]) and np.array_equal(so["conflicting"], sr["conflicting"])
        assert np.array_equal(so["support_counts"], sr["support_counts"])
        assert np.array_equal(so["first_depth"].view(np.uint32), sr["first_depth"].view(np.uint32))
        # blended depth: the reference's float atomicAdd depth sums depend on scheduling -> a stray LSB
        dd = po.depth_final.astype(np.int32) - depth_r.astype(np.int32)
        assert np.abs(dd).max() <= 1 and np.count_nonzero(dd) <= 6 * scale, (g, np.count_nonzero(dd))
        stray_depth += np.count_nonzero(dd)
        So, Sr = po.recon.surfels()[:, :n], rr.surfels(n)
        for row in INT_ROWS:
            assert np.array_equal(So[row].view(np.uint32), Sr[row].view(np.uint32)), (g, row)
        for row in FLOAT_ROWS:
            a, b = So[row], Sr[row]
            neq = a.view(np.uint32) != b.view(np.uint32)
            if row in SMOOTH_ROWS:
                # regulariser: float atomicAdd order in the reference, 2^-22 m fixed point in the oracle
                # (a few ulp at metres; surfels fed by a stray blended depth follow it)
                off = np.abs(a - b) > 5e-6
                assert off.sum() <= 8 * np.count_nonzero(dd) and np.abs(a - b).max() <= 1.01 / pre.depth_scaling, \
                    (g, row, off.sum(), np.abs(a - b).max())
            else:
                # (the few surfels that integrated one of the stray blended depths: several surfels share a pixel)
                assert neq.sum() <= 8 * np.count_nonzero(dd), (g, row, neq.sum())
                if neq.any():   # one depth unit is 1 / depth_scaling = 0.2 mm
                    assert np.abs(a - b)[neq].max() <= 1.01 / pre.depth_scaling, (g, row, np.abs(a - b)[neq].max())
    # the extra regulariser iteration (Regularize, cc:322-337) and ExportVertices on the final common state
    n = po.recon.surfels_size
    rr.upload_surfels(po.recon.surfels()[:, :n].copy(), po.recon.merge_count)
    po.recon.regularize(frames[-1], 6.0, 1.8, 12)
    rr.regularize(frames[-1], 6.0, 1.8, 12)
    So, Sr = po.recon.surfels()[:, :n], rr.surfels(n)
    for row in INT_ROWS + [r for r in FLOAT_ROWS if r not in SMOOTH_ROWS]:
        assert np.array_equal(So[row].view(np.uint32), Sr[row].view(np.uint32)), row
    for row in SMOOTH_ROWS:
        assert np.abs(So[row] - Sr[row]).max() <= 5e-6
    rr.upload_surfels(So.copy(), po.recon.merge_count)
    pos_o, col_o = po.recon.export_vertices()
    pos_r, col_r = rr.export_vertices()
    assert np.array_equal(pos_o.view(np.uint32), pos_r.view(np.uint32)) and np.array_equal(col_o, col_r)   # (NaN = merged)
    rr.close()
    assert po.recon.surfels_size > 12000 * scale * len(frames) // 20 and applied > 1000
    if not kw and len(frames) >= 20:
        assert merges > 50 and replaced > 50 and stray_depth <= 25 * scale, (merges, replaced, stray_depth)


def test_first_frame_without_any_race_is_identical(ref):
    """Integration into an empty map has no race at all: no override, everything but the creation-time smooth
    positions (neighbour average, float sum order) is bit-identical."""
    w, h = 160, 120
    s = small_stream(w, h)
    pre = small_pre(w)
    po = OraclePipeline(w, h, s.fx, s.fy, s.cx, s.cy, 30000, pre)
    rr = ref.Recon(30000, w, h, s.fx, s.fy, s.cx, s.cy)
    for g in range(0, 10):
        d, c = s.frame(g)
        po.upload(g, d, c)
    po.preprocess(5, s.outlier_frames(5), s.others_TR_reference(5))
    depth_r = po.depth_final.copy()
    rr.integrate(5, pre.depth_scaling, depth_r, po.normals, po.radius, po.color[5], s.pose(5), po.params)
    po.integrate(5, s.pose(5))
    n = po.recon.surfels_size
    assert n == rr.counts()["surfels_size"] > 5000
    So, Sr = po.recon.surfels()[:, :n], rr.surfels(n)
    for row in INT_ROWS + [r for r in FLOAT_ROWS if r not in SMOOTH_ROWS]:
        assert np.array_equal(So[row].view(np.uint32), Sr[row].view(np.uint32)), row
    for row in SMOOTH_ROWS:
        assert np.abs(So[row] - Sr[row]).max() <= 1e-6
    rr.close()


def test_product_depth_kernels_against_the_reference_at_full_resolution(ref, smx):
    """The HIP product path directly against the reference's kernels (not via the oracle): 640x480, the five
    preprocessing kernels chained as APP/main.cc:1015-1191 chains them.  Identical up to the bilateral filter's expf."""
    from surfelmeshing_amd.pipeline import FramePipeline
    w, h = 640, 480
    s = small_stream(w, h, obstacle_until=8)
    pre = small_pre(w)
    pg = FramePipeline(w, h, s.fx, s.fy, s.cx, s.cy, 1000, pre)
    f = 6
    frames = {g: s.frame(g) for g in range(f - 4, f + 5)}
    for g, (d, c) in frames.items():
        pg.upload(g, d, c)
    others, T = s.outlier_frames(f), s.others_TR_reference(f)
    pg.preprocess(f, others, T)
    smx.StreamSynchronize(None)
    depth_p, normals_p, radius_p = pg.depth_final.Download(), pg.normals.Download(), pg.radius.Download()
    cam = (s.fx, s.fy, s.cx, s.cy)
    a = ref.bilateral_filter_and_cutoff(frames[f][0], pre.bilateral_filter_sigma_xy, pre.bilateral_filter_sigma_depth_factor,
                                        0, pre.bilateral_filter_radius_factor, pre.max_depth_u16(), pre.depth_valid_region_radius)
    b = ref.outlier_depth_map_fusion(a, [frames[g][0] for g in others], T, *cam, pre.outlier_filtering_depth_tolerance_factor, -1)
    e = ref.erode_depth_map(b, pre.depth_erosion_radius)
    nd, nn = ref.compute_normals_and_drop_bad_pixels(e, *cam, pre.observation_angle_threshold_deg, pre.depth_scaling)
    rd, rad = ref.compute_point_radii_and_remove_isolated_pixels(nd, *cam, pre.point_radius_extension_factor,
                                                                pre.point_radius_clamp_factor, pre.depth_scaling)
    valid = (rd > 0) & (depth_p > 0)
    # a bilateral LSB can move a pixel across a later threshold: allow a handful of such pixels
    assert np.count_nonzero((rd > 0) != (depth_p > 0)) <= 20 and valid.sum() > 100000
    dd = np.abs(rd.astype(np.int32) - depth_p.astype(np.int32))[valid]
    assert dd.max() <= 1 and np.count_nonzero(dd) <= valid.sum() // 1000
    same = valid & (rd == depth_p)
    # normals and radii read the neighbouring depths: only pixels next to one of those LSBs may differ
    n_lsb = np.count_nonzero(dd) + np.count_nonzero((rd > 0) != (depth_p > 0))
    n_off = np.count_nonzero(np.any(nn[same].view(np.uint32) != normals_p[same].view(np.uint32), axis=-1))
    r_off = np.count_nonzero(rad[same].view(np.uint32) != radius_p[same].view(np.uint32))
    assert n_off <= 8 * n_lsb + 8 and r_off <= 12 * n_lsb + 8, (n_lsb, n_off, r_off)
    assert np.abs(nn[same] - normals_p[same]).max() <= 0.05
