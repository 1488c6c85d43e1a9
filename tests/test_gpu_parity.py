"""GPU parity tests: the HIP path (through the C-ABI, via surfelmeshing_amd.api) against the CPU oracle on the
same seeded inputs.  Bar (BASELINE.json): surfel indices / counts bit-exact, per-surfel floats within 1e-4
relative -- the two sides share one arithmetic contract, so the float rows are in fact required bit-equal."""
import numpy as np
import pytest

import oracle as orc
from common import assert_surfels_match, run_both, small_pre, small_stream
from oracle_pipeline import OraclePipeline

pytestmark = pytest.mark.gpu


def _pipes(smx, s, max_surfels, pre=None, params_kw=None, scan_mode=0):
    from surfelmeshing_amd.pipeline import FramePipeline
    from surfelmeshing_amd._lib import IntegrateParams
    pre = pre or small_pre(s.width)
    kw = params_kw or {}
    po = OraclePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, max_surfels, pre, orc.IntegrateParams.defaults(**kw))
    pg = FramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, max_surfels, pre, IntegrateParams.defaults(**kw))
    pg.reconstruction.set_scan_mode(scan_mode)
    return po, pg


def _compare_state(po, pg, check_scratch=True, check_stats=True):
    n = po.recon.surfels_size
    assert pg.reconstruction.surfels_size() == n
    assert pg.reconstruction.surfel_count() == po.recon.surfel_count
    assert_surfels_match(pg.reconstruction.debug_download_surfels(n), po.recon.surfels(), n)
    if check_scratch:
        for name, ref in po.recon.scratch().items():
            got = pg.reconstruction.debug_download_scratch(name)
            assert np.array_equal(got, ref), name
        assert np.array_equal(pg.depth_final.Download(), po.depth_final), "blended depth"
    if not check_stats:
        return
    so, sg = po.recon.stats(), pg.reconstruction.stats()
    for k, v in so.items():
        assert sg[k] == v, (k, v, sg[k])


# ---- preprocessing stages -------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(160, 120), (200, 77), (640, 480)])
def test_depth_stages_bit_exact(smx, w, h):
    s = small_stream(w, h)
    pre = small_pre(w)
    raw = {f: s.frame(f)[0] for f in range(0, 9)}
    f = 4
    stream = None
    bufs = {g: smx.CUDABuffer(h, w, np.uint16) for g in raw}
    for g, b in bufs.items():
        b.UploadAsync(stream, raw[g])
    A, B = smx.CUDABuffer(h, w, np.uint16), smx.CUDABuffer(h, w, np.uint16)
    N, R = smx.CUDABuffer(h, w, np.float32, 2), smx.CUDABuffer(h, w, np.float32)
    R.Clear(0.0)
    smx.BilateralFilteringAndDepthCutoffCUDA(stream, 3.0, 0.05, 0, 2.0, pre.max_depth_u16(), pre.depth_valid_region_radius, bufs[f], A)
    o = orc.bilateral_filter_and_cutoff(raw[f], max_depth=pre.max_depth_u16(), depth_valid_region_radius=pre.depth_valid_region_radius)
    assert np.array_equal(A.Download(), o)
    for count in (2, 4, 6, 8):
        for req in (-1, count - 1):
            others = s.outlier_frames(f, count)
            T = s.others_TR_reference(f, count)
            smx.OutlierDepthMapFusionCUDA(stream, 0.02, A, s.fx, s.fy, s.cx, s.cy, [bufs[g] for g in others], T, B, required_count=req)
            oo = orc.outlier_depth_map_fusion(o, [raw[g] for g in others], T, s.fx, s.fy, s.cx, s.cy, 0.02, req)
            assert np.array_equal(B.Download(), oo), (count, req)
            # the filter and the cull as ONE call (one launch for eight other frames, two through the scratch image otherwise)
            S2, B2 = smx.CUDABuffer(h, w, np.uint16), smx.CUDABuffer(h, w, np.uint16)
            smx.BilateralFilteringAndOutlierFusionCUDA(stream, 3.0, 0.05, 2.0, pre.max_depth_u16(), pre.depth_valid_region_radius, bufs[f],
                                                       0.02, s.fx, s.fy, s.cx, s.cy, [bufs[g] for g in others], T, S2, B2, required_count=req)
            assert np.array_equal(B2.Download(), oo), ("fused", count, req)
    for radius in (0, 1, 2, 3):
        if radius == 0:
            smx.CopyWithoutBorderCUDA(stream, B, A)
        else:
            smx.ErodeDepthMapCUDA(stream, radius, B, A)
        oe = orc.erode_depth_map(oo, radius)
        assert np.array_equal(A.Download(), oe), radius
    smx.ErodeDepthMapCUDA(stream, 2, B, A)
    oe = orc.erode_depth_map(oo, 2)
    smx.ComputeNormalsAndDropBadPixelsCUDA(stream, 85.0, 5000.0, s.fx, s.fy, s.cx, s.cy, A, B, N)
    on_d, on = orc.compute_normals_and_drop_bad_pixels(oe, s.fx, s.fy, s.cx, s.cy)
    assert np.array_equal(B.Download(), on_d)
    assert np.array_equal(N.Download().view(np.uint32), on.view(np.uint32))
    smx.ComputePointRadiiAndRemoveIsolatedPixelsCUDA(stream, 1.5, float("inf"), 5000.0, s.fx, s.fy, s.cx, s.cy, B, R, A)
    or_d, orad = orc.compute_point_radii_and_remove_isolated_pixels(on_d, s.fx, s.fy, s.cx, s.cy)
    assert np.array_equal(A.Download(), or_d)
    m = on_d > 0                                      # radius is written wherever the input depth is valid
    assert np.array_equal(R.Download()[m].view(np.uint32), orad[m].view(np.uint32))
    assert (or_d > 0).sum() > 0.2 * w * h


def test_depth_stage_edge_cases(smx):
    h, w = 37, 131                                    # ragged: not a multiple of the 64x16 tile
    z = np.zeros((h, w), np.uint16)
    A, B = smx.CUDABuffer(h, w, np.uint16), smx.CUDABuffer(h, w, np.uint16)
    N, R = smx.CUDABuffer(h, w, np.float32, 2), smx.CUDABuffer(h, w, np.float32)
    R.Clear(0.0)
    for img in (z, np.full((h, w), 65535, np.uint16), np.full((h, w), 1, np.uint16)):
        A.Upload(img)
        smx.BilateralFilteringAndDepthCutoffCUDA(None, 3.0, 0.05, 0, 2.0, 65535, 1000.0, A, B)
        assert np.array_equal(B.Download(), orc.bilateral_filter_and_cutoff(img, max_depth=65535, depth_valid_region_radius=1000.0))
        smx.ComputeNormalsAndDropBadPixelsCUDA(None, 85.0, 5000.0, 100.0, 100.0, 65.5, 18.5, A, B, N)
        od, on = orc.compute_normals_and_drop_bad_pixels(img, 100.0, 100.0, 65.5, 18.5)
        assert np.array_equal(B.Download(), od)
        assert np.array_equal(N.Download().view(np.uint32), on.view(np.uint32))
        smx.ComputePointRadiiAndRemoveIsolatedPixelsCUDA(None, 1.5, float("inf"), 5000.0, 100.0, 100.0, 65.5, 18.5, A, R, B)
        od, _ = orc.compute_point_radii_and_remove_isolated_pixels(img, 100.0, 100.0, 65.5, 18.5)
        assert np.array_equal(B.Download(), od)
    rng = np.random.default_rng(3)
    img = np.where(rng.random((h, w)) < 0.3, 0, rng.integers(400, 30000, (h, w))).astype(np.uint16)
    A.Upload(img)
    smx.BilateralFilteringAndDepthCutoffCUDA(None, 2.0, 0.1, 0, 3.0, 20000, 50.0, A, B)
    assert np.array_equal(B.Download(), orc.bilateral_filter_and_cutoff(img, 2.0, 0.1, 0, 3.0, 20000, 50.0))
    with pytest.raises(smx.SmxError):
        smx.ErodeDepthMapCUDA(None, 4, A, B)          # "radius value of 4 is not supported." (cu:572-574)
    with pytest.raises(smx.SmxError):
        smx.OutlierDepthMapFusionCUDA(None, 0.02, A, 1, 1, 1, 1, [A] * 3, np.zeros((3, 12), np.float32), B)


# ---- CUDABuffer -----------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(160, 120), (203, 77), (640, 480)])
def test_fused_erode_normals_radii_bit_exact(smx, w, h):
    """smx_erode_normals_radii (one launch, intermediate images in LDS tiles) leaves exactly the final depth, normals and
    radii of the oracle's three stages, for every erosion radius (0 = border copy) and sizes that are not tile multiples;
    the radius buffer keeps its old value where the depth is dropped (cu:777-780)."""
    s = small_stream(w, h)
    pre = small_pre(w)
    raw = s.frame(4)[0]
    o = orc.bilateral_filter_and_cutoff(raw, max_depth=pre.max_depth_u16(), depth_valid_region_radius=pre.depth_valid_region_radius)
    rng = np.random.default_rng(3)
    o[rng.random(o.shape) < 0.02] = 0                      # holes: exercises every validity test
    A, B = smx.CUDABuffer(h, w, np.uint16), smx.CUDABuffer(h, w, np.uint16)
    N, R = smx.CUDABuffer(h, w, np.float32, 2), smx.CUDABuffer(h, w, np.float32)
    A.UploadAsync(None, o)
    for radius in (0, 1, 2, 3):
        R.Clear(-7.0)
        smx.ErodeNormalsRadiiCUDA(None, radius, 85.0, 1.5, float("inf"), 5000.0, s.fx, s.fy, s.cx, s.cy, A, B, N, R)
        oe = orc.erode_depth_map(o, radius)
        on_d, on = orc.compute_normals_and_drop_bad_pixels(oe, s.fx, s.fy, s.cx, s.cy)
        or_d, orad = orc.compute_point_radii_and_remove_isolated_pixels(on_d, s.fx, s.fy, s.cx, s.cy,
                                                                       radius_init=np.full((h, w), -7.0, np.float32))
        assert np.array_equal(B.Download(), or_d), radius
        assert np.array_equal(N.Download().view(np.uint32), on.view(np.uint32)), radius
        assert np.array_equal(R.Download().view(np.uint32), orad.view(np.uint32)), radius
        assert (or_d > 0).sum() > 100
    with pytest.raises(smx.SmxError):
        smx.ErodeNormalsRadiiCUDA(None, 4, 85.0, 1.5, float("inf"), 5000.0, s.fx, s.fy, s.cx, s.cy, A, B, N, R)


@pytest.mark.parametrize("w,h", [(160, 120), (203, 77), (640, 480)])
def test_median_filter_and_densify_bit_exact(smx, w, h):
    """The reference's CPU MedianFilterAndDensifyDepthMap (APP/main.cc:206-252) on the GPU: two iterations, sparse
    input (45 % holes), odd sizes, borders."""
    rng = np.random.default_rng(w * 7 + h)
    d = rng.uniform(400, 9000, (h, w)).astype(np.uint16)
    d[rng.uniform(size=(h, w)) < 0.45] = 0
    d[:3, :] = 0
    a, b = smx.CUDABuffer(h, w, np.uint16), smx.CUDABuffer(h, w, np.uint16)
    a.UploadAsync(None, d)
    smx.MedianFilterAndDensifyDepthMapCUDA(None, a, b)
    smx.MedianFilterAndDensifyDepthMapCUDA(None, b, a)
    smx.StreamSynchronize(None)
    assert np.array_equal(b.Download(), orc.median_filter_and_densify(d, 1))
    assert np.array_equal(a.Download(), orc.median_filter_and_densify(d, 2))
    with pytest.raises(smx.SmxError):
        smx.MedianFilterAndDensifyDepthMapCUDA(None, a, a)                  # in place is not supported


@pytest.mark.parametrize("size,out", [((480, 640), (240, 320)), ((480, 640), (120, 160)), ((77, 203), (38, 100)),
                                      ((96, 128), (12, 16))])
def test_downscale_using_median_while_excluding_bit_exact(smx, size, out):
    """Image<u16>::DownscaleUsingMedianWhileExcluding (VIS/image.h:1003-1053; --pyramid_level's depth image): 2x2,
    4x4, uneven and 8x8 blocks, holes excluded."""
    rng = np.random.default_rng(size[0] + out[1])
    d = rng.uniform(400, 9000, size).astype(np.uint16)
    d[rng.uniform(size=size) < 0.4] = 0
    d[:8, :16] = 0                                                         # blocks without any value
    a, b = smx.CUDABuffer(size[0], size[1], np.uint16), smx.CUDABuffer(out[0], out[1], np.uint16)
    a.UploadAsync(None, d)
    smx.DownscaleUsingMedianWhileExcludingCUDA(None, 0, a, b)
    smx.StreamSynchronize(None)
    ref = orc.downscale_using_median_while_excluding(d, out[1], out[0], 0)
    assert np.array_equal(b.Download(), ref) and (ref == 0).sum() > 0 and (ref > 0).sum() > ref.size // 2
    small = smx.CUDABuffer(4, 4, np.uint16)
    if size == (480, 640):
        with pytest.raises(smx.SmxError):
            smx.DownscaleUsingMedianWhileExcludingCUDA(None, 0, a, small)  # 160 x 120 pixel blocks: unsupported


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_color_image_pyramid_bit_exact(smx, level):
    """ImagePyramid(color, level) (VIS/image_cache.h:203-275 over Image<Vec3u8>::DownscaleToHalfSize,
    VIS/image.h:929-948; --pyramid_level's colour image, APP/main.cc:973-981)."""
    rng = np.random.default_rng(level)
    for h, w in ((480, 640), (48, 80)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        a, b = smx.CUDABuffer(h, w, np.uint8, 3), smx.CUDABuffer(h >> level, w >> level, np.uint8, 3)
        a.UploadAsync(None, img)
        smx.ColorImagePyramidCUDA(None, level, a, b)
        smx.StreamSynchronize(None)
        assert np.array_equal(b.Download(), orc.color_image_pyramid(img, level))
    with pytest.raises(smx.SmxError):
        smx.ColorImagePyramidCUDA(None, level, a, a)                         # wrong output size
    odd = smx.CUDABuffer(50, 80, np.uint8, 3)
    with pytest.raises(smx.SmxError):
        smx.ColorImagePyramidCUDA(None, 2, odd, smx.CUDABuffer(12, 20, np.uint8, 3))   # 50 is not divisible by 4
    with pytest.raises(smx.SmxError):
        smx.ColorImagePyramidCUDA(None, 5, a, b)


def test_cuda_buffer_roundtrips(smx):
    rng = np.random.default_rng(0)
    for dtype, ch, shape in ((np.uint16, 1, (31, 77)), (np.float32, 2, (9, 130)), (np.uint8, 3, (17, 65)), (np.float32, 1, (25, 1000))):
        b = smx.CUDABuffer(shape[0], shape[1], dtype, ch)
        assert b.ToCUDA().pitch % 256 == 0 and b.ToCUDA().pitch >= shape[1] * np.dtype(dtype).itemsize * ch
        hs = shape + ((ch,) if ch > 1 else ())
        a = (rng.random(hs) * 200).astype(dtype)
        b.Upload(a)
        assert np.array_equal(b.Download(), a)
        b2 = smx.CUDABuffer(shape[0], shape[1], dtype, ch)
        b2.SetTo(b)
        assert np.array_equal(b2.Download(), a)
        val = np.arange(1, ch + 1).astype(dtype)
        b.Clear(val if ch > 1 else val[0])
        assert np.all(b.Download().reshape(-1, ch) == val)
    # the copy as a kernel that reads page-locked memory over the bus: 16-byte and byte-wise rows; pageable memory is refused
    for dtype, ch, shape in ((np.uint16, 1, (48, 64)), (np.uint8, 3, (30, 64)), (np.uint8, 3, (7, 13)), (np.float32, 1, (3, 5))):
        b = smx.CUDABuffer(shape[0], shape[1], dtype, ch)
        hs = shape + ((ch,) if ch > 1 else ())
        src = smx.PagelockedArray(hs, dtype)
        src.array[...] = (rng.random(hs) * 200).astype(dtype)
        b.Clear(np.zeros(ch, dtype) if ch > 1 else dtype(0))
        b.UploadByKernelAsync(None, src.array)
        smx.StreamSynchronize(None)
        assert np.array_equal(b.Download(), src.array)
        with pytest.raises(smx.SmxError):
            b.UploadByKernelAsync(None, np.ascontiguousarray(src.array.copy()))
    # ... and so is a page-locked source that is too small for the rows the kernel would read (advisor r5: it used to fault the GPU)
    import ctypes as C
    from surfelmeshing_amd import _lib
    L = _lib.load()
    big, small = smx.PagelockedArray((256, 1024), np.uint16), smx.PagelockedArray((16, 1024), np.uint16)
    yes = C.c_int32(-1)
    for arr, nbytes, want in ((big.array, big.array.nbytes, 1), (small.array, small.array.nbytes, 1), (small.array, big.array.nbytes, 0),
                              (np.zeros(64, np.uint16), 128, 0)):
        _lib.check(L.smx_host_is_page_locked(C.c_void_p(arr.ctypes.data), C.c_size_t(nbytes), C.byref(yes)))
        assert yes.value == want, (nbytes, want)
    b = smx.CUDABuffer(256, 1024, np.uint16)
    rc = L.smx_buffer_upload_by_kernel(b._h, C.c_void_p(0), C.c_void_p(small.array.ctypes.data), C.c_size_t(0), C.c_void_p(0))
    assert rc != 0 and b"needs" in L.smx_last_error()
    b.UploadByKernelAsync(None, big.array)
    smx.StreamSynchronize(None)
    # byte-range part transfers on a 1-row buffer (UploadPartAsync / DownloadPartAsync)
    b = smx.CUDABuffer(1, 1000, np.uint32)
    b.Clear(0)
    part = np.arange(10, dtype=np.uint32)
    b.UploadPartAsync(40, 40, None, part)
    out = np.zeros(10, np.uint32)
    b.DownloadPartAsync(40, 40, None, out)
    smx.StreamSynchronize(None)
    assert np.array_equal(out, part) and b.Download()[0, 10:20].tolist() == part.tolist() and b.Download()[0, 9] == 0


# ---- full pipeline --------------------------------------------------------------------------
@pytest.mark.parametrize("scan_mode", [0, 1, 2, 3, 4, 8, 16, 24, 32, 64, 96, 128, 256, 256 + 64, 256 + 96, 512, 512 + 96, 512 + 256 + 96])   # (512: pass B and the edge kernel fused into one launch -- the work list in LDS -- alone and with the spill paths; 256: the list kernels on a grid of four workgroups, every one of them walks many steps -- with 64 / 96 the regulariser step then meets segments whose far terms ALL spilled to the atomic accumulators, walk step after walk step; 4: pass B without its hot-group filter; 8: association bins of 16 pairs, the rest through the overflow list; 16: bin space reserved pair by pair; 32: far-term bins of 4 records, 64: two destinations per sender workgroup -- the rest of the regulariser's far terms through the atomic accumulators; 128: the blend's other tile size, 40 x 40 pixels here)
def test_stream_parity_every_frame(smx, scan_mode):
    s = small_stream(obstacle_until=10)               # vanishing obstacle -> conflicts and replacements
    po, pg = _pipes(smx, s, 60000, scan_mode=scan_mode)
    seen = {"replaced": 0, "merged": 0, "conflict": 0}

    def check(f):
        _compare_state(po, pg)
        st = po.recon.stats()
        seen["replaced"] += st["n_replaced"]
        seen["merged"] += st["n_merged"]
        seen["conflict"] += st["n_conflict_hits"]

    run_both(po, pg, s, list(range(4, 26)), check)
    assert seen["replaced"] > 50 and seen["merged"] > 10 and seen["conflict"] > 100, seen
    assert po.recon.surfels_size > 10000


@pytest.mark.parametrize("overlap", [True, False])
def test_stream_parity_with_skipped_segments(smx, overlap):
    """A camera that turns away from what it mapped, short windows, the statistics counters off: the regulariser's link
    scan then skips the segments whose links all stay among slots nothing has happened to for two calls (with the
    counters on it reads every link to count it, which is why no other stream test exercises this).  Every frame is
    compared; the number of skipped segments must be substantial, and the map the camera comes BACK to at the end -- cold
    segments turning hot again, new links into them -- must match as well."""
    s = small_stream(yaw_deg_per_frame=3.0, obstacle_until=8)
    kw = dict(surfel_integration_active_window_size=4, regularization_frame_window_size=3)
    po, pg = _pipes(smx, s, 200000, params_kw=kw)
    pg.reconstruction.set_stats_enabled(False)
    pg.reconstruction.set_overlap(overlap)
    skipped = []

    def check(f):
        _compare_state(po, pg, check_stats=False)
        skipped.append(pg.reconstruction.debug_count_skipped_segments())

    run_both(po, pg, s, list(range(4, 60)), check)
    n_segments = (po.recon.surfels_size + 1023) // 1024
    assert n_segments >= 20 and max(skipped) >= n_segments // 3, (n_segments, skipped)
    # ... and back again: the yaw runs backwards under increasing frame numbers
    s2 = small_stream(yaw_deg_per_frame=-3.0, start_yaw_deg=3.0 * 120, obstacle_until=-1)
    run_both(po, pg, s2, list(range(60, 90)), check)
    assert min(skipped[-10:]) >= 1, (n_segments, skipped)


def test_culled_segments_and_standalone_regularize(smx):
    """Pass A's cull step and the flag bytes of culled segments (round 4): a camera that turns away leaves segments culled
    for many calls in a row -- from the third call on their flag bytes are no longer copied between the two copies of the
    flag table.  A standalone Regularize with ANOTHER window in the middle rewrites the current copy for every slot (recent
    bits appear in culled segments), so the copies have to be made again afterwards; the frames that follow -- more culled
    calls, then the camera coming back -- must match the oracle frame by frame, regulariser included."""
    s = small_stream(yaw_deg_per_frame=3.0, obstacle_until=8)
    kw = dict(surfel_integration_active_window_size=4, regularization_frame_window_size=3)
    po, pg = _pipes(smx, s, 200000, params_kw=kw)
    pg.reconstruction.set_stats_enabled(False)

    def check(f):
        _compare_state(po, pg, check_stats=False)

    run_both(po, pg, s, list(range(4, 40)), check)
    skipped_before = pg.reconstruction.debug_count_skipped_segments()
    assert skipped_before >= 5, skipped_before
    for window in (40, 3):   # a window that makes old slots recent again, then the stream's own
        po.recon.regularize(39, 10.0, 2.0, window)
        pg.reconstruction.Regularize(None, 39, 10.0, 2.0, window)
        _compare_state(po, pg, check_scratch=False, check_stats=False)
    run_both(po, pg, s, list(range(40, 60)), check)
    s2 = small_stream(yaw_deg_per_frame=-3.0, start_yaw_deg=3.0 * 120, obstacle_until=-1)
    run_both(po, pg, s2, list(range(60, 85)), check)


@pytest.mark.parametrize("w,h,scan_mode", [(170, 101, 0), (97, 64, 0), (170, 101, 128), (97, 64, 128)])
def test_stream_parity_image_sizes_that_cut_tiles(smx, w, h, scan_mode):
    """Image sizes that are no multiple of the association tiles (32 x 8), the blend tiles (32 x 32, or 40 x 40 with
    scan mode 128) or the scan blocks: partial tiles at the right and bottom edges in every per-pixel kernel."""
    s = small_stream(w, h, obstacle_until=8)
    po, pg = _pipes(smx, s, 60000, scan_mode=scan_mode)
    run_both(po, pg, s, list(range(4, 18)), lambda f: _compare_state(po, pg))
    assert po.recon.surfels_size > 3000


@pytest.mark.parametrize("kw", [
    dict(do_blending=0),
    dict(regularization_iterations_per_integration_iteration=0),
    dict(regularization_iterations_per_integration_iteration=3),
    dict(surfel_integration_active_window_size=4, regularization_frame_window_size=3),
    dict(measurement_blending_radius=4, sensor_noise_factor=0.02, max_surfel_confidence=2.5,
         normal_compatibility_threshold_deg=20.0, radius_factor_for_regularization_neighbors=1.2, regularizer_weight=3.0),
])
def test_stream_parity_parameter_variants(smx, kw):
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000, params_kw=kw)
    run_both(po, pg, s, list(range(4, 20)), lambda f: _compare_state(po, pg))


@pytest.mark.parametrize("overlap,handover", [(True, 1), (True, 0), (False, 1)])
def test_pipelined_frames_parity(smx, overlap, handover):
    """No download between frames: with pipelining on, the regulariser of frame f really runs beside the first
    kernels of frame f+1 (a per-frame state download would order them).  Same final state either way.  handover: how the
    front of a call reaches the internal stream (smx_recon_set_handover_mode: 1 = device word + gate kernel, 0 = event)."""
    s = small_stream(obstacle_until=10)
    po, pg = _pipes(smx, s, 60000)
    pg.reconstruction.set_overlap(overlap)
    pg.reconstruction.set_handover_mode(handover)
    run_both(po, pg, s, list(range(4, 30)), None)
    _compare_state(po, pg)
    # an extra Regularize() and a second run of frames on top of the pending regulariser
    po.recon.regularize(29, 10.0, 2.0, 30)
    pg.reconstruction.Regularize(None, 29, 10.0, 2.0, 30)
    run_both(po, pg, s, list(range(30, 36)), None)
    _compare_state(po, pg)


class _EditedStream:
    """A synthetic stream whose frames are edited before they reach both pipelines."""

    def __init__(self, base, edit):
        self._b, self._edit = base, edit
        for k in ("width", "height", "fx", "fy", "cx", "cy"):
            setattr(self, k, getattr(base, k))

    def frame(self, f):
        d, c = self._b.frame(f)
        return self._edit(f, d.copy(), c)

    def __getattr__(self, name):
        return getattr(self._b, name)


def test_empty_and_degenerate_frames(smx):
    """Empty inputs in the middle of a run: frames without a single valid depth pixel (also as outlier-cull
    partners), a frame where only one isolated 8x8 patch is valid, then normal frames again.  Empty map first."""
    base = small_stream(obstacle_until=8)

    def edit(f, d, c):
        if f in (2, 3, 9, 10, 11):
            d[:] = 0                                   # nothing measured
        if f == 14:
            keep = d[40:48, 60:68].copy()
            d[:] = 0
            d[40:48, 60:68] = keep                      # one patch: everything else must be carved / unsupported
        return d, c

    s = _EditedStream(base, edit)
    po, pg = _pipes(smx, s, 60000)
    seen_empty = []

    def check(f):
        _compare_state(po, pg)
        if f in (9, 10, 11):
            seen_empty.append(po.recon.stats()["n_new"])

    # frame 4 integrates into an EMPTY map whose outlier-cull partners include the empty frames 2 and 3
    run_both(po, pg, s, list(range(4, 20)), check)
    assert seen_empty == [0, 0, 0]
    assert po.recon.surfels_size > 5000
    # Regularize / TransferAllToCPU / ExportVertices on an object that never integrated anything
    cam = smx.PinholeCamera4f(64, 48, 50.0, 50.0, 32.0, 24.0)
    rec = smx.CUDASurfelReconstruction(1000, cam)
    rec.Regularize(None, 0, 10.0, 2.0, 30)
    cpu = smx.CUDASurfelsCPU(1000)
    cpu.LockWriteBuffers()
    rec.TransferAllToCPU(None, 0, cpu)
    smx.StreamSynchronize(None)
    cpu.UnlockWriteBuffers()
    cpu.WaitForLockAndSwapBuffers()
    assert cpu.read_buffers().surfel_count == 0 and rec.surfels_size() == 0 and rec.surfel_count() == 0


def test_large_blending_radius_uses_multi_launch_path(smx):
    """measurement_blending_radius beyond the fused kernel's LDS halo (> 17) takes the start + iteration launches."""
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000, params_kw=dict(measurement_blending_radius=24))
    run_both(po, pg, s, list(range(4, 14)), lambda f: _compare_state(po, pg))


def test_camera_turns_away_and_back(smx):
    """Frames whose view shares nothing with the map (no visible surfel, empty work lists), then the old view again
    (surfels outside the regulariser window re-enter it)."""
    from surfelmeshing_amd._lib import IntegrateParams
    s = small_stream(obstacle_until=6, yaw_deg_per_frame=2.0)
    po, pg = _pipes(smx, s, 90000, params_kw=dict(regularization_frame_window_size=3))
    frames = list(range(4, 10)) + list(range(94, 100)) + list(range(4 + 180, 10 + 180))   # yaw 2 deg/frame: 180 deg away, back
    lo, hi = 0, max(frames) + 5
    for f in range(lo, hi):
        if any(abs(f - g) <= 4 for g in frames):
            d, c = s.frame(f)
            po.upload(f, d, c)
            pg.upload(f, d, c)
    for f in frames:
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.process(f, others, T, pose)
        pg.process(f, others, T, pose)
        _compare_state(po, pg)
        st = po.recon.stats()
        if f == 94:
            assert st["n_visible"] == 0 and st["n_new"] > 1000          # nothing of the map in view
        if f in (96, 97):
            # pass A's segment culling: the segments of frames 4..9 are behind the camera and unchanged
            assert pg.reconstruction.stats()["n_segments_skipped"] >= 4
        if f == 184:
            assert st["n_visible"] > 1000 and st["n_integrated"] > 1000  # old surfels, long outside the window


def test_segment_culling_while_panning(smx):
    """Pass A's segment culling with segments leaving and re-entering the view gradually: a pan of 3 deg/frame
    one way and back, short regulariser window (a segment can only be culled once its newest stamp left the window).
    Every frame is compared; culled segments must have occurred."""
    s = small_stream(obstacle_until=6, yaw_deg_per_frame=3.0)
    po, pg = _pipes(smx, s, 120000, params_kw=dict(regularization_frame_window_size=2))
    out = list(range(4, 44))
    frames = out + [88 - f for f in range(45, 80)]          # forward, then the same poses backwards (frames 43 .. 9)
    need = sorted(set(g for f in frames for g in range(f - 4, f + 5)))
    for f in need:
        d, c = s.frame(f)
        po.upload(f, d, c)
        pg.upload(f, d, c)
    skipped = 0
    for k, f in enumerate(frames):
        # logical frame index k + 4 keeps the stamps increasing while the pose index f goes back and forth
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.preprocess(f, others, T)
        pg.preprocess(f, others, T)
        po.integrate_as(k + 4, f, pose)
        pg.integrate_as(k + 4, f, pose)
        _compare_state(po, pg)
        skipped += pg.reconstruction.stats()["n_segments_skipped"]
    assert skipped >= 10, skipped


def test_stream_with_median_densify_iterations(smx):
    """median_filter_and_densify_iterations = 2 (APP/main.cc:929-939) in front of the preprocessing, whole pipeline."""
    s = small_stream(obstacle_until=8, dropout=0.03)
    pre = small_pre(s.width, median_filter_and_densify_iterations=2)
    po, pg = _pipes(smx, s, 60000, pre=pre)
    run_both(po, pg, s, list(range(4, 14)), lambda f: _compare_state(po, pg))
    assert po.recon.surfels_size > 8000


def test_stream_with_pyramid_level(smx):
    """--pyramid_level 1 (APP/main.cc:299-303, 751, 941-981): 320 x 240 input frames, depth reduced by the median of the
    valid pixels, colour by the truncating 2 x 2 mean, camera scaled by 1/2 -- whole pipeline, every frame."""
    s = small_stream(320, 240, obstacle_until=8, dropout=0.03)
    pre = small_pre(160, pyramid_level=1)
    po, pg = _pipes(smx, s, 60000, pre=pre)
    assert (pg.w, pg.h, po.w, po.h) == (160, 120, 160, 120) and pg.fx == po.fx == s.fx / 2 and pg.cx == po.cx == s.cx / 2
    run_both(po, pg, s, list(range(4, 12)), lambda f: _compare_state(po, pg))
    assert po.recon.surfels_size > 2000
    f = 8
    assert np.array_equal(pg.color[f].Download(), po.color[f]) and np.array_equal(pg.raw_depth[f].Download(), po.raw_depth[f])
    from surfelmeshing_amd.pipeline import FramePipeline
    with pytest.raises(ValueError):
        FramePipeline(320, 240, s.fx, s.fy, s.cx, s.cy, 1000, small_pre(160, pyramid_level=1, median_filter_and_densify_iterations=1))
    with pytest.raises(ValueError):
        FramePipeline(322, 240, s.fx, s.fy, s.cx, s.cy, 1000, small_pre(160, pyramid_level=2))


def test_full_resolution_parity(smx):
    s = small_stream(640, 480)
    po, pg = _pipes(smx, s, 1200000)
    run_both(po, pg, s, list(range(4, 9)), None)
    _compare_state(po, pg)
    assert po.recon.surfels_size > 150000


def test_c3_resolution_parity_with_state_injection(smx):
    """Config C3's frame size (1280x960, fx = fy = 1050; BASELINE.json configs[2], APP/main.cc:318): six frames from an
    empty map (> 1 M slots), the GPU state re-injected into a fresh object through debug_upload_surfels, then five more
    frames on both sides -- every surfel row, the association images, the blended depth and the counters bit-equal after
    EVERY frame.  The camera pans 6 degrees per frame with the counting variant of the outlier cull (4 of 8 neighbours),
    so every frame adds ~10^5 surfels.  (The 20 M-slot state itself is compared inside `bench.py --config C3`:
    parity_check in its JSON line, kept under profiles/.)"""
    import os
    from oracle import binding
    from surfelmeshing_amd.pipeline import FramePipeline, PreprocessParams
    s = small_stream(1280, 960, yaw_deg_per_frame=6.0)
    pre = PreprocessParams(max_depth=10.0, depth_valid_region_radius=2000.0, outlier_filtering_required_inliers=4)
    cap = 4_000_000
    binding.set_row_threads(min(os.cpu_count() or 1, 32))   # (row-parallel oracle stages: identical images, tests/test_oracle_properties.py)
    try:
        po, pg = _pipes(smx, s, cap, pre)
        run_both(po, pg, s, list(range(4, 10)), lambda f: _compare_state(po, pg))
        n = po.recon.surfels_size
        assert n > 1_000_000
        rows = pg.reconstruction.debug_download_surfels(n)
        pg2 = FramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, cap, pre)
        pg2.reconstruction.debug_upload_surfels(rows, po.recon.merge_count)
        assert pg2.reconstruction.surfels_size() == n and pg2.reconstruction.surfel_count() == po.recon.surfel_count
        for f in range(6, 19):
            d, c = s.frame(f)
            pg2.upload(f, d, c)
            if f not in po.raw_depth:
                po.upload(f, d, c)
        merged_before = po.recon.merge_count
        for f in range(10, 15):
            for p in (po, pg2):
                p.process(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f))
            _compare_state(po, pg2)
        st = pg2.reconstruction.stats()
        assert st["n_integrated"] > 300_000 and po.recon.merge_count > merged_before and po.recon.surfels_size > n
    finally:
        binding.set_row_threads(1)


def test_regularize_transfer_export(smx):
    s = small_stream()
    po, pg = _pipes(smx, s, 60000)
    run_both(po, pg, s, list(range(4, 12)), None)
    # extra regulariser iteration, cc:322-337
    po.recon.regularize(11, 10.0, 2.0, 30)
    pg.reconstruction.Regularize(None, 11, 10.0, 2.0, 30)
    _compare_state(po, pg, check_scratch=False)
    # TransferAllToCPU through the CUDASurfelsCPU double buffer (protocol of test_triangulation.cc:71-99)
    cpu = smx.CUDASurfelsCPU(60000)
    with pytest.raises(smx.SmxError):
        cpu.WaitForLockAndSwapBuffers()               # nothing written yet -> LOG(FATAL) in the reference
    cpu.LockWriteBuffers()
    pg.reconstruction.TransferAllToCPU(None, 11, cpu)
    smx.StreamSynchronize(None)
    cpu.UnlockWriteBuffers()
    cpu.WaitForLockAndSwapBuffers()
    rb = cpu.read_buffers()
    t = po.recon.transfer_all()
    n = t["surfel_count"]
    assert rb.surfel_count == n and rb.frame_index == 11
    for a, b in (("surfel_x_buffer", "x"), ("surfel_y_buffer", "y"), ("surfel_z_buffer", "z"),
                 ("surfel_radius_squared_buffer", "radius_squared"), ("surfel_normal_x_buffer", "normal_x"),
                 ("surfel_normal_y_buffer", "normal_y"), ("surfel_normal_z_buffer", "normal_z"),
                 ("surfel_last_update_stamp_buffer", "last_update_stamp")):
        assert np.array_equal(getattr(rb, a)[:n].view(np.uint32), t[b].view(np.uint32)), a
    # ExportVertices
    pos, col = smx.CUDABuffer(1, 3 * n, np.float32), smx.CUDABuffer(1, 3 * n, np.uint8)
    pg.reconstruction.ExportVertices(None, pos, col)
    opos, ocol = po.recon.export_vertices()
    assert np.array_equal(pos.Download()[0].view(np.uint32), opos.view(np.uint32))
    assert np.array_equal(col.Download()[0], ocol)
    tm = pg.reconstruction.GetTimings()
    assert len(tm) == 7 and all(x >= 0 for x in tm)


def test_changed_surfel_delta_reproduces_full_transfers(smx):
    """a full transfer patched with every delta since equals the current full transfer bit for bit
    (merges, replacements, new surfels and the regulariser's moves included); steady-state deltas are small."""
    s = small_stream(obstacle_until=10, yaw_deg_per_frame=2.0)
    po, pg = _pipes(smx, s, 60000, params_kw=dict(regularization_frame_window_size=2))
    rec = pg.reconstruction
    with pytest.raises(smx.SmxError):
        rec.TransferChangedToCPU(None, 0)                       # tracking is off
    rec.SetDeltaTracking(None, True)

    def full():
        cpu = smx.CUDASurfelsCPU(60000)
        cpu.LockWriteBuffers()
        rec.TransferAllToCPU(None, 0, cpu)
        smx.StreamSynchronize(None)
        cpu.UnlockWriteBuffers()
        cpu.WaitForLockAndSwapBuffers()
        return cpu.read_buffers()

    names = [full_name for _, _, full_name in smx.CUDASurfelDeltaCPU.ROWS]
    mirror = smx.CUDASurfelBuffersCPU(60000)
    for n in names:
        getattr(mirror, n)[:] = 0
    sizes = []
    frames = list(range(4, 30))
    for f in range(0, 36):
        d, c = s.frame(f)
        po.upload(f, d, c)
        pg.upload(f, d, c)
    for k, f in enumerate(frames):
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.process(f, others, T, pose)
        pg.process(f, others, T, pose)
        if k % 3 == 2 or f == frames[-1]:                       # (several frames per delta, like the mesher's cadence)
            delta = rec.TransferChangedToCPU(None, f)
            assert np.all(np.diff(delta.surfel_index[:delta.count].astype(np.int64)) > 0)
            delta.ApplyTo(mirror)
            sizes.append((delta.count, delta.surfel_count))
            ref = full()
            n = ref.surfel_count
            assert mirror.surfel_count == n == po.recon.surfels_size
            for name in names:
                a, b = getattr(mirror, name)[:n], getattr(ref, name)[:n]
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (f, name)
    # the first delta carries everything, later ones only what moved
    assert sizes[0][0] == sizes[0][1] and all(c < 0.8 * n_ for c, n_ in sizes[3:]), sizes
    assert rec.TransferChangedToCPU(None, frames[-1]).count == 0   # nothing changed since
    # too small a capacity: the needed count is reported and nothing is lost
    pg.process(30, s.outlier_frames(30), s.others_TR_reference(30), s.pose(30))
    po.process(30, s.outlier_frames(30), s.others_TR_reference(30), s.pose(30))
    with pytest.raises(smx.SmxError):
        rec.TransferChangedToCPU(None, 30, capacity=10)
    needed = rec.last_failed_delta.count_needed
    delta = rec.TransferChangedToCPU(None, 30, capacity=needed)
    assert delta.count == needed > 10
    delta.ApplyTo(mirror)
    ref = full()
    for name in names:
        assert np.array_equal(getattr(mirror, name)[:ref.surfel_count].view(np.uint32),
                              getattr(ref, name)[:ref.surfel_count].view(np.uint32)), name
    rec.SetDeltaTracking(None, False)


def test_capacity_clamp(smx):
    s = small_stream()
    cap = 9000                                         # first frame alone wants ~9.4k surfels
    po, pg = _pipes(smx, s, cap)
    run_both(po, pg, s, list(range(4, 8)), None)
    assert po.recon.surfels_size == cap
    _compare_state(po, pg)
    assert pg.reconstruction.stats()["capacity_clamped"] in (0, 1)


def test_state_injection_roundtrip(smx):
    s = small_stream()
    po, pg = _pipes(smx, s, 60000)
    run_both(po, pg, s, list(range(4, 10)), None)
    n = po.recon.surfels_size
    rows = pg.reconstruction.debug_download_surfels(n)
    from surfelmeshing_amd.pipeline import FramePipeline
    pg2 = FramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, small_pre(s.width))
    pg2.reconstruction.debug_upload_surfels(rows, po.recon.merge_count)
    assert pg2.reconstruction.surfels_size() == n and pg2.reconstruction.surfel_count() == po.recon.surfel_count
    for f in range(6, 15):
        pg2.upload(f, *s.frame(f))
        po.upload(f, *s.frame(f))
    f = 10
    for p in (po, pg2):
        p.process(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f))
    _compare_state(po, pg2)


def test_row_upload_download_roundtrip_every_row(smx):
    """The debug row accessors convert between the reference's 25-row layout and the grouped records: every row with
    storage comes back bit for bit (distinct values per row and slot, so that a write into a neighbouring record shows),
    the rows without storage (14-16 Accum*, 23 GradientCount) read as 0 and are ignored on upload; counts that are and
    are not multiples of the segment size."""
    cam = smx.PinholeCamera4f(160, 120, 130.0, 130.0, 80.0, 60.0)
    for n, cap in ((1000, 60000), (70001, 100000), (1_300_000, 1_500_000)):
        r = smx.CUDASurfelReconstruction(cap, cam)
        rows = (np.arange(25, dtype=np.uint32)[:, None] * np.uint32(50_000_000) + np.arange(n, dtype=np.uint32)[None, :] + np.uint32(7)).view(np.float32)
        r.debug_upload_surfels(np.ascontiguousarray(rows), 3)
        assert r.surfels_size() == n and r.surfel_count() == n - 3
        back = r.debug_download_surfels(n)
        for k in range(25):
            if k in (14, 15, 16, 23):
                assert not back[k].view(np.uint32).any(), k
            else:
                assert np.array_equal(back[k].view(np.uint32), rows[k].view(np.uint32)), k
        r.close()


def test_invalid_arguments_raise(smx):
    cam = smx.PinholeCamera4f(64, 48, 50.0, 50.0, 32.0, 24.0)
    rec = smx.CUDASurfelReconstruction(1000, cam)
    d = smx.CUDABuffer(48, 64, np.uint16)
    wrong = smx.CUDABuffer(40, 64, np.uint16)
    n, r, c = smx.CUDABuffer(48, 64, np.float32, 2), smx.CUDABuffer(48, 64, np.float32), smx.CUDABuffer(48, 64, np.uint8, 3)
    from surfelmeshing_amd._lib import IntegrateParams
    with pytest.raises(smx.SmxError):
        rec.IntegrateP(None, 0, 5000.0, wrong, n, r, c, np.eye(3, 4), IntegrateParams.defaults())
    assert rec.surfels_size() == 0 and rec.surfel_count() == 0


def test_streams_with_priority_and_explicit_stream(smx):
    """Integrate on an explicit (non-default) caller stream, with pipelining: same result as the oracle; stream
    creation with a priority class validates its argument."""
    with pytest.raises(smx.SmxError):
        smx.Stream(priority_class=5)
    st = smx.Stream(priority_class=1)
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000)
    smx.StreamSynchronize(None)                        # (construction used the default stream)
    pg.stream = st
    run_both(po, pg, s, list(range(4, 16)), None)
    st.synchronize()
    _compare_state(po, pg)
    st.close()


def test_stream_on_a_subset_of_the_compute_units(smx):
    """smx_stream_create_with_cu_mask (an experiment's plumbing, kept as API): Integrate on a stream that may only use
    half of the compute units, the internal stream on the other half, gives the same map."""
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000)
    lo = [0xFFFFFFFF] * 4 + [0] * 4
    hi = [0] * 4 + [0xFFFFFFFF] * 4
    st = smx.Stream(cu_mask=lo)
    smx.StreamSynchronize(None)                        # (construction used the default stream)
    pg.stream = st
    pg.reconstruction.set_internal_cu_mask(hi)
    run_both(po, pg, s, list(range(4, 14)), None)
    st.synchronize()
    _compare_state(po, pg)
    pg.reconstruction.set_internal_cu_mask([])   # (all of them again)
    run_both(po, pg, s, list(range(14, 18)), None)
    st.synchronize()
    _compare_state(po, pg)
    st.close()
    with pytest.raises(smx.SmxError):
        smx.Stream(cu_mask=[0, 0])


def test_handover_gate_gives_up_instead_of_hanging(smx, monkeypatch):
    """The device-word hand-over's safety net: a gate that waits for more than will ever come (debug_skip bit 5) gives up after its
    bound instead of hanging the queue, and the next synchronising entry point says so; under a counter-collecting profiler
    (ROCPROF_COUNTER_COLLECTION) an object starts with the event hand-over."""
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000)
    assert pg.reconstruction.handover_mode() == 1
    run_both(po, pg, s, list(range(4, 8)), None)
    _compare_state(po, pg)
    pg.reconstruction.debug_set_skip(32)
    run_both(po, pg, s, [8], None)
    pg.reconstruction.debug_set_skip(0)
    with pytest.raises(smx.SmxError, match="hand-over timed out"):
        pg.reconstruction.surfels_size()
    run_both(po, pg, s, [9], None)                      # the next call finds the gate's mark and goes back to the event
    assert pg.reconstruction.handover_mode() == 0
    monkeypatch.setenv("ROCPROF_COUNTER_COLLECTION", "1")
    _, pg2 = _pipes(smx, s, 60000)
    assert pg2.reconstruction.handover_mode() == 0


def test_two_objects_on_two_host_threads(smx):
    """The 8-GPU code path minus the other seven GPUs: two host threads, each with its own native frame
    driver (reconstruction object, streams, work sets) and its own synthetic stream, run CONCURRENTLY on the one GPU;
    both maps equal the oracle's.  Shakes out state shared between objects (there must be none)."""
    import threading
    from surfelmeshing_amd.pipeline import NativeFramePipeline
    from surfelmeshing_amd._lib import IntegrateParams
    frames = list(range(4, 24))
    jobs = []
    for seed, until in ((1, 8), (7, 12)):
        s = small_stream(obstacle_until=until, seed=seed)
        pre = small_pre(s.width)
        po = OraclePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre)
        pn = NativeFramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre, IntegrateParams.defaults())
        for f in range(0, 28):
            d, c = s.frame(f)
            po.upload(f, d, c)
            pn.upload(f, d, c)
        steps = []
        for f in frames:
            others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
            po.process(f, others, T, pose)
            steps.append(pn.make_step(f, others, T, pose))
        jobs.append((po, pn, steps))
    smx.StreamSynchronize(None)
    errors = []

    def work(pn, steps):
        try:
            st = smx.Stream()
            pn.stream = st
            for rep in range(3):                      # several enqueue calls per thread, interleaving with the other one
                pn.run(steps[rep * 7:(rep + 1) * 7] if rep < 2 else steps[14:])
            st.synchronize()
            pn.stream = None
            st.close()
        except Exception as e:                         # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(pn, steps)) for _, pn, steps in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for po, pn, _ in jobs:
        n = po.recon.surfels_size
        assert n > 5000 and pn.reconstruction.surfels_size() == n
        assert_surfels_match(pn.reconstruction.debug_download_surfels(n), po.recon.surfels(), n)
    assert jobs[0][0].recon.surfels_size != jobs[1][0].recon.surfels_size   # (two different streams)


def test_frame_index_may_decrease(smx):
    """A sequence replayed on a live object (frame indices jump back): accepted like in the reference, same state as
    the oracle -- the segment / hot-group caches that presume forward time are dropped for that call."""
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000, params_kw=dict(regularization_frame_window_size=4))
    run_both(po, pg, s, list(range(4, 16)), None)
    _compare_state(po, pg)
    run_both(po, pg, s, [8, 9, 10, 5, 6, 30], lambda f: _compare_state(po, pg))


def test_get_timings_are_consistent(smx):
    """GetTimings (cc:409-436): seven stage times of the last Integrate call, on from the first call like the reference's
    events -- served by the kernels' own stage stamps (device wall clock).  Every stage that has a launch of its own takes
    measurable time, the sum is below the wall time of the call, the stamps agree with the reference's 14 event records
    (the measurement mode), and the per-kernel slots cover the same kernels."""
    import time
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000)
    rec = pg.reconstruction
    assert rec.GetTimings() == (0.0,) * 7            # no Integrate call yet
    run_both(po, pg, s, list(range(4, 12)), None)
    first = rec.GetTimings()                         # no arming call: the times of the last call at once
    assert first[0] > 0.001 and first[2] > 0.001 and first[3] > 0.001 and first[4] > 0.001 and first[6] > 0.001, first
    assert first[1] == 0.0 and first[5] == 0.0       # merging / creation: fused into other stages' launches
    smx.StreamSynchronize(None)
    t0 = time.perf_counter()
    run_both(po, pg, s, [12], None)
    by_stamps = rec.GetTimings()
    wall_ms = 1e3 * (time.perf_counter() - t0)
    assert all(x >= 0 for x in by_stamps) and sum(by_stamps) < wall_ms, (by_stamps, wall_ms)
    # the same stages by the reference's own event records (mode bit 0), a few frames of each kind in alternation; tiny
    # launches at this size (10 - 30 us each, an event packet is 5 us), hence the wide tolerance -- bench.py reports both
    # at full size
    sums = {4: np.zeros(7), 1: np.zeros(7)}
    f = 13
    for rep in range(6):
        for mode in (4, 1):
            rec.set_timing_enabled(mode)
            run_both(po, pg, s, [f], None)
            f += 1
            sums[mode] += np.array(rec.GetTimings())
    st_, ev_ = sums[4] / 6, sums[1] / 6
    # (a band, not a distance: the event mode pays for its fourteen records -- a packet each between launches of 10 - 30 us -- and,
    # since round 6, for the event hand-over it keeps while the stamp mode hands over by a device word; the stamps must never
    # exceed the events by more than their noise and must not shrink to nothing)
    assert 0.3 * ev_.sum() < st_.sum() < 1.2 * ev_.sum() + 0.03, (st_, ev_)
    for k in (0, 2, 3, 4, 6):
        assert st_[k] > 0.001 and ev_[k] > 0.001 and 0.2 * ev_[k] - 0.03 < st_[k] < 1.5 * ev_[k] + 0.03, (k, st_, ev_)
    rec.set_timing_enabled(3)
    smx.StreamSynchronize(None)
    t0 = time.perf_counter()
    run_both(po, pg, s, [f], None)
    t = rec.GetTimings()
    wall_ms = 1e3 * (time.perf_counter() - t0)
    assert len(t) == 7 and all(x >= 0 for x in t)
    assert t[0] > 0.001 and t[3] > 0.001 and t[4] > 0.001 and t[6] > 0.001, t    # association, integration, neighbours, regulariser
    assert sum(t) < wall_ms, (t, wall_ms)
    k = dict(zip(rec.kernel_time_names(), rec.kernel_times_ms()))
    for name in ("scan_visible", "assoc_tiles", "blend", "integrate+new_flags", "update_neighbors+create", "neighbor_scan",
                 "reg_accumulate", "reg_step"):
        assert 0.0005 < k[name] < wall_ms, (name, k)
    rec.set_timing_enabled(0)
    run_both(po, pg, s, [f + 1], None)
    assert rec.GetTimings() == (0.0,) * 7            # everything off: zeros
    rec.set_timing_enabled(4)
    n = po.recon.surfels_size
    assert_surfels_match(rec.debug_download_surfels(n), po.recon.surfels(), n)


def test_frame_loop_reads_stage_times_every_frame(smx):
    """APP/main.cc:1511-1524 reads GetTimings after every Integrate and accumulates; the native frame loop does the same
    with the non-waiting read (every completed call counted once, lagging the queue) or with the blocking one (every
    call, the reference's semantics)."""
    from surfelmeshing_amd.pipeline import NativeFramePipeline
    from surfelmeshing_amd._lib import IntegrateParams
    s = small_stream(obstacle_until=8)
    pre = small_pre(s.width)
    pn = NativeFramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre, IntegrateParams.defaults())
    frames = list(range(4, 36))
    for f in range(0, 48):
        d, c = s.frame(f)
        pn.upload(f, d, c)
    steps = [pn.make_step(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)) for f in frames]
    pn.set_read_timings(1)
    pn.run(steps)
    smx.StreamSynchronize(None)
    sums, calls = pn.timing_sums()
    assert 1 <= calls <= len(frames), calls
    assert sums[0] > 0 and sums[3] > 0 and sums[6] > 0 and sums[1] == 0.0, sums
    t, call = pn.reconstruction.GetTimingsNoWait()
    assert call == len(frames) - 2 and t[0] > 0 and t[6] > 0   # every call hands over the record of the call before the previous one
    pn.set_read_timings(2)
    pn.run([pn.make_step(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)) for f in range(36, 44)])
    sums2, calls2 = pn.timing_sums()
    assert calls2 == 8 and all(x >= 0 for x in sums2) and sums2[0] > 0 and sums2[6] > 0
    blocking = pn.reconstruction.GetTimings()
    assert blocking[0] > 0 and blocking[6] > 0


@pytest.mark.parametrize("run_ahead,fused_head,split_pre,handover", [(False, False, False, 1), (True, False, False, 1), (False, True, False, 1),
                                                                     (False, False, True, 1), (True, False, True, 1), (False, False, False, 0),
                                                                     (True, False, True, 0)])
def test_native_driver_matches_oracle(smx, run_ahead, fused_head, split_pre, handover):
    """The C++ frame loop (include/smx_driver.h, written against the shim classes of smx_shim.hpp) produces the
    same state as the oracle; many frames are enqueued by one call.  run_ahead: the preprocessing two steps ahead with
    its dependencies routed through smx_recon_integrate_hooks.  fused_head: bilateral filter + outlier cull in one launch."""
    from surfelmeshing_amd.pipeline import NativeFramePipeline
    from surfelmeshing_amd._lib import IntegrateParams
    s = small_stream(obstacle_until=8)
    pre = small_pre(s.width)
    po = OraclePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre)
    pn = NativeFramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre, IntegrateParams.defaults())
    pn.set_run_ahead(run_ahead)
    pn.set_fused_head(fused_head)
    pn.set_split_preprocessing(split_pre)     # two preprocessing queues: the filter of frame f + 1 beside the cull of frame f
    pn.reconstruction.set_handover_mode(handover)   # front -> internal stream by a device word + gate kernel (1) or by an event (0)
    frames = list(range(4, 20))
    for f in range(0, 24):
        d, c = s.frame(f)
        po.upload(f, d, c)
        pn.upload(f, d, c)
    steps = []
    for f in frames:
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.process(f, others, T, pose)
        steps.append(pn.make_step(f, others, T, pose))
    pn.run(steps)
    n = po.recon.surfels_size
    assert pn.reconstruction.surfels_size() == n
    assert_surfels_match(pn.reconstruction.debug_download_surfels(n), po.recon.surfels(), n)
    d, nrm, rad = pn.download_work()
    assert np.array_equal(d, po.depth_final)
    assert np.array_equal(nrm.view(np.uint32), po.normals.view(np.uint32))
    dd, cc = pn.download_frame(10)
    assert np.array_equal(dd, s.frame(10)[0]) and np.array_equal(cc, s.frame(10)[1])
    pn.release(10)
    with pytest.raises(smx.SmxError):
        pn.download_frame(10)


def test_native_driver_prepared_steps_and_stage_timing(smx):
    """The measurement hooks of the driver leave the results alone: smx_driver_debug_prepare (bench.py --ub hoist-pre:
    the steps preprocessed up front into work images of their own, the run integrates those) gives the map of the plain
    run, and smx_driver_profile_begin/_end (the in-frame stage durations of the bench line) report a positive average
    over the frames they saw."""
    from surfelmeshing_amd.pipeline import NativeFramePipeline, DriverStep
    from surfelmeshing_amd._lib import IntegrateParams
    s = small_stream(obstacle_until=8)
    pre = small_pre(s.width)
    pipes = [NativeFramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre, IntegrateParams.defaults()) for _ in range(2)]
    for f in range(0, 24):
        d, c = s.frame(f)
        for p in pipes:
            p.upload(f, d, c)
    steps = [pipes[0].make_step(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)) for f in range(4, 17)]
    arr = (DriverStep * len(steps))(*steps)
    pipes[0].run_array(arr, len(steps))
    pipes[1].prepare_array(arr, len(steps))
    pipes[1].profile_begin(1, 8)
    pipes[1].run_array(arr, len(steps))
    ms_prepared, n_prepared = pipes[1].profile_end()
    assert n_prepared == 0        # nothing preprocessed while the prepared steps ran
    n = pipes[0].reconstruction.surfels_size()
    assert n > 1000 and pipes[1].reconstruction.surfels_size() == n
    assert_surfels_match(pipes[1].reconstruction.debug_download_surfels(n), pipes[0].reconstruction.debug_download_surfels(n), n,
                         exact=True)
    # (the working-buffer getters refer to the driver's own three sets, which a prepared run does not touch)
    # a plain continuation with the stage timer on: one more frame per stage, each seen
    for stage, f in enumerate((17, 18, 19)):
        one = (DriverStep * 1)(pipes[0].make_step(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)))
        pipes[0].profile_begin(stage, 8)
        pipes[0].run_array(one, 1)
        ms, seen = pipes[0].profile_end()
        assert seen == 1 and 0.0005 < ms < 5000.0, (stage, ms, seen)   # (a duration; the first launch of a kernel in a process may load its code)


@pytest.mark.parametrize("overlap,staged", [(True, True), (True, False), (False, True)])
def test_native_driver_streamed_uploads(smx, overlap, staged):
    """smx_driver_run_streamed: the frames arrive from (page-locked) host memory with the frame loop
    (APP/main.cc:905-984).  Frame f+4 is copied in for the step of frame f -- the first one over a slot that holds
    zeros, one over a slot that the steps in flight read -- and the map equals the oracle's.  staged: the copies are kernels
    on a staging queue of their own (page-locked sources; pageable ones fall back to the copy engine in the step's queue)."""
    from surfelmeshing_amd.pipeline import NativeFramePipeline
    from surfelmeshing_amd._lib import IntegrateParams
    s = small_stream(obstacle_until=8)
    pre = small_pre(s.width)
    po = OraclePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre)
    pn = NativeFramePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 60000, pre, IntegrateParams.defaults())
    pn.set_overlap(overlap)
    pn.set_staged_uploads(staged)
    for f in range(0, 30):
        d, c = s.frame(f)
        po.upload(f, d, c)
        if f < 8:
            pn.upload(f, d, c)
    pn.upload(8, np.zeros((s.height, s.width), np.uint16), np.zeros((s.height, s.width, 3), np.uint8))
    pn.upload(21, *s.frame(21))
    keep, steps, uploads = [], [], []
    for f in range(4, 20):
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.process(f, others, T, pose)
        steps.append(pn.make_step(f, others, T, pose))
        src = f + 4
        if f == 17:                                      # frame 21 is resident already: copy frame 13 again instead,
            src = 13                                     # a slot the steps up to this one read (ordered after them)
        d, c = s.frame(src)
        if f % 3 == 0:                                   # pageable memory works too (the copy then blocks the host)
            hd, hc = np.ascontiguousarray(d), np.ascontiguousarray(c)
        elif f == 10:                                    # depth page-locked, colour not: the frame takes ONE route (copy engine)
            pd = smx.PagelockedArray(d.shape, np.uint16)
            pd.array[...] = d
            keep.append(pd)
            hd, hc = pd.array, np.ascontiguousarray(c)
        else:
            pd, pc = smx.PagelockedArray(d.shape, np.uint16, write_combined=(f % 4 == 1)), smx.PagelockedArray(c.shape, np.uint8)
            pd.array[...] = d
            pc.array[...] = c
            keep += [pd, pc]
            hd, hc = pd.array, pc.array
        uploads.append((src, hd, hc))
    pn.run_streamed(steps[:9], uploads[:9])
    pn.run_streamed(steps[9:], uploads[9:])          # a second call continues the stream
    smx.StreamSynchronize(None)
    # every frame took one route, and the driver says which (a loop that believes it stages and does not would only be slower)
    pinned = sum(1 for f in range(4, 20) if f % 3 != 0 and f != 10)
    assert pn.upload_counts() == ((pinned, 16 - pinned) if (overlap and staged) else (0, 16))
    n = po.recon.surfels_size
    assert pn.reconstruction.surfels_size() == n
    assert_surfels_match(pn.reconstruction.debug_download_surfels(n), po.recon.surfels(), n)
    dd, cc = pn.download_frame(8)
    assert np.array_equal(dd, s.frame(8)[0]) and np.array_equal(cc, s.frame(8)[1])
    # steps without an upload, and the argument checks
    others, T, pose = s.outlier_frames(20), s.others_TR_reference(20), s.pose(20)
    po.process(20, others, T, pose)
    pn.upload(24, *s.frame(24))
    pn.run_streamed([pn.make_step(20, others, T, pose)], [None])
    assert_surfels_match(pn.reconstruction.debug_download_surfels(po.recon.surfels_size), po.recon.surfels(), po.recon.surfels_size)


def test_loop_closure_deformation_hook(smx):
    """The hook the reference describes but does not ship (README.md:152-176, call site main.cc:1194-1200): a rigid
    correction per creation frame, applied between two frames of a stream.  The map right after the deformation and
    after every following frame equals the oracle's (moved surfels now disagree with the measurements: conflicts,
    merges and re-activated old surfels follow); the changed-surfel delta covers every moved slot."""
    s = small_stream(obstacle_until=8, yaw_deg_per_frame=2.0)
    po, pg = _pipes(smx, s, 60000, params_kw=dict(surfel_integration_active_window_size=6))
    rec = pg.reconstruction
    rec.SetDeltaTracking(None, True)
    run_both(po, pg, s, list(range(4, 12)), None)
    _compare_state(po, pg)
    rec.TransferChangedToCPU(None, 11)                          # (drain: everything so far)
    n = po.recon.surfels_size
    before = po.recon.surfels().copy()
    a = np.deg2rad(0.4)
    T = np.tile(np.eye(4, dtype=np.float32)[:3].reshape(1, 12), (10, 1))       # creation frames 0..9; 10, 11 stay
    for c in range(4, 8):
        T[c] = np.array([[np.cos(a), 0, np.sin(a), 0.004], [0, 1, 0, -0.002], [-np.sin(a), 0, np.cos(a), 0.006]],
                        np.float32).reshape(12)
    reactivate = np.zeros(10, np.uint8)
    reactivate[4:6] = 1
    po.recon.deform_by_creation_frame(T, reactivate, 12)
    rec.DeformByCreationFrame(None, T, reactivate, 12)
    _compare_state(po, pg, check_scratch=False)
    after = po.recon.surfels()
    live = before[7, :n] >= 0
    creation = before[17, :n].view(np.uint32)
    moved = live & (creation >= 4) & (creation < 8)
    assert moved.sum() > 3000 and (live & ~moved).sum() > 1000
    assert np.array_equal(after[:, :n][:, ~moved].view(np.uint32), before[:, :n][:, ~moved].view(np.uint32))
    assert np.all(after[0, :n][moved] != before[0, :n][moved])
    assert np.all(after[18, :n].view(np.uint32)[live & (creation >= 4) & (creation < 6)] == 12)
    delta = rec.TransferChangedToCPU(None, 12)
    assert np.array_equal(delta.surfel_index[:delta.count], np.flatnonzero(moved).astype(np.uint32))
    # the stream goes on against the deformed map
    for f in range(8, 26):
        d, c = s.frame(f)
        po.upload(f, d, c)
        pg.upload(f, d, c)
    for f in range(12, 22):
        others, Tr, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.process(f, others, Tr, pose)
        pg.process(f, others, Tr, pose)
        _compare_state(po, pg)
    assert po.recon.merge_count > 0
    # nothing to do / argument errors
    rec.DeformByCreationFrame(None, np.zeros((0, 12), np.float32))
    with pytest.raises(ValueError):
        rec.DeformByCreationFrame(None, T, np.zeros(3, np.uint8), 0)


def test_export_obj_and_ply_files(smx, tmp_path):
    """SaveMeshAsOBJ / SavePointCloudAsPLY (APP/main.cc:128-203) from the GPU map: vertices = the live surfels in slot
    order with the oracle's positions, colours and normals; merged surfels dropped and triangle indices renumbered."""
    from surfelmeshing_amd import export
    s = small_stream(obstacle_until=8)
    po, pg = _pipes(smx, s, 60000)
    run_both(po, pg, s, list(range(4, 14)), None)
    rec = pg.reconstruction
    pos, col = po.recon.export_vertices()
    pos, col = pos.reshape(-1, 3), col.reshape(-1, 3)
    live = ~np.isnan(pos[:, 0])
    n = live.size
    assert (~live).sum() > 20
    merged = np.flatnonzero(~live)
    tris = np.array([[0, 1, 2], [merged[0], 0, 1], [n - 1, n - 2, n - 3], [5, n + 3, 6]], np.int64)
    obj = str(tmp_path / "mesh.obj")
    assert export.SaveMeshAsOBJ(rec, obj, None, tris)
    lines = open(obj).read().splitlines()
    v = [ln for ln in lines if ln.startswith("v ")]
    f = [ln for ln in lines if ln.startswith("f ")]
    assert len(v) == live.sum() and len(v) + len(f) == len(lines)
    k = np.float32(1) / np.float32(255)
    for j, i in list(enumerate(np.flatnonzero(live)))[::97]:
        want = "v " + " ".join("%g" % x for x in list(pos[i]) + list(col[i].astype(np.float32) * k))
        assert v[j] == want, (i, v[j], want)
    remap = np.cumsum(live) - 1
    keep = [t for t in tris if all(0 <= a < n and live[a] for a in t)]
    assert len(keep) == 2 and f == ["f %d %d %d" % tuple(remap[t] + 1) for t in keep]
    ply = str(tmp_path / "cloud.ply")
    assert export.SavePointCloudAsPLY(rec, ply, None, export_colors=True)
    r = export.read_ply(ply)
    t = po.recon.transfer_all()
    assert r.size == live.sum()
    for name, row in (("x", "x"), ("y", "y"), ("z", "z"), ("nx", "normal_x"), ("ny", "normal_y"), ("nz", "normal_z")):
        assert np.array_equal(r[name].view(np.uint32), t[row][live].view(np.uint32)), name
    assert np.array_equal(np.stack([r["red"], r["green"], r["blue"]], 1), col[live])
    export.SavePointCloudAsPLY(rec, ply)                                  # the reference writes white (main.cc:194)
    assert np.all(export.read_ply(ply)["green"] == 255)


def test_tum_dataset_through_the_pipeline(smx, tmp_path):
    """Input side end to end: a TUM-format folder (PNG files, associated.txt, calibration.txt, a
    trajectory sampled between the frames) read by surfelmeshing_amd.tum, fed to the GPU pipeline and to the oracle
    with the reader's interpolated poses and the caller-side relative poses of APP/main.cc:1037-1059."""
    from scipy.spatial.transform import Rotation
    from surfelmeshing_amd import tum
    from surfelmeshing_amd.pipeline import FramePipeline, others_TR_reference
    from surfelmeshing_amd._lib import IntegrateParams
    s = small_stream(obstacle_until=6)
    n = 16
    stamps = [50.0 + f / 30.0 for f in range(n)]
    traj = []
    for k in range(-1, 2 * n + 1):                                         # poses at twice the frame rate, offset
        t = 50.0 + (k + 0.37) / 60.0
        T = np.asarray(s.pose64(k / 2.0 + 0.37 / 2.0)[0]), np.asarray(s.pose64(k / 2.0 + 0.37 / 2.0)[1])
        traj.append((t, T[1], Rotation.from_matrix(T[0]).as_quat()))
    folder = str(tmp_path / "seq")
    tum.write_tum_dataset(folder, [s.frame(f) for f in range(n)], stamps, (s.fx, s.fy, s.cx - 0.5, s.cy - 0.5), traj)
    video = tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, "groundtruth.txt")
    assert video.frame_count() == n
    cam = video.depth_camera
    fx, fy, cx, cy = cam.parameters()
    pre = small_pre(cam.width())
    po = OraclePipeline(cam.width(), cam.height(), fx, fy, cx, cy, 60000, pre)
    pg = FramePipeline(cam.width(), cam.height(), fx, fy, cx, cy, 60000, pre, IntegrateParams.defaults())
    for f in range(n):
        d, c = video.depth_frame(f).GetImage(), video.color_frame(f).GetImage()
        assert np.array_equal(d, s.frame(f)[0])
        po.upload(f, d, c)
        pg.upload(f, d, c)
    for f in range(4, n - 4):
        others = [f - k for k in range(1, 5)] + [f + k for k in range(1, 5)]
        G = video.depth_frame(f).global_T_frame()
        assert np.allclose(G[:, :3] @ G[:, :3].T, np.eye(3), atol=1e-6)
        assert np.allclose(G, np.asarray(s.pose(f)).reshape(3, 4), atol=2e-3)        # interpolated, close to the true pose
        T = others_TR_reference(G, [video.depth_frame(g).global_T_frame() for g in others], pre.depth_scaling)
        po.process(f, others, T, G)
        pg.process(f, others, T, G)
        _compare_state(po, pg)
    assert po.recon.surfels_size > 5000
