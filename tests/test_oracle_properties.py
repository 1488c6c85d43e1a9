"""Size-independent properties of the oracle's restatements (hypothesis-driven), so that the checker itself is
checked beyond its hand-made known answers."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle as orc

_SET = dict(max_examples=40, deadline=None)


def _depth(draw, h, w, hole=0.4):
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    d = rng.integers(300, 20000, (h, w)).astype(np.uint16)
    d[rng.random((h, w)) < hole] = 0
    return d


@settings(**_SET)
@given(st.data())
def test_downscale_median_picks_a_value_of_its_block(data):
    """VIS/image.h:1003-1053: every output is one of the block's valid values (never an average), 0 iff the block has
    none; with odd counts it is the median."""
    bh, bw = data.draw(st.integers(1, 4)), data.draw(st.integers(1, 4))
    oh, ow = data.draw(st.integers(1, 9)), data.draw(st.integers(1, 9))
    d = _depth(data.draw, oh * bh, ow * bw)
    out = orc.downscale_using_median_while_excluding(d, ow, oh, 0)
    for y in range(oh):
        for x in range(ow):
            v = d[y * bh:(y + 1) * bh, x * bw:(x + 1) * bw].ravel()
            v = np.sort(v[v != 0])
            if v.size == 0:
                assert out[y, x] == 0
            elif v.size % 2:
                assert out[y, x] == v[v.size // 2]
            else:
                assert out[y, x] in (v[v.size // 2 - 1], v[v.size // 2])


@settings(**_SET)
@given(st.data())
def test_color_pyramid_bounds_and_level_composition(data):
    """VIS/image.h:929-948 per level: min - 3 <= out <= max of the 2x2 block (four truncated quarters); L levels in one
    call equal L calls of one level."""
    level = data.draw(st.integers(1, 3))
    h, w = data.draw(st.integers(1, 5)) << level, data.draw(st.integers(1, 5)) << level
    rng = np.random.default_rng(data.draw(st.integers(0, 2 ** 31 - 1)))
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    one = orc.color_image_pyramid(img, 1).astype(np.int32)
    blocks = img.reshape(h // 2, 2, w // 2, 2, 3).astype(np.int32)
    assert np.all(one <= blocks.max(axis=(1, 3))) and np.all(one >= blocks.min(axis=(1, 3)) - 3)
    step = img
    for _ in range(level):
        step = orc.color_image_pyramid(step, 1)
    assert np.array_equal(orc.color_image_pyramid(img, level), step)


@settings(**_SET)
@given(st.data())
def test_median_densify_keeps_the_valid_set_growing(data):
    """APP/main.cc:206-252: a pixel that was valid stays valid (its value becomes a median of its neighbourhood's
    valid values), holes are only filled -- so the valid set never shrinks, and every output value occurs in the
    input's 3x3 neighbourhood."""
    h, w = data.draw(st.integers(3, 12)), data.draw(st.integers(3, 12))
    d = _depth(data.draw, h, w, hole=data.draw(st.floats(0.0, 0.9)))
    out = orc.median_filter_and_densify(d, 1)
    assert np.all((out > 0) | (d == 0))
    pad = np.pad(d, 1)
    for y in range(h):
        for x in range(w):
            if out[y, x]:
                assert out[y, x] in pad[y:y + 3, x:x + 3]


@settings(**_SET)
@given(st.data())
def test_check_triangles_vertex_rotation(data):
    """APP/surfel_meshing.cc:590-650: the long-edge condition does not depend on which vertex comes first, and the
    per-pivot normal bits rotate with the vertex order."""
    rng = np.random.default_rng(data.draw(st.integers(0, 2 ** 31 - 1)))
    n = 12
    rows = [rng.normal(size=n).astype(np.float32) * 0.05 for _ in range(3)]
    r2 = (rng.random(n).astype(np.float32) * 0.002 + 1e-5)
    r2[rng.random(n) < 0.15] = -1
    nrm = rng.normal(size=(3, n)).astype(np.float32)
    tris = rng.integers(0, n, (30, 3)).astype(np.uint32)
    factor = float(data.draw(st.sampled_from([0.25, 1.0, 4.0, 16.0])))
    f0 = orc.check_triangles(*rows, r2, *nrm, tris, factor)
    f1 = orc.check_triangles(*rows, r2, *nrm, tris[:, [1, 2, 0]], factor)
    assert np.array_equal(f0 & 17, f1 & 17)                                 # long edge + merged bits
    p0 = (f0 >> 1) & 7
    p1 = (f1 >> 1) & 7
    # pivot k of the rotated triangle is pivot k + 1 of the original
    assert np.array_equal(p1 & 1, (p0 >> 1) & 1) and np.array_equal((p1 >> 1) & 1, (p0 >> 2) & 1)
    assert np.array_equal((p1 >> 2) & 1, p0 & 1)


@settings(max_examples=15, deadline=None)
@given(st.data())
def test_deformation_and_its_inverse(data):
    """README.md:152-176 hook: identity rows change nothing bit for bit; a rigid correction followed by its inverse
    brings positions and normals back (to float rounding), smooth - raw offsets are preserved exactly by neither but
    to rounding, and radii / confidences / links are never touched."""
    from scipy.spatial.transform import Rotation
    h, w = 12, 16
    rec = orc.Recon(2000, w, h, 60.0, 60.0, 8.0, 6.0)
    rng = np.random.default_rng(data.draw(st.integers(0, 2 ** 31 - 1)))
    depth = np.full((h, w), 10000, np.uint16)
    normals = np.zeros((h, w, 2), np.float32)
    radius = np.full((h, w), 0.02, np.float32)
    color = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    ident = np.eye(4, dtype=np.float32)[:3].reshape(12)
    rec.integrate(2, 5000.0, depth.copy(), normals, radius, color, ident)
    n = rec.surfels_size
    assert n > 50
    before = rec.surfels().copy()
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.2).as_matrix()
    t = rng.normal(size=3) * 0.3
    T = np.tile(ident, (3, 1))
    T[2] = np.concatenate([R, t[:, None]], 1).astype(np.float32).reshape(12)
    Ti = np.tile(ident, (3, 1))
    Ti[2] = np.concatenate([R.T, (-R.T @ t)[:, None]], 1).astype(np.float32).reshape(12)
    rec.deform_by_creation_frame(np.tile(ident, (3, 1)))
    assert np.array_equal(rec.surfels().view(np.uint32), before.view(np.uint32))
    rec.deform_by_creation_frame(T)
    mid = rec.surfels().copy()
    assert not np.allclose(mid[0:3, :n], before[0:3, :n], atol=1e-3)
    assert np.allclose(np.linalg.norm(mid[8:11, :n], axis=0), 1.0, atol=1e-5)
    rec.deform_by_creation_frame(Ti)
    after = rec.surfels()
    assert np.allclose(after[0:6, :n], before[0:6, :n], atol=2e-5) and np.allclose(after[8:11, :n], before[8:11, :n], atol=1e-5)
    keep = [6, 7, 17, 18, 19, 20, 21, 22, 24]
    assert np.array_equal(after[keep].view(np.uint32), before[keep].view(np.uint32))


def test_row_parallel_depth_stages_are_identical():
    """The all-host-cores CPU baseline of bench.py splits the rows of the per-pixel stages over threads
    (orc_set_row_range, thread-local): every image equals the single-threaded one, for band counts that do and do not
    divide the height, and the calling thread's own range stays "all rows"."""
    from common import small_pre, small_stream
    from oracle import binding
    from oracle_pipeline import OraclePipeline
    s = small_stream(200, 77)
    pre = small_pre(200)
    imgs = {}
    for threads in (1, 3, 8):
        binding.set_row_threads(threads)
        try:
            po = OraclePipeline(s.width, s.height, s.fx, s.fy, s.cx, s.cy, 1000, pre)
            for f in range(0, 9):
                po.upload(f, *s.frame(f))
            po.preprocess(4, s.outlier_frames(4), s.others_TR_reference(4))
            imgs[threads] = (po.stages["bilateral"].copy(), po.stages["outlier"].copy(), po.stages["erode"].copy(),
                             po.depth_final.copy(), po.normals.copy(), po.radius.copy())
        finally:
            binding.set_row_threads(1)
    for threads in (3, 8):
        for a, b in zip(imgs[1], imgs[threads]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), threads
    assert (imgs[1][3] > 0).sum() > 1000
