"""The oracle's three input-side image functions against the REFERENCE's own function text, compiled
from /root/reference by oracle/ref_build.py into oracle/_ref/libsmx_ref_image.so (CPU code; the library travels with the
repository, the reference does not): MedianFilterAndDensifyDepthMap (APP/main.cc:207-252),
Image<T>::DownscaleUsingMedianWhileExcluding (VIS/image.h:1003-1053) and Image<Vec3u8>::DownscaleToHalfSize
(VIS/image.h:929-948, the colour pyramid's step).  Bit-exact on random, sparse, ragged and degenerate images."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as orc
from common import ROOT

SO = os.path.join(ROOT, "oracle", "_ref", "libsmx_ref_image.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libsmx_ref_image.so not built (no reference sources here)")


@pytest.fixture(scope="module")
def ref():
    return C.CDLL(SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _depth_images(rng):
    for (h, w, p_zero) in ((48, 64, 0.0), (48, 64, 0.3), (37, 53, 0.7), (5, 3, 0.5), (1, 9, 0.2), (2, 2, 0.0), (60, 80, 0.97)):
        d = rng.integers(300, 9000, (h, w)).astype(np.uint16)
        d[rng.random((h, w)) < p_zero] = 0
        yield d
    flat = np.full((20, 30), 1234, np.uint16)      # ties everywhere: the even-count branch picks by distance to the average
    flat[::3, ::2] = 1236
    yield flat
    yield np.zeros((8, 8), np.uint16)


def test_median_filter_and_densify_matches_the_reference_text(ref):
    rng = np.random.default_rng(5)
    for d in _depth_images(rng):
        out = np.empty_like(d)
        ref.ref_median_filter_and_densify(C.c_int(d.shape[1]), C.c_int(d.shape[0]), _p(d), _p(out))
        assert np.array_equal(orc.median_filter_and_densify(d, 1), out), d.shape
        # (APP/main.cc:929-939 applies it several times)
        out2 = np.empty_like(d)
        ref.ref_median_filter_and_densify(C.c_int(d.shape[1]), C.c_int(d.shape[0]), _p(out), _p(out2))
        assert np.array_equal(orc.median_filter_and_densify(d, 2), out2), d.shape


def test_downscale_using_median_matches_the_reference_text(ref):
    rng = np.random.default_rng(6)
    for d in _depth_images(rng):
        h, w = d.shape
        for (oh, ow) in ((max(1, h // 2), max(1, w // 2)), (max(1, h // 3), max(1, w // 4)), (h, w), (1, 1)):
            for ignore in (0, 1234):
                out = np.empty((oh, ow), np.uint16)
                ref.ref_downscale_using_median_while_excluding(C.c_uint16(ignore), C.c_int(w), C.c_int(h), _p(d), C.c_int(ow), C.c_int(oh), _p(out))
                mine = orc.downscale_using_median_while_excluding(d, ow, oh, ignore)
                assert np.array_equal(mine, out), (d.shape, oh, ow, ignore)


def test_color_pyramid_matches_the_reference_text(ref):
    rng = np.random.default_rng(7)
    for (h, w) in ((48, 64), (16, 8), (2, 2), (120, 160)):
        c = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        c[: h // 4] = 255                                   # 4 x 63 = 252: the truncating quarters never reach 255
        level, cur = 0, c
        while cur.shape[0] % 2 == 0 and cur.shape[1] % 2 == 0 and level < 3:
            nxt = np.empty((cur.shape[0] // 2, cur.shape[1] // 2, 3), np.uint8)
            ref.ref_downscale_to_half_size_rgb(C.c_int(cur.shape[1]), C.c_int(cur.shape[0]), _p(np.ascontiguousarray(cur)), _p(nxt))
            level, cur = level + 1, nxt
            assert np.array_equal(orc.color_image_pyramid(c, level), cur), (h, w, level)
