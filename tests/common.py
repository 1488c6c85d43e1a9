"""Shared helpers of the test-suite (test infrastructure)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from surfelmeshing_amd.pipeline import PreprocessParams  # noqa: E402
from surfelmeshing_amd.synth import SyntheticStream  # noqa: E402

# rows of the surfel SoA that hold results (scratch rows 11-16, 23 are excluded)
RESULT_ROWS = [r for r in range(25) if r not in (11, 12, 13, 14, 15, 16, 23)]
FLOAT_ROWS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
INT_ROWS = [17, 18, 19, 20, 21, 22, 24]


def small_stream(w=160, h=120, **kw):
    sc = w / 640.0
    return SyntheticStream(width=w, height=h, fx=525.0 * sc, fy=525.0 * sc, cx=320.0 * sc, cy=240.0 * sc, **kw)


def small_pre(w=160, **kw):
    return PreprocessParams(max_depth=10.0, depth_valid_region_radius=333.0 * w / 640.0, **kw)


def assert_surfels_match(gpu_rows, orc_rows, n, float_rtol=1e-4, exact=True):
    """Parity bar of BASELINE.json: indices/counts bit-exact, floats within 1e-4 relative.
    With exact=True (the default: both sides share the arithmetic contract) floats must be bit-equal."""
    g, o = gpu_rows[:, :n], orc_rows[:, :n]
    for r in INT_ROWS:
        a, b = g[r].view(np.uint32), o[r].view(np.uint32)
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, "row %d: %d integer mismatches, first at surfel %d (%d vs %d)" % (
            r, bad.size, bad[0], a[bad[0]], b[bad[0]])
    for r in FLOAT_ROWS:
        a, b = g[r], o[r]
        if exact:
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            assert bad.size == 0, "row %d: %d float bit mismatches, first at surfel %d (%r vs %r)" % (
                r, bad.size, bad[0], a[bad[0]], b[bad[0]])
        else:
            scale = np.maximum(np.abs(b), 1e-3)
            err = np.abs(a - b) / scale
            assert np.all(err <= float_rtol), "row %d: max rel err %g at surfel %d" % (r, err.max(), err.argmax())


def run_both(po, pg, s, frames, on_frame=None):
    """Feed the same frames to the oracle pipeline `po` and the HIP pipeline `pg`."""
    lo, hi = min(frames) - 4, max(frames) + 4
    for f in range(lo, hi + 1):
        d, c = s.frame(f)
        po.upload(f, d, c)
        pg.upload(f, d, c)
    for f in frames:
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        po.process(f, others, T, pose)
        pg.process(f, others, T, pose)
        if on_frame:
            on_frame(f)
