"""Per-kernel time of the blend kernel as a function of the blending radius (640x480, small map)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch  # noqa
from common import small_stream, small_pre
from surfelmeshing_amd import api, _lib
from surfelmeshing_amd.pipeline import FramePipeline
from surfelmeshing_amd._lib import IntegrateParams
_lib.require_gpu()
s = small_stream(640, 480, obstacle_until=6)
for radius in (2, 3, 6, 9, 12, 17):
    pg = FramePipeline(640, 480, s.fx, s.fy, s.cx, s.cy, 2_000_000, small_pre(640), IntegrateParams.defaults(measurement_blending_radius=radius))
    for f in range(0, 30):
        d, c = s.frame(f); pg.upload(f, d, c)
    rec = pg.reconstruction
    rec.set_overlap(False); rec.set_stats_enabled(False)
    for f in range(4, 14):
        pg.process(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f))
    rec.set_timing_enabled(2)
    names = rec.kernel_time_names(); acc = np.zeros(len(names))
    for f in range(14, 24):
        pg.process(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f))
        acc += np.array(rec.kernel_times_ms())
    print('radius %2d: blend slot %.1f us, association tiles %.1f us' % (radius, acc[names.index('blend')] * 100, acc[names.index('assoc_tiles')] * 100))
    pg.reconstruction.close()
