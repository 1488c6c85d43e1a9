#!/usr/bin/env python
"""Where in the slot range are the surfels a frame works on?  Grows the C2 map like bench.py, runs 30 frames of the timed
window and prints, per tenth of the slot range, the share of slots updated in the last 30 frames (the regulariser's
window) and the share of segments that hold at least one of them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from surfelmeshing_amd import api  # noqa: E402

wl = bench.Workload(api, 640, 480, 5_000_000, 6_250_000, 0x5EED0002, 0.0)
g_end, n_live = wl.grow(False)
first = g_end + 10
for j in range(-4, 40):
    wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(30)]
wl.pipe.run_array(*wl.steps(plan))
api.StreamSynchronize(None)
rows = wl.pipe.reconstruction.debug_download_surfels()
stamp = rows[18].view(np.uint32).astype(np.int64)
n = stamp.size
last = first + 29
recent = stamp > last - 30
print("slots %d, growth frames %d, recent %d" % (n, g_end, recent.sum()))
seg = np.add.reduceat(recent.astype(np.int64), np.arange(0, n, 1024))
for d in range(10):
    a, b = d * n // 10, (d + 1) * n // 10
    sa, sb = a // 1024, b // 1024
    print("slots %3d-%3d %%: recent %6.2f %% of all recent, segments with a recent slot %5.1f %%" %
          (10 * d, 10 * d + 10, 100.0 * recent[a:b].sum() / max(1, recent.sum()), 100.0 * (seg[sa:sb] > 0).mean()))
