#!/usr/bin/env python
"""How full are the walk steps of the list kernels?  Grows the C2 map like bench.py, runs 30 frames of the timed window and
counts, per segment of 1024 slots, the slots integrated in the last frame (a lower bound of the visible ones): a segment's
visible slots are listed in chunks of 256, one walk step (= one 256-lane workgroup round) per chunk."""
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
last = int(stamp.max())
vis = stamp == last
cnt = np.add.reduceat(vis.astype(np.int64), np.arange(0, n, 1024))
seg = cnt[cnt > 0]
chunks = (seg + 255) // 256
print("slots %d, integrated in the last frame %d, in %d segments; chunks (walk steps) %d, mean fill %.1f of 256 lanes (%.0f %%); dense packing would need %d steps"
      % (n, vis.sum(), seg.size, chunks.sum(), vis.sum() / chunks.sum(), 100.0 * vis.sum() / chunks.sum() / 256, (vis.sum() + 255) // 256))
h = np.bincount(np.minimum(seg // 64, 16))
print("segments by integrated slots (bins of 64):", h.tolist())
lastc = seg - (chunks - 1) * 256
print("last-chunk fill: p10 %d p50 %d p90 %d" % tuple(np.percentile(lastc, [10, 50, 90])))
