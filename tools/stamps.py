#!/usr/bin/env python
"""Phase times inside the association-tile and blend kernels (a build with SMX_EXTRA_FLAGS=-DSMX_STAMPS): runs the
bench workload for a few frames and prints, per kernel, the mean / max over the workgroups of the shader-clock
differences between the stamps of the LAST frame.   SMX_EXTRA_FLAGS=-DSMX_STAMPS python -m surfelmeshing_amd.build --force;
python tools/stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from surfelmeshing_amd import _lib, api  # noqa: E402


def main():
    L = _lib.load()
    if not True:
        sys.exit("libsmx.so was not built with -DSMX_STAMPS")
    wl = bench.Workload(api, 640, 480, int(os.environ.get("SMX_STAMP_LIVE", 5_000_000)), 6_250_000, 0x5EED0002, 0.0)
    g_end, n_live = wl.grow(False)
    first = g_end + 10
    for j in range(-4, 40):
        wl.render(first + j, 4 + j)
    plan = [wl.plan(first + j, 4 + j) for j in range(30)]
    if os.environ.get("SMX_STAMP_ALONE"):   # every kernel alone on the chip: no frame pipelining, one frame at a time
        wl.pipe.reconstruction.set_overlap(False)
        for step in plan:
            wl.pipe.run_array(*wl.steps([step]))
            api.StreamSynchronize(None)
    else:
        wl.pipe.run_array(*wl.steps(plan))
    api.StreamSynchronize(None)
    out = np.zeros((3, 8192, 16), np.uint64)
    _lib.check(L.smx_recon_debug_download_stamps(wl.pipe.reconstruction._h, out.ctypes.data_as(C.c_void_p)))
    for name, a, n in (("k_assoc_tiles", out[0], 6), ("k_blend_tiles", out[1], 6)):
        a = a[a[:, 0] > 0][:, :n].astype(np.int64)
        d = np.diff(a, axis=1)
        t0 = a[:, 0].min()
        print("%s: %d workgroups; kernel span %.1f us (first stamp -> last stamp, 100 MHz.. see note)" % (name, len(a), (a[:, n - 1].max() - t0) / 100.0))
        print("   phase mean:", " ".join("%8.0f" % x for x in d.mean(axis=0)))
        print("   phase max: ", " ".join("%8.0f" % x for x in d.max(axis=0)))
        print("   start offset of workgroups (mean / max): %.0f / %.0f" % ((a[:, 0] - t0).mean(), (a[:, 0] - t0).max()))
        print("   end-to-end per workgroup (mean / max): %.0f / %.0f" % ((a[:, n - 1] - a[:, 0]).mean(), (a[:, n - 1] - a[:, 0]).max()))
    # the edge kernel (k_reg_accumulate): wall clocks (100 MHz) at entry / exit of every workgroup, steps, entries, largest step
    a = out[2]
    a = a[a[:, 0] > 0].astype(np.int64)
    if len(a):
        t0 = a[:, 0].min()
        dur = (a[:, 1] - a[:, 0]) / 100.0
        print("k_reg_accumulate: %d workgroups, %d walk steps; span %.1f us; workgroup duration mean %.1f / median %.1f / max %.1f us; start offset mean %.1f / max %.1f us"
              % (len(a), int(a[0, 5]), (a[:, 1].max() - t0) / 100.0, dur.mean(), np.median(dur), dur.max(), ((a[:, 0] - t0) / 100.0).mean(), ((a[:, 0] - t0) / 100.0).max()))
        print("   steps per workgroup: " + " ".join("%d:%d" % (k, (a[:, 2] == k).sum()) for k in range(0, int(a[:, 2].max()) + 1)))
        for lo, hi in ((0, 1), (1, 65), (65, 257), (257, 513), (513, 769), (769, 1025), (1025, 1 << 30)):
            m = (a[:, 3] >= lo) & (a[:, 3] < hi)
            if m.any():
                print("   workgroups with %5d <= entries < %5d: %5d, duration mean %.1f / max %.1f us, exit offset mean %.1f / max %.1f us"
                      % (lo, min(hi, 99999), m.sum(), dur[m].mean(), dur[m].max(), ((a[m, 1] - t0) / 100.0).mean(), ((a[m, 1] - t0) / 100.0).max()))
        order = np.argsort(a[:, 1])[-12:]
        print("   the last 12 to leave: " + "; ".join("wg %d steps %d entries %d max %d dur %.1f exit %.1f" % (i, a[i, 2], a[i, 3], a[i, 4], dur[i], (a[i, 1] - t0) / 100.0) for i in order))


if __name__ == "__main__":
    main()
