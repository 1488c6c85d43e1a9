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
    out = np.zeros((2, 8192, 16), np.uint64)
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


if __name__ == "__main__":
    main()
