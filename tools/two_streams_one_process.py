#!/usr/bin/env python
"""Two independent C2 streams (two native frame loops, two maps) driven from two host threads of ONE process on one GPU:
does a second stream find idle capacity beside the first?  (Measurement, round 4; the two-process variant over gloo pays
for the processes' queues being time-sliced against each other.)     python tools/two_streams_one_process.py [steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import bench
from surfelmeshing_amd import api, _lib
_lib.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def prepare(seed, phase):
    wl = bench.Workload(api, 640, 480, 5_000_000, 6_250_000, seed, phase)
    g_end, n_live = wl.grow(False)
    first = g_end + 10
    for j in range(-4, 64 + 3 * K + 8):
        wl.render(first + j, 4 + j)
    plan = [wl.plan(first + j, 4 + j) for j in range(64 + 3 * K)]
    api.StreamSynchronize(None)
    wl.pipe.stream = api.Stream(None)      # its own caller stream
    rec = wl.pipe.reconstruction
    rec.set_stats_enabled(False); rec.set_timing_enabled(0)
    wl.pipe.run_array(*wl.steps(plan[:64]))
    api.StreamSynchronize(wl.pipe.stream)
    return wl, [wl.steps(plan[64 + r * K:64 + (r + 1) * K]) for r in range(3)]


def timed(pairs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda wl=wl, st=st: wl.pipe.run_array(*st)) for wl, st in pairs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


a = prepare(0x5EED0001, 0.0)
b = prepare(0x5EED0002, 0.37)
# three consecutive windows of K frames per stream: [0] each stream alone, [1] both together, [2] each alone again
ta0 = timed([(a[0], a[1][0])]); tb0 = timed([(b[0], b[1][0])])
t12 = timed([(a[0], a[1][1]), (b[0], b[1][1])])
ta2 = timed([(a[0], a[1][2])]); tb2 = timed([(b[0], b[1][2])])
print("alone: A %.1f, B %.1f frames/s | both together: %.1f frames/s in sum | alone again: A %.1f, B %.1f" %
      (K / ta0, K / tb0, 2 * K / t12, K / ta2, K / tb2))
