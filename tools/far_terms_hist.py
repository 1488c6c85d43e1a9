#!/usr/bin/env python
"""How are the regulariser's far terms spread over the slot range?  Grows the C2 map like bench.py, runs 30 frames of the
timed window, downloads the map and classifies every in-window edge the way k_reg_accumulate does: target in the source's
own 1024-slot segment (LDS sum), target lists the source back (inbox store), neither (the packed atomics).  For the last
class it prints how many distinct destination segments one source workgroup addresses and how many terms one destination
segment receives: the numbers a per-destination bin scheme would live on."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from surfelmeshing_amd import api  # noqa: E402

SEG = 1024
REG_WINDOW = 30  # bench.py's regulariser window


def main():
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
    T = np.stack([rows[19 + q].view(np.uint32).astype(np.int64) for q in range(4)], 1)
    n = stamp.size
    last = first + 29
    recent = ~(stamp < last - REG_WINDOW)
    print("slots %d, recent %d" % (n, recent.sum()))
    valid = T != 0xFFFFFFFF
    Tc = np.where(valid, T, 0)
    inwin = valid & recent[Tc]
    src = np.repeat(np.arange(n)[:, None], 4, 1)
    near = inwin & ((Tc // SEG) == (src // SEG))
    back = np.zeros_like(inwin)
    for k in range(4):
        back |= T[Tc][:, :, k] == src
    back &= inwin
    inbox = inwin & ~near & back
    atom = inwin & ~near & ~back
    print("window edges %d: in-segment %d, inbox %d, asymmetric far %d" % (inwin.sum(), near.sum(), inbox.sum(), atom.sum()))
    for name, m in (("asymmetric far", atom), ("all far", inwin & ~near)):
        s = src[m] // SEG
        d = Tc[m] // SEG
        pairs = np.unique(s * (1 << 20) + d)
        per_src = np.bincount(pairs >> 20)
        terms_src = np.bincount(s)
        act = terms_src > 0
        print("%s: %d terms from %d source segments; (source, destination) segment pairs %d" % (name, m.sum(), act.sum(), pairs.size))
        print("  distinct destinations per source segment: mean %.1f, median %d, p90 %d, max %d" %
              (per_src[per_src > 0].mean(), np.median(per_src[per_src > 0]), np.percentile(per_src[per_src > 0], 90), per_src.max()))
        print("  terms per source segment: mean %.1f, p90 %d, max %d" % (terms_src[act].mean(), np.percentile(terms_src[act], 90), terms_src.max()))
        per_dst = np.bincount(d)
        pd = per_dst[per_dst > 0]
        print("  destinations %d segments; terms per destination: mean %.1f, p90 %d, p99 %d, max %d" %
              (pd.size, pd.mean(), np.percentile(pd, 90), np.percentile(pd, 99), pd.max()))
        for g in (4, 16):
            pg = np.unique(s * (1 << 20) + d // g)
            print("  with destinations grouped %d segments: pairs %d" % (g, pg.size))
    # pass B's segment skipping: a segment has to be read if it holds a recent slot or a link of it reaches a "hot" unit
    # (a unit of `gran` slots that holds a recent slot)
    for gran in (2048, 1024, 256, 64, 16, 1):
        hot_unit = np.zeros(n // gran + 1, bool)
        hot_unit[np.nonzero(recent)[0] // gran] = True
        reach = valid & hot_unit[Tc // gran] & ((Tc // SEG) != (src // SEG))
        seg_reach = np.zeros(n // SEG + 1, bool)
        seg_reach[src[reach] // SEG] = True
        nseg_all = n // SEG + 1
        own = np.zeros(nseg_all, bool)            # the segment itself holds a recent slot
        own[np.nonzero(recent)[0] // SEG] = True
        # "own group hot": the unit that contains the segment, if units are larger than segments
        own_g = hot_unit[(np.arange(nseg_all) * SEG) // gran] if gran >= SEG else own
        nseg = n // SEG + 1
        print("units of %5d slots: hot units %6d (%.1f %%); segments to read: own hot %d, + reached %d = %d of %d (%.1f %%)" %
              (gran, hot_unit.sum(), 100.0 * hot_unit.mean(), own_g.sum(), (seg_reach & ~own_g).sum(), (seg_reach | own_g).sum(), nseg,
               100.0 * (seg_reach | own_g).mean()))
    # what the library's pass B actually skips (statistics off: the edge counters read every link)
    rec = wl.pipe.reconstruction
    rec.set_stats_enabled(False)
    for j in range(30, 35):
        wl.pipe.run_array(*wl.steps([wl.plan(first + j, 4 + j)]))
        api.StreamSynchronize(None)
        print("frame +%d: pass B skipped %d of %d segments" % (j, rec.debug_count_skipped_segments(), n // SEG + 1))
    # how far do far links reach?
    m = inwin & ~near
    dist = np.abs(Tc[m] - src[m])
    print("far link |target - source| slots: median %d, p90 %d" % (np.median(dist), np.percentile(dist, 90)))


if __name__ == "__main__":
    main()
