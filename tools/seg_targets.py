"""Analysis: per 1024-slot segment, how many distinct target segments do its neighbour links reach, and how many
segments could pass B skip (no recent slot inside, no link into a segment with a recent slot)?"""
import sys, os
sys.argv=['bench.py','--steps','20','--warmup','5','--cpu-frames','0','--quiet']
sys.path.insert(0,'/root/repo')
import numpy as np
import bench
import torch
from surfelmeshing_amd import api, _lib
_lib.require_gpu()
wl = bench.Workload(api, 640, 480, 5_000_000, 5_500_000, 0x5EED0001, 0.0)
g_end, n = wl.grow(False)
first = g_end + 10
for j in range(-4, 60): wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(50)]
wl.pipe.run_array(*wl.steps(plan))
rec = wl.pipe.reconstruction
S = rec.debug_download_surfels()
n = S.shape[1]
stamp = S[18].view(np.uint32).astype(np.int64)
frame = first + 49
recent = stamp >= frame - 30
nb = S[19:23].view(np.uint32).astype(np.int64)
valid = nb != 0xFFFFFFFF
nseg = (n + 1023) // 1024
seg_of = np.arange(n) // 1024
seg_recent = np.zeros(nseg, bool)
np.logical_or.at(seg_recent, seg_of[recent], True)
print('segments', nseg, 'with recent slots', seg_recent.sum())
# distinct target segments per source segment
src_seg = np.repeat(seg_of[None, :], 4, 0)[valid]
tgt_seg = (nb[valid] // 1024)
pairs = np.unique(src_seg * (1 << 20) + tgt_seg)
ps, pt = pairs >> 20, pairs & ((1 << 20) - 1)
other = ps != pt
cnt = np.bincount(ps[other], minlength=nseg)
for q in (50, 75, 90, 95, 99, 100):
    print('distinct foreign target segments per segment, p%d: %d' % (q, np.percentile(cnt, q)))
# segments that link into a segment with recent slots (by segment granularity)
hits = np.zeros(nseg, bool)
np.logical_or.at(hits, ps, seg_recent[pt])
unskippable = seg_recent | hits
print('unskippable at segment granularity: %d of %d (%.1f %%)' % (unskippable.sum(), nseg, 100.0 * unskippable.mean()))
# exact: segments with a link to a recent SLOT
tgt_recent = np.zeros_like(valid)
tgt_recent[valid] = recent[nb[valid]]
hit_exact = np.zeros(nseg, bool)
np.logical_or.at(hit_exact, np.repeat(seg_of[None, :], 4, 0)[tgt_recent], True)
print('exact lower bound (own recent or link to a recent slot): %d (%.1f %%)' % ((seg_recent | hit_exact).sum(), 100.0 * (seg_recent | hit_exact).mean()))
for cap in (8, 16, 24, 32):
    print('cap %d: segments over the cap: %d' % (cap, (cnt > cap).sum()))
