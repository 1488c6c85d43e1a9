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
for j in range(-4, 40): wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(30)]
wl.pipe.run_array(*wl.steps(plan))
rec = wl.pipe.reconstruction
S = rec.debug_download_surfels()
n = S.shape[1]
stamp = S[18].view(np.uint32).astype(np.int64)
frame = first + 29
recent = stamp >= frame - 30
nb = S[19:23].view(np.uint32).astype(np.int64)
idx = np.arange(n)[None, :].repeat(4, 0)
valid = nb != 0xFFFFFFFF
tgt_recent = np.zeros_like(valid)
tgt_recent[valid] = recent[nb[valid]]
d = np.abs(nb - idx)[valid & tgt_recent]
print('recent', recent.sum(), 'edges into recent', d.size)
for h in (64, 256, 512, 1024, 2048, 4096, 8192, 65536, 1<<20):
    print('dist <= %7d: %.4f' % (h, (d <= h).mean()))
# symmetry of edges into recent targets: does the target list the source?
src = idx[valid & tgt_recent]
tgt = nb[valid & tgt_recent]
sym = np.zeros(src.size, bool)
for q in range(4):
    sym |= nb[q][tgt] == src
print('symmetric fraction of edges into recent targets: %.4f (asymmetric edges: %d)' % (sym.mean(), (~sym).sum()))
far = np.abs(tgt - src) > 1024
print('far edges: %d, symmetric among far: %.4f; near asym: %d' % (far.sum(), sym[far].mean(), (~sym & ~far).sum()))
da = np.abs(tgt - src)[~sym]
print('asymmetric edges: distance distribution')
for h in (64, 256, 512, 1024, 2048, 3072, 4096, 8192, 65536, 1<<20):
    print('  dist <= %7d: %.4f' % (h, (da <= h).mean()))
seg_same = ((tgt // 1024) == (src // 1024))[~sym]
print('asym same 1024-segment: %.4f' % seg_same.mean())
# in-degree from asymmetric edges
cnt = np.bincount(tgt[~sym], minlength=n)
print('targets with asym in-edges: %d, max in-degree %d, mean %.3f' % ((cnt > 0).sum(), cnt.max(), cnt[cnt > 0].mean()))
