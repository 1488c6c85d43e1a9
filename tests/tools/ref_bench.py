"""The reference's OWN kernels (oracle/_ref/libsmx_ref.so: its two .cu files compiled by hipcc, see oracle/ref_build.py)
timed on this GPU at the bench's C2 state: the surfel map grown by the product (5 M slots) is handed to the reference's
kernels, which then integrate the same preprocessed frames.  Device time per Integrate (clears .. regulariser, the
reference's launch sequence with its two host round trips), no preprocessing.  A baseline for BASELINE.md, not part of
bench.py.      python tests/tools/ref_bench.py [frames]
"""
import sys, time
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch  # noqa
import bench
import oracle as orc
from oracle import ref_binding as ref
from surfelmeshing_amd import api, _lib
_lib.require_gpu()
wl = bench.Workload(api, 640, 480, 5_000_000, 5_500_000, 0x5EED0001, 0.0)
g_end, n = wl.grow(False)
first = g_end + 10
total = n_frames + 5
for j in range(-4, total + 5): wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(total)]
rec = wl.pipe.reconstruction
api.StreamSynchronize(None)
S = rec.debug_download_surfels()
merge0 = rec.surfels_size() - rec.surfel_count()
print('state: %d slots, %d merged' % (S.shape[1], merge0))
rr = ref.Recon(5_500_000, 640, 480, wl.fx, wl.fy, wl.cx, wl.cy)
rr.upload_surfels(S, merge0)
del S
params = orc.IntegrateParams.defaults()
# the same preprocessed images the product would integrate: raw frames -> the product's preprocessing kernels
from surfelmeshing_amd.pipeline import FramePipeline
pf = FramePipeline(640, 480, wl.fx, wl.fy, wl.cx, wl.cy, 1000, wl.pre)
need = set()
for j in range(total):
    need.add(plan[j][0]); need.update(plan[j][1])
colors = {}
for f in sorted(need):
    d, c = wl.pipe.download_frame(f)
    pf.upload(f, d, c)
    colors[f] = c
ms = []
for j in range(total):
    f, others, T, pose = plan[j]
    pf.preprocess(f, others, T)
    api.StreamSynchronize(None)
    depth, normals, radius = pf.depth_final.Download(), pf.normals.Download(), pf.radius.Download()
    rr.integrate(f, wl.pre.depth_scaling, np.ascontiguousarray(depth).copy(), normals, radius, colors[f], pose, params)
    ms.append(rr.last_integrate_ms())
    c = rr.counts()
    print('frame %d: %.3f ms  (slots %d, new %d)' % (f, ms[-1], c['surfels_size'], c['n_new']))
ms = np.array(ms[5:])
print('reference kernels on this GPU: %.3f ms per Integrate (median of %d), %.1f Integrate/s' % (np.median(ms), ms.size, 1e3 / np.median(ms)))
