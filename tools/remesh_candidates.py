"""SURVEY 8f-2 composed from the pieces of this path: for the surfels of one changed-surfel delta, the K = 64 nearest
surfels within the mesher's search radius (SurfelMeshing::TriangulateSurfel starts from radius_squared,
APP/surfel_meshing.cc:330-352), batched on the GPU instead of one octree query per surfel on the mesher thread.
      python tools/remesh_candidates.py"""
import sys, time
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
import numpy as np
import torch  # noqa
import bench
from surfelmeshing_amd import api, _lib
_lib.require_gpu()
wl = bench.Workload(api, 640, 480, 5_000_000, 5_500_000, 0x5EED0001, 0.0)
g_end, n = wl.grow(False)
first = g_end + 10
for j in range(-4, 25): wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(20)]
rec = wl.pipe.reconstruction
rec.set_stats_enabled(False)
rec.SetDeltaTracking(None, True)
full = api.CUDASurfelsCPU(5_500_000)
full.LockWriteBuffers(); rec.TransferAllToCPU(None, 0, full); api.StreamSynchronize(None); full.UnlockWriteBuffers(); full.WaitForLockAndSwapBuffers()
mirror = full.read_buffers()
reuse = rec.TransferChangedToCPU(None, 0)           # (everything: discard)
wl.pipe.run_array(*wl.steps(plan[:10]))
api.StreamSynchronize(None)
d = rec.TransferChangedToCPU(None, 10, delta=reuse)
d.ApplyTo(mirror)
N = mirror.surfel_count
alive = mirror.surfel_radius_squared_buffer[:N] >= 0
print('map: %d slots; delta after 10 frames: %d surfels' % (N, d.count))
idx = d.surfel_index[:d.count]
idx = idx[alive[idx]]
r_med = float(np.sqrt(np.median(mirror.surfel_radius_squared_buffer[:N][alive])))
nn = api.SurfelNeighborIndex()
for rep in range(2):                                   # (second round: allocations warm)
    t = time.perf_counter()
    nn.BuildFromReconstruction(rec, 2.0 * r_med)       # smooth positions straight from the device records
    api.StreamSynchronize(None)
    t_build = time.perf_counter() - t
    t = time.perf_counter()
    cnt, d2, ids = nn.FindNeighborCandidates(rec, idx, 4.0, 64)   # ball = (2 r)^2, the widest TriangulateSurfel asks for
    api.StreamSynchronize(None)
    t_query = time.perf_counter() - t
print('index build over %d slots from the device map: %.1f ms; %d candidate lists, K = 64, r^2 = 4 x the surfel\'s own: '
      '%.1f ms (indices in, results out to host arrays), %.1f neighbours on average, %.1f %% of the lists full, '
      'every live query returns itself first: %s' % (
          N, t_build * 1e3, idx.size, t_query * 1e3, cnt.mean(), 100.0 * np.mean(cnt == 64),
          bool(np.all(ids[cnt > 0, 0] == idx[cnt > 0]))))
# cross-check against the host-row path (index from the mirrored rows, explicit positions and radii)
nn2 = api.SurfelNeighborIndex()
x = np.where(alive, mirror.surfel_x_buffer[:N], np.nan).astype(np.float32)
nn2.Build(x, mirror.surfel_y_buffer[:N], mirror.surfel_z_buffer[:N], cell_size=2.0 * r_med)
sel = idx[:: max(1, idx.size // 20000)]
q = np.stack([mirror.surfel_x_buffer[sel], mirror.surfel_y_buffer[sel], mirror.surfel_z_buffer[sel]], axis=1)
c2, dd2, i2 = nn2.FindNearestSurfelsWithinRadius(q, np.float32(4.0) * mirror.surfel_radius_squared_buffer[sel], 64)
c1, dd1, i1 = nn.FindNeighborCandidates(rec, sel, 4.0, 64)
print('device-map path == host-row path on %d sampled lists: %s' % (
    sel.size, bool(np.array_equal(c1, c2) and all(np.array_equal(i1[j, :c1[j]], i2[j, :c2[j]]) for j in range(sel.size)))))
