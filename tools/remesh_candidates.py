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
t = time.perf_counter()
nn = api.SurfelNeighborIndex()
r_med = float(np.sqrt(np.median(mirror.surfel_radius_squared_buffer[:N][alive])))
# merged slots are parked far away so that they are never returned
x = np.where(alive, mirror.surfel_x_buffer[:N], 1e6).astype(np.float32)
nn.Build(x, mirror.surfel_y_buffer[:N], mirror.surfel_z_buffer[:N], cell_size=2.0 * r_med)
api.StreamSynchronize(None)
t_build = time.perf_counter() - t
q = np.stack([mirror.surfel_x_buffer[idx], mirror.surfel_y_buffer[idx], mirror.surfel_z_buffer[idx]], axis=1)
t = time.perf_counter()
cnt, d2, ids = nn.FindNearestSurfelsWithinRadius(q, mirror.surfel_radius_squared_buffer[idx], 64)
api.StreamSynchronize(None)
t_query = time.perf_counter() - t
print('index build over %d surfels: %.1f ms (host rows, upload included); %d queries, K = 64, r^2 = the surfel\'s own: %.1f ms '
      '(host arrays in and out), %.1f neighbours on average, every query returns itself first: %s' % (
          N, t_build * 1e3, idx.size, t_query * 1e3, cnt.mean(), bool(np.all(ids[:, 0] == idx))))
