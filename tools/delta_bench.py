"""SURVEY 8f-1 at the bench's C2 state: TransferAllToCPU (32 B x N over PCIe) vs the changed-surfel delta, every 10
frames.      python tools/delta_bench.py"""
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
total = 60
for j in range(-4, total + 5): wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(total)]
rec = wl.pipe.reconstruction
rec.set_stats_enabled(False)
cpu = api.CUDASurfelsCPU(5_500_000)
def full():
    cpu.LockWriteBuffers()
    t = time.perf_counter()
    rec.TransferAllToCPU(None, 0, cpu)
    api.StreamSynchronize(None)
    dt = time.perf_counter() - t
    cpu.UnlockWriteBuffers(); cpu.WaitForLockAndSwapBuffers()
    return dt
full()
rec.SetDeltaTracking(None, True)
d = rec.TransferChangedToCPU(None, 0)          # the first delta is everything
reuse = d
print('slots %d; first delta %d' % (d.surfel_count, d.count))
for k in range(0, total, 10):
    t = time.perf_counter()
    wl.pipe.run_array(*wl.steps(plan[k:k + 10]))
    api.StreamSynchronize(None)
    t_frames = time.perf_counter() - t
    t = time.perf_counter()
    d = rec.TransferChangedToCPU(None, k, delta=reuse)
    t_delta = time.perf_counter() - t
    t_full = full()
    print('10 frames %.2f ms | delta: %7d slots (%.1f %% of %d), %5.1f MB, %.2f ms | full transfer: %5.1f MB, %.2f ms' % (
        t_frames * 1e3, d.count, 100.0 * d.count / d.surfel_count, d.surfel_count, d.count * 36 / 1e6, t_delta * 1e3,
        d.surfel_count * 32 / 1e6, t_full * 1e3))
