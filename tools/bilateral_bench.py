"""Times the bilateral filter alone (640x480, the bench's parameters) -- python tools/bilateral_bench.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from surfelmeshing_amd import api, _lib
from surfelmeshing_amd.pipeline import PreprocessParams
_lib.require_gpu()
W, H = 640, 480
rng = np.random.default_rng(1)
d = (rng.uniform(1.0, 4.0, (H, W)) * 5000).astype(np.uint16)
d[rng.uniform(size=(H, W)) < 0.02] = 0
src = api.CUDABuffer(H, W, np.uint16); dst = api.CUDABuffer(H, W, np.uint16)
src.UploadAsync(None, d)
p = PreprocessParams(max_depth=10.0, depth_valid_region_radius=333.0)
def run(n):
    for _ in range(n):
        api.BilateralFilteringAndDepthCutoffCUDA(None, p.bilateral_filter_sigma_xy, p.bilateral_filter_sigma_depth_factor, 0,
                                                 p.bilateral_filter_radius_factor, p.max_depth_u16(), p.depth_valid_region_radius, src, dst)
    api.StreamSynchronize(None)
outs = []
for variant in (0, 1):
    _lib.check(_lib.load().smx_debug_set_bilateral_variant(variant))
    run(20)
    t = time.perf_counter(); run(500); dt = time.perf_counter() - t
    print('bilateral variant %d (%s): %.1f us per call' % (variant, 'two taps per instruction' if variant == 0 else 'one tap per instruction', dt / 500 * 1e6))
    outs.append(dst.Download())
print('results identical:', bool(np.array_equal(outs[0], outs[1])))
