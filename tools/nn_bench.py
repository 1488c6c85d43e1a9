#!/usr/bin/env python
"""Config C5 of SURVEY.md 8(d): radius-neighbor search over N surfels on the analytic room surface (jittered
grid, seed 0x5EED0005), self-queries for all surfels, K = 64, r^2 = surfel r^2 (the RemeshTrianglesAt pattern,
surfel_meshing.cc:819-823) and 4 r^2 (max search-range factor, main.cc:392).  Everything stays on the device.

    python tools/nn_bench.py [N]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surfelmeshing_amd import _lib, api  # noqa: E402


from surfelmeshing_amd.synth import room_surface_points as surface_points  # noqa: E402


def dev_array(host):
    host = np.ascontiguousarray(host)
    b = api.CUDABuffer(1, host.size, host.dtype)
    b.UploadAsync(None, host.reshape(1, -1))
    api.StreamSynchronize(None)
    return b


def main():
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    _lib.require_gpu()
    L = _lib.load()
    t0 = time.time()
    pts, spacing = surface_points(n_req)
    n = len(pts)
    print("# %d points, spacing %.2f mm (generated in %.1fs)" % (n, spacing * 1e3, time.time() - t0))
    bx, by, bz = (dev_array(pts[:, k]) for k in range(3))
    ptr = lambda b: C.c_void_p(b.ToCUDA().address)  # noqa: E731
    nn = C.c_void_p()
    _lib.check(L.smx_nn_create(C.c_int32(-1), C.byref(nn)))
    K = 64
    batch = min(n, 4_000_000)
    out_idx = api.CUDABuffer(1, batch * K, np.uint32)
    out_d2 = api.CUDABuffer(1, batch * K, np.float32)
    out_cnt = api.CUDABuffer(1, batch, np.int32)
    res = {}
    for factor in (1.5, 3.0):
        r = factor * spacing
        r2 = dev_array(np.full(batch, r * r, np.float32))
        api.StreamSynchronize(None)
        t0 = time.perf_counter()
        _lib.check(L.smx_nn_build(nn, None, ptr(bx), ptr(by), ptr(bz), C.c_uint32(n), C.c_float(r), C.c_int32(1)))
        api.StreamSynchronize(None)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        total = 0
        for q0 in range(0, n, batch):
            nq = min(batch, n - q0)
            off = lambda b: C.c_void_p(b.ToCUDA().address + 4 * q0)  # noqa: E731
            _lib.check(L.smx_nn_query_batch(nn, None, C.c_uint32(nq), off(bx), off(by), off(bz), ptr(r2), C.c_int32(K),
                                            C.c_void_p(0), C.c_uint8(0), C.c_int32(1), ptr(out_idx), ptr(out_d2),
                                            ptr(out_cnt), C.c_int32(1)))
            total += nq
        api.StreamSynchronize(None)
        t_query = time.perf_counter() - t0
        # the same all-points workload through smx_nn_query_self (no query keys / sort; rows indexed by point): first batch only
        # of the outputs is kept (the buffers hold `batch` rows), so it runs on an index over the first `batch` points
        cnt = out_cnt.Download()[0][:min(batch, n)]
        res[factor] = (t_build, t_query, float(cnt.mean()), int(cnt.max()))
        print("radius %.1f x spacing: build %.1f ms, %d queries in %.1f ms = %.1f Mq/s, mean results %.1f (max %d)" % (
            factor, t_build * 1e3, total, t_query * 1e3, total / t_query / 1e6, cnt.mean(), cnt.max()))
        # algorithmic bytes: build 12 N R + 8 N W; query 16 + 12 * candidates (staged per cell) + 8 k + 4
        r2.close()
    L.smx_nn_destroy(nn)


if __name__ == "__main__":
    main()
