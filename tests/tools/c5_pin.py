#!/usr/bin/env python
"""Config C5 at full size (50 M surfel positions on the room surface): the GPU index against TRUE brute
force (the reference's own pinning protocol, APP/test/test_octree.cc:369-495: exact equality of indices and float
squared distances) on a sample of self-queries, K = 64, with the surfel radius (1.5 x spacing) and twice that radius
(max search-range factor, APP/main.cc:392).  Checker-side tool: it runs the oracle's brute force (single thread, about
a minute for 2 x 200 queries over 50 M points); the record goes to gpurun_out/ and is kept under profiles/.

    python tests/tools/c5_pin.py [n_points] [n_queries] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
from surfelmeshing_amd import _lib, api  # noqa: E402
from surfelmeshing_amd.synth import room_surface_points  # noqa: E402


def main():
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "c5_pin.json")
    _lib.require_gpu()
    orc.build()
    pts, spacing = room_surface_points(n_req)
    n = len(pts)
    x, y, z = (np.ascontiguousarray(pts[:, k]) for k in range(3))
    r = np.float32(1.5 * spacing)
    rng = np.random.default_rng(0xC5)
    sel = rng.choice(n, nq, replace=False)
    nn = api.SurfelNeighborIndex()
    t0 = time.perf_counter()
    nn.Build(x, y, z, float(r))
    build_s = time.perf_counter() - t0
    info = nn.stats()
    rec = {"n_points": n, "spacing_mm": spacing * 1e3, "cell_mm": float(info["cell_size"]) * 1e3, "dim": info["dim"],
           "n_bricks": info["n_bricks"], "key_bits": info["key_bits"], "build_incl_upload_s": build_s, "queries": nq,
           "K": 64, "radii": []}
    ok_all = True
    for factor in (1.0, 2.0):
        r2 = np.full(nq, (factor * r) ** 2, np.float32)
        cnt, d2, idx = nn.FindNearestSurfelsWithinRadius(pts[sel], r2, 64)
        t0 = time.perf_counter()
        bad = 0
        for j in range(nq):
            c, od2, oidx = orc.nn_bruteforce(x, y, z, pts[sel[j]], float(r2[j]), 64)
            same = (cnt[j] == c and np.array_equal(idx[j, :c], oidx[:c])
                    and np.array_equal(d2[j, :c].view(np.uint32), od2[:c].view(np.uint32)))
            bad += 0 if same else 1
        rec["radii"].append({"radius_over_spacing": 1.5 * factor, "mean_results": float(cnt.mean()), "max_results": int(cnt.max()),
                             "queries_not_equal_to_brute_force": bad, "brute_force_s": time.perf_counter() - t0})
        ok_all = ok_all and bad == 0
    rec["all_equal_to_brute_force"] = ok_all
    nn.close()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
