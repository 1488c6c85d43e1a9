#!/usr/bin/env python
"""Generates the committed golden vectors of tests/golden/.

They are OUTPUTS OF THE CPU ORACLE (oracle/), pinned at the commit that created them -- the reference itself
ships no golden vectors for this path and cannot be built here.  They serve two purposes:
 * regression pin of the oracle (tests/test_golden.py, CPU);
 * a GPU parity check that does not execute the oracle at all (tests/test_gpu_golden.py).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as orc  # noqa: E402
from common import small_pre, small_stream  # noqa: E402
from oracle_pipeline import OraclePipeline  # noqa: E402

W, H = 96, 72
FRAMES = list(range(4, 16))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    s = small_stream(W, H, obstacle_until=9)
    pre = small_pre(W)
    po = OraclePipeline(W, H, s.fx, s.fy, s.cx, s.cy, 30000, pre)
    depth, color = {}, {}
    for f in range(0, FRAMES[-1] + 5):
        depth[f], color[f] = s.frame(f)
        po.upload(f, depth[f], color[f])
    per_frame = []
    stages = None
    for f in FRAMES:
        po.process(f, s.outlier_frames(f), s.others_TR_reference(f), s.pose(f))
        if f == FRAMES[3]:
            stages = {k: v.copy() for k, v in po.stages.items()}
            stages["normals"] = po.normals.copy()
            stages["radius"] = po.radius.copy()
            stages["final_depth"] = po.depth_final.copy()
        per_frame.append((po.recon.surfels_size, po.recon.merge_count))
    n = po.recon.surfels_size
    S = po.recon.surfels()[:, :n].copy()
    for r in orc.SCRATCH_ROWS:
        S[r] = 0
    np.savez_compressed(
        os.path.join(HERE, "stream_96x72.npz"),
        depth=np.stack([depth[f] for f in sorted(depth)]), color=np.stack([color[f] for f in sorted(color)]),
        poses=np.stack([s.pose(f) for f in sorted(depth)]),
        others_T=np.stack([s.others_TR_reference(f) for f in FRAMES]),
        frames=np.array(FRAMES), per_frame_counts=np.array(per_frame, np.int64),
        stage_frame=np.array(FRAMES[3]),
        **{"stage_" + k: v for k, v in stages.items()},
        surfels=S, intr=np.array([s.fx, s.fy, s.cx, s.cy], np.float32),
        surfels_sha256=np.array(digest(S)))
    print("wrote stream_96x72.npz: %d surfels, %d merged, sha256 %s" % (n, po.recon.merge_count, digest(S)[:16]))


if __name__ == "__main__":
    main()
