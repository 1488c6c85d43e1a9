"""Oracle regression pin against the committed golden vectors (tests/golden/, made by make_golden.py)."""
import os

import numpy as np

import oracle as orc
from common import small_pre
from oracle_pipeline import OraclePipeline

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stream_96x72.npz"))


def run_golden_stream(pipe, after_stage_frame=None):
    fx, fy, cx, cy = [float(v) for v in G["intr"]]
    for f in range(G["depth"].shape[0]):
        pipe.upload(f, G["depth"][f], G["color"][f])
    counts = []
    for k, f in enumerate(G["frames"]):
        f = int(f)
        others = [f - 1, f - 2, f - 3, f - 4, f + 1, f + 2, f + 3, f + 4]   # APP/main.cc:1046-1059
        pipe.process(f, others, G["others_T"][k], G["poses"][f])
        if after_stage_frame is not None and f == int(G["stage_frame"]):
            after_stage_frame(pipe)
        counts.append(pipe_counts(pipe))
    return counts


def pipe_counts(pipe):
    rec = getattr(pipe, "recon", None)
    if rec is not None:
        return rec.surfels_size, rec.merge_count
    r = pipe.reconstruction
    return r.surfels_size(), r.surfels_size() - r.surfel_count()


def test_oracle_reproduces_golden():
    fx, fy, cx, cy = [float(v) for v in G["intr"]]
    h, w = G["depth"].shape[1:]
    po = OraclePipeline(w, h, fx, fy, cx, cy, 30000, small_pre(w))

    def check_stages(p):
        for name in ("bilateral", "outlier", "erode", "normals_depth"):
            assert np.array_equal(p.stages[name], G["stage_" + name]), name
        assert np.array_equal(p.normals.view(np.uint32), G["stage_normals"].view(np.uint32))
        assert np.array_equal(p.depth_final, G["stage_final_depth"])

    counts = run_golden_stream(po, check_stages)
    assert np.array_equal(np.array(counts), G["per_frame_counts"])
    n = po.recon.surfels_size
    S = po.recon.surfels()[:, :n].copy()
    for r in orc.SCRATCH_ROWS:
        S[r] = 0
    assert np.array_equal(S.view(np.uint32), G["surfels"].view(np.uint32))
