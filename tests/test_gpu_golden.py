"""GPU path against the committed golden vectors -- the oracle is not executed here."""
import numpy as np
import pytest

from common import RESULT_ROWS, small_pre
from test_golden import G, run_golden_stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("native", [False, True])
def test_hip_path_reproduces_golden(smx, native):
    from surfelmeshing_amd.pipeline import FramePipeline, NativeFramePipeline
    fx, fy, cx, cy = [float(v) for v in G["intr"]]
    h, w = G["depth"].shape[1:]
    cls = NativeFramePipeline if native else FramePipeline
    pg = cls(w, h, fx, fy, cx, cy, 30000, small_pre(w))

    def check_stage(p):
        if native:
            d, n, r = p.download_work()
        else:
            d, n, r = p.depth_final.Download(), p.normals.Download(), p.radius.Download()
        assert np.array_equal(d, G["stage_final_depth"])
        assert np.array_equal(n.view(np.uint32), G["stage_normals"].view(np.uint32))

    counts = run_golden_stream(pg, check_stage)
    assert np.array_equal(np.array(counts), G["per_frame_counts"])       # counts bit-exact, every frame
    n = int(G["per_frame_counts"][-1][0])
    S = pg.reconstruction.debug_download_surfels(n)
    ref = G["surfels"]
    for r in RESULT_ROWS:
        assert np.array_equal(S[r].view(np.uint32), ref[r].view(np.uint32)), "row %d" % r
