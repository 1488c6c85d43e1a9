"""Ad-hoc oracle-vs-HIP comparison on a synthetic stream (development aid; tests/ hold the real checks)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle as orc
from oracle_pipeline import OraclePipeline
from surfelmeshing_amd import api
from surfelmeshing_amd.pipeline import FramePipeline, PreprocessParams
from surfelmeshing_amd.synth import SyntheticStream


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 120)
    nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    scan_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sc = w / 640.0
    s = SyntheticStream(width=w, height=h, fx=525.0 * sc, fy=525.0 * sc, cx=320.0 * sc, cy=240.0 * sc)
    pre = PreprocessParams(max_depth=10.0, depth_valid_region_radius=333.0 * sc)
    maxn = 40 * w * h // 10
    po = OraclePipeline(w, h, s.fx, s.fy, s.cx, s.cy, maxn, pre)
    pg = FramePipeline(w, h, s.fx, s.fy, s.cx, s.cy, maxn, pre)
    pg.reconstruction.set_scan_mode(scan_mode)
    first = 4
    frames = {f: s.frame(f) for f in range(0, first + nframes + 4)}
    for f, (d, c) in frames.items():
        po.upload(f, d, c)
        pg.upload(f, d, c)
    api.StreamSynchronize(None)
    ok = True
    for f in range(first, first + nframes):
        others, T, pose = s.outlier_frames(f), s.others_TR_reference(f), s.pose(f)
        t0 = time.time()
        po.preprocess(f, others, T)
        t1 = time.time()
        pg.preprocess(f, others, T)
        gd = pg.depth_final.Download()
        gn = pg.normals.Download()
        gr = pg.radius.Download()
        m = po.depth_final > 0
        bad_d = int((gd != po.depth_final).sum())
        bad_n = int((gn != po.normals).any(axis=2).sum())
        bad_r = int((gr[m] != po.radius[m]).sum())
        po.integrate(f, pose)
        t2 = time.time()
        pg.integrate(f, pose)
        st = pg.reconstruction.stats()
        so = po.recon.stats()
        n = po.recon.surfels_size
        ng = pg.reconstruction.surfels_size()
        line = "f=%d pre mism d/n/r=%d/%d/%d  N orc=%d hip=%d merged %d/%d" % (
            f, bad_d, bad_n, bad_r, n, ng, po.recon.merge_count, st["merge_count"])
        if n == ng:
            go = pg.reconstruction.debug_download_surfels(n)
            oo = po.recon.surfels()[:, :n]
            rows_bad = []
            for r in range(25):
                if r in orc.SCRATCH_ROWS:
                    continue
                nb = int((go[r].view(np.uint32) != oo[r].view(np.uint32)).sum())
                if nb:
                    rows_bad.append((r, nb))
            line += " rows_bad=%s" % rows_bad
            for name in ("supporting", "support_counts", "depth_sums_q", "conflicting", "first_depth", "new_flags", "new_indices"):
                a = pg.reconstruction.debug_download_scratch(name)
                b = po.recon.scratch()[name]
                nb = int((a.view(np.uint8) != b.view(np.uint8)).reshape(h, -1).any(axis=1).sum()) if a.dtype != b.dtype else int((a != b).sum())
                if nb:
                    line += " %s:%d" % (name, nb)
            gdep = pg.depth_final.Download()
            nb = int((gdep != po.depth_final).sum())
            if nb:
                line += " blended_depth:%d" % nb
            if rows_bad or bad_d or bad_n or bad_r:
                ok = False
        else:
            ok = False
        for k in so:
            if so[k] != st.get(k, so[k]):
                line += " stat[%s] %d!=%d" % (k, so[k], st[k])
        print(line, " (orc pre %.2fs int %.2fs)" % (t1 - t0, t2 - t1), flush=True)
    print("TIMINGS ms", pg.reconstruction.GetTimings())
    print("PARITY", "OK" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
