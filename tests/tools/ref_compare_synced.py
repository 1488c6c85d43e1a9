"""Exploration: per frame, from the SAME surfel state: the reference's kernels vs the oracle with the reference's race
outcomes imposed."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch  # noqa
import oracle as orc
from oracle import ref_binding as ref
from common import small_stream, small_pre, RESULT_ROWS, FLOAT_ROWS, INT_ROWS
from oracle_pipeline import OraclePipeline

W, H = 160, 120
s = small_stream(W, H, obstacle_until=10)
pre = small_pre(W)
po = OraclePipeline(W, H, s.fx, s.fy, s.cx, s.cy, 60000, pre)
rr = ref.Recon(60000, W, H, s.fx, s.fy, s.cx, s.cy)
for g in range(0, 34):
    dd, cc = s.frame(g); po.upload(g, dd, cc)
params = orc.IntegrateParams.defaults()
names = {0:'X',1:'Y',2:'Z',3:'SX',4:'SY',5:'SZ',6:'conf',7:'r2',8:'NX',9:'NY',10:'NZ',17:'created',18:'stamp',19:'nb0',20:'nb1',21:'nb2',22:'nb3',24:'color'}
for g in range(4, 28):
    po.preprocess(g, s.outlier_frames(g), s.others_TR_reference(g))
    n0 = po.recon.surfels_size
    state = po.recon.surfels()[:, :n0].copy()
    rr.upload_surfels(state, po.recon.merge_count)
    depth_r = po.depth_final.copy()
    rr.integrate(g, pre.depth_scaling, depth_r, po.normals, po.radius, po.color[g], s.pose(g), params)
    sr = rr.scratch()
    sup_r, conf_r = np.ascontiguousarray(sr['supporting']), np.ascontiguousarray(sr['conflicting'])
    orc.set_race_overrides(sup_r, conf_r)
    po.integrate(g, s.pose(g))
    st = orc.race_override_stats()
    orc.set_race_overrides(None, None)
    cr = rr.counts()
    n = po.recon.surfels_size
    line = 'f%2d size %5d/%5d merges %3d/%3d new %4d/%4d ovr %s' % (g, n, cr['surfels_size'], po.recon.merge_count, cr['merge_count'], po.recon.stats()['n_new'], cr['n_new'], list(st.values()))
    so = po.recon.scratch()
    line += ' | sup!= %d cnt!= %d conf!= %d first!= %d depth!= %d' % ((so['supporting'] != sr['supporting']).sum(), (so['support_counts'] != sr['support_counts']).sum(), (so['conflicting'] != sr['conflicting']).sum(), (so['first_depth'].view(np.uint32) != sr['first_depth'].view(np.uint32)).sum(), (po.depth_final != depth_r).sum())
    if n == cr['surfels_size']:
        So, Sr = po.recon.surfels()[:, :n], rr.surfels(n)
        bad = []
        for r_ in INT_ROWS:
            k = (So[r_].view(np.uint32) != Sr[r_].view(np.uint32)).sum()
            if k: bad.append('%s:%d' % (names[r_], k))
        for r_ in FLOAT_ROWS:
            a, b = So[r_], Sr[r_]
            neq = (a.view(np.uint32) != b.view(np.uint32))
            if neq.any():
                rel = np.abs(a - b)[neq] / np.maximum(np.abs(b[neq]), 1e-3)
                bad.append('%s:%d(max rel %.1e)' % (names[r_], neq.sum(), rel.max()))
        line += ' | rows ' + (' '.join(bad) if bad else 'ALL BIT-EQUAL')
    print(line)
