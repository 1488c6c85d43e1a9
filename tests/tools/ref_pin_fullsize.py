"""The reference pin of tests/test_gpu_reference_pin.py at the bench's size: the 5 M-slot map grown by the product
(640x480) is handed to the reference's own kernels (oracle/_ref) and to the CPU oracle; per frame, both start from the
same state, the oracle gets the reference run's race outcomes imposed, and everything is compared.
      python tests/tools/ref_pin_fullsize.py [frames]"""
import json, os, sys, time
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out_json = sys.argv[2] if len(sys.argv) > 2 else None   # e.g. gpurun_out/rNN_ref_pin_fullsize.json (kept under profiles/)
sys.argv = ['bench.py']
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa
import bench
import oracle as orc
from oracle import ref_binding as ref
from common import FLOAT_ROWS, INT_ROWS
from surfelmeshing_amd import api, _lib
from surfelmeshing_amd.pipeline import FramePipeline
_lib.require_gpu()
CAP = 5_500_000
wl = bench.Workload(api, 640, 480, 5_000_000, CAP, 0x5EED0001, 0.0)
g_end, n = wl.grow(False)
first = g_end + 10
for j in range(-4, n_frames + 5): wl.render(first + j, 4 + j)
plan = [wl.plan(first + j, 4 + j) for j in range(n_frames)]
rec = wl.pipe.reconstruction
api.StreamSynchronize(None)
S = rec.debug_download_surfels()
merge0 = rec.surfels_size() - rec.surfel_count()
print('state: %d slots, %d merged' % (S.shape[1], merge0))
po = orc.Recon(CAP, 640, 480, wl.fx, wl.fy, wl.cx, wl.cy)
po.surfels()[:, :S.shape[1]] = S
po.set_counts(S.shape[1], merge0)
rr = ref.Recon(CAP, 640, 480, wl.fx, wl.fy, wl.cx, wl.cy)
pf = FramePipeline(640, 480, wl.fx, wl.fy, wl.cx, wl.cy, 1000, wl.pre)
need = set()
for p in plan:
    need.add(p[0]); need.update(p[1])
colors = {}
# Conflicts at full size (the plain re-traversal integrates into a consistent map and sees none): a patch of
# the surface is pushed 25 % farther away in every frame of the pin -- a hole opens behind mapped surface, the surfels in
# front of it are seen through (free-space carving: conflict, confidence decrement, replacement).
# The hole is fixed in the WORLD (a sphere around the point the first pin frame sees at pixel (400, 200)): a disc fixed in
# the image moves over the surface with the camera, and the 9-frame outlier cull removes it.
yy, xx = np.mgrid[0:480, 0:640]
def world_points(f, d):
    R, t = bench.pose64(f - first + 4, 0.0)
    z = d.astype(np.float64) / wl.pre.depth_scaling
    pc = np.stack([(xx + 0.5 - wl.cx) / wl.fx * z, (yy + 0.5 - wl.cy) / wl.fy * z, z], -1)
    return pc @ R.T + t
d0, _ = wl.pipe.download_frame(plan[0][0])
centre = world_points(plan[0][0], d0)[200, 400]
for f in sorted(need):
    d, c = wl.pipe.download_frame(f)
    d = d.copy()
    hole = (np.linalg.norm(world_points(f, d) - centre, axis=-1) < 0.35) & (d > 0)
    d[hole] = np.minimum(65535, d[hole].astype(np.float64) * 1.25).astype(np.uint16)
    pf.upload(f, d, c); colors[f] = c
params = orc.IntegrateParams.defaults()
report = {"what": "reference kernels (oracle/_ref, compiled from /root/reference) vs the CPU oracle at the bench's size, "
                  "same state and preprocessed frame per frame, the reference run's race outcomes imposed on the oracle",
          "conflict_stream": "the surface inside a world-fixed sphere of 0.35 m pushed 25 % farther away in every frame (a hole behind mapped surface)",
          "width": 640, "height": 480, "slots_at_start": int(S.shape[1]), "merged_at_start": int(merge0), "frames": []}
for f, others, T, pose in plan:
    pf.preprocess(f, others, T)
    api.StreamSynchronize(None)
    depth, normals, radius = pf.depth_final.Download(), pf.normals.Download(), pf.radius.Download()
    n0 = po.surfels_size
    rr.upload_surfels(po.surfels()[:, :n0].copy(), po.merge_count)
    depth_r, depth_o = np.ascontiguousarray(depth).copy(), np.ascontiguousarray(depth).copy()
    rr.integrate(f, wl.pre.depth_scaling, depth_r, normals, radius, colors[f], pose, params)
    sr = rr.scratch()
    sup_r, conf_r = np.ascontiguousarray(sr['supporting']), np.ascontiguousarray(sr['conflicting'])
    orc.set_race_overrides(sup_r, conf_r)
    t = time.time()
    po.integrate(f, wl.pre.depth_scaling, depth_o, normals, radius, colors[f], pose, params)
    t = time.time() - t
    ovr = orc.race_override_stats()
    orc.set_race_overrides(None, None)
    cr = rr.counts()
    n = po.surfels_size
    so = po.scratch()
    line = 'frame %d: slots %d/%d merges %d/%d new %d/%d overrides %s | images differ: sup %d cnt %d conf %d first %d, blended depth %d' % (
        f, n, cr['surfels_size'], po.merge_count, cr['merge_count'], po.stats()['n_new'], cr['n_new'], list(ovr.values()),
        (so['supporting'] != sr['supporting']).sum(), (so['support_counts'] != sr['support_counts']).sum(),
        (so['conflicting'] != sr['conflicting']).sum(), (so['first_depth'].view(np.uint32) != sr['first_depth'].view(np.uint32)).sum(),
        (depth_o != depth_r).sum())
    So, Sr = po.surfels()[:, :n], rr.surfels(n)
    bad = []
    for r_ in INT_ROWS:
        k = int((So[r_].view(np.uint32) != Sr[r_].view(np.uint32)).sum())
        if k: bad.append('row%d:%d' % (r_, k))
    for r_ in FLOAT_ROWS:
        neq = So[r_].view(np.uint32) != Sr[r_].view(np.uint32)
        if neq.any(): bad.append('row%d:%d(max|d| %.1e)' % (r_, neq.sum(), np.abs(So[r_] - Sr[r_])[neq].max()))
    print(line + ' | rows: ' + (' '.join(bad) if bad else 'ALL BIT-EQUAL') + ' | oracle %.1fs' % t)
    smooth = [3, 4, 5]
    report["frames"].append({
        "frame": int(f), "slots": [int(n), int(cr['surfels_size'])], "merges": [int(po.merge_count), int(cr['merge_count'])],
        "new": [int(po.stats()['n_new']), int(cr['n_new'])], "race_outcomes": {k: int(v) for k, v in ovr.items()},
        "conflict_hits": int(po.stats()['n_conflict_hits']), "replaced": int(po.stats()['n_replaced']),
        "association_images_differing_pixels": {
            "supporting": int((so['supporting'] != sr['supporting']).sum()), "counts": int((so['support_counts'] != sr['support_counts']).sum()),
            "conflicting": int((so['conflicting'] != sr['conflicting']).sum()),
            "first_depth": int((so['first_depth'].view(np.uint32) != sr['first_depth'].view(np.uint32)).sum())},
        "blended_depth_differing_pixels": int((depth_o != depth_r).sum()),
        "integer_rows_differing": {str(r_): int((So[r_].view(np.uint32) != Sr[r_].view(np.uint32)).sum()) for r_ in INT_ROWS
                                   if (So[r_].view(np.uint32) != Sr[r_].view(np.uint32)).any()},
        "float_rows_differing": {str(r_): {"count": int((So[r_].view(np.uint32) != Sr[r_].view(np.uint32)).sum()),
                                           "max_abs_diff": float(np.abs(So[r_] - Sr[r_]).max())}
                                 for r_ in FLOAT_ROWS if (So[r_].view(np.uint32) != Sr[r_].view(np.uint32)).any()},
        "max_smooth_position_diff_m": float(max(np.abs(So[r_] - Sr[r_]).max() for r_ in smooth))})
if out_json:
    json.dump(report, open(out_json, 'w'), indent=1)
