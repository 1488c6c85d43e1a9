"""GPU radius-neighbor search vs the brute-force oracle (the reference's own pinning protocol,
APP/test/test_octree.cc:369-495: exact equality of indices and float squared distances)."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


def _check(smx, pts, queries, r2, k, cell, state=None, skip_mask=0):
    nn = smx.SurfelNeighborIndex()
    nn.Build(pts[:, 0], pts[:, 1], pts[:, 2], cell)
    cnt, d2, idx = nn.FindNearestSurfelsWithinRadius(queries, r2, k, state=state, skip_mask=skip_mask)
    r2a = np.broadcast_to(np.asarray(r2, np.float32), (len(queries),))
    for q in range(len(queries)):
        n, od2, oidx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], queries[q], float(r2a[q]), k,
                                         state=state, skip_mask=skip_mask)
        assert cnt[q] == n, (q, cnt[q], n)
        assert np.array_equal(idx[q, :n], oidx[:n]), q
        assert np.array_equal(d2[q, :n].view(np.uint32), od2[:n].view(np.uint32)), q
    nn.close()
    return cnt


def test_reference_protocol_random_cube(smx):
    rng = np.random.default_rng(0)
    for trial in range(20):
        pts = rng.uniform(-10, 10, (100, 3)).astype(np.float32)
        qs = rng.uniform(-10, 10, (5, 3)).astype(np.float32)
        _check(smx, pts, qs, 9.0, 10, 3.0)


def test_self_queries_k64_on_surface(smx):
    # the RemeshTrianglesAt pattern (surfel_meshing.cc:819-823): query = a surfel's own position, K = 64
    rng = np.random.default_rng(1)
    g = np.stack(np.meshgrid(np.arange(60), np.arange(50)), -1).reshape(-1, 2).astype(np.float32) * 0.01
    pts = np.concatenate([g + rng.normal(0, 0.002, g.shape).astype(np.float32),
                          (0.02 * np.sin(5 * g[:, :1])).astype(np.float32)], axis=1).astype(np.float32)
    sel = rng.choice(len(pts), 300, replace=False)
    r2 = (rng.uniform(0.015, 0.06, 300) ** 2).astype(np.float32)
    cnt = _check(smx, pts, pts[sel], r2, 64, 0.03)
    assert cnt.max() == 64 and cnt.min() >= 1          # some queries truncate at K, every query finds itself
    state = (rng.random(len(pts)) < 0.4).astype(np.uint8)
    _check(smx, pts, pts[sel[:100]], r2[:100], 64, 0.03, state=state, skip_mask=1)


def test_edge_cases(smx):
    pts = np.zeros((7, 3), np.float32)                  # all points identical: ties ordered by index
    q = np.zeros((2, 3), np.float32)
    q[1] = 5.0
    cnt = _check(smx, pts, q, 1.0, 4, 0.5)
    assert list(cnt) == [4, 0]
    one = np.array([[1.0, 2.0, 3.0]], np.float32)
    _check(smx, one, one, 0.0, 1, 1.0)                  # radius 0 still returns the point itself (d2 <= r2)
    rng = np.random.default_rng(2)
    pts = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    _check(smx, pts, pts[:50], 100.0, 64, 0.05)         # radius covering everything, tiny cells
    nn = smx.SurfelNeighborIndex()
    nn.Build(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), 1.0)
    cnt, _, _ = nn.FindNearestSurfelsWithinRadius(np.zeros((3, 3), np.float32), 1.0, 8)
    assert list(cnt) == [0, 0, 0]
    with pytest.raises(smx.SmxError):
        nn.FindNearestSurfelsWithinRadius(np.zeros((3, 3), np.float32), 1.0, 65)


def test_medium_cloud_against_oracle_grid(smx):
    """200k points on a curved surface, 3000 self-queries: the GPU grid search equals the oracle's grid search
    (itself equal to brute force, tests/test_nn_oracle.py) -- indices, squared distances and counts."""
    rng = np.random.default_rng(5)
    n = 200_000
    u = rng.uniform(-3, 3, n).astype(np.float32)
    v = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    w = (3.0 + 0.02 * np.sin(5 * u) * np.sin(5 * v)).astype(np.float32)
    pts = np.stack([u, v, w], 1)
    sel = rng.choice(n, 3000, replace=False)
    r2 = (rng.uniform(0.01, 0.05, 3000) ** 2).astype(np.float32)
    nn = smx.SurfelNeighborIndex()
    nn.Build(pts[:, 0], pts[:, 1], pts[:, 2], 0.05)
    cnt, d2, idx = nn.FindNearestSurfelsWithinRadius(pts[sel], r2, 64)
    ocnt, od2, oidx = orc.nn_grid_batch(pts[:, 0], pts[:, 1], pts[:, 2], 0.05, pts[sel, 0], pts[sel, 1], pts[sel, 2], r2, 64)
    assert np.array_equal(cnt, ocnt)
    for q in range(len(sel)):
        k = cnt[q]
        assert np.array_equal(idx[q, :k], oidx[q, :k]) and np.array_equal(d2[q, :k].view(np.uint32), od2[q, :k].view(np.uint32))
    assert cnt.max() == 64 and cnt.min() >= 1


def test_points_with_nan_coordinates_are_not_indexed(smx):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    pts[rng.choice(3000, 500, replace=False), rng.integers(0, 3, 500)] = np.nan
    q = rng.uniform(-1, 1, (40, 3)).astype(np.float32)
    _check(smx, pts, q, 0.09, 64, 0.3)
    nn = smx.SurfelNeighborIndex()
    allnan = np.full(10, np.nan, np.float32)
    nn.Build(allnan, allnan, allnan, 1.0)
    cnt, _, _ = nn.FindNearestSurfelsWithinRadius(np.zeros((2, 3), np.float32), 1.0, 8)
    assert list(cnt) == [0, 0]


def test_candidate_lists_from_the_device_map(smx):
    """index built from the device-resident map (merged slots left out) and the candidate lists of
    SurfelMeshing::TriangulateSurfel (APP/surfel_meshing.cc:359-360, 417-425) for a batch of slots, against brute
    force over the oracle's own map after the same frames: counts, indices and squared distances bit-equal."""
    from common import run_both, small_stream
    from test_gpu_parity import _pipes
    s = small_stream(obstacle_until=8, yaw_deg_per_frame=2.0)
    po, pg = _pipes(smx, s, 60000)
    run_both(po, pg, s, list(range(4, 18)), None)
    rec = pg.reconstruction
    t = po.recon.transfer_all()
    n = t["surfel_count"]
    merged = t["radius_squared"] < 0
    assert rec.surfels_size() == n and merged.sum() > 50 and n > 10000
    px, py, pz = (np.where(merged, np.float32(np.nan), t[a]) for a in "xyz")
    rng = np.random.default_rng(11)
    slots = np.concatenate([rng.choice(n, 600, replace=False), np.flatnonzero(merged)[:20],
                            [n, n + 5, 0xFFFFFFFF]]).astype(np.uint32)
    r_med = float(np.sqrt(np.median(t["radius_squared"][~merged])))
    nn = smx.SurfelNeighborIndex()
    nn.BuildFromReconstruction(rec, 2.0 * r_med)
    state = (rng.random(n) < 0.3).astype(np.uint8) * 2

    def expect(factor_sq, k, st, mask):
        out = []
        for i in slots:
            if i >= n or merged[i]:
                out.append((0, None, None))
                continue
            q = (t["x"][i], t["y"][i], t["z"][i])
            out.append(orc.nn_bruteforce(px, py, pz, q, float(np.float32(factor_sq) * t["radius_squared"][i]), k,
                                         state=st, skip_mask=mask))
        return out

    for factor_sq, k, st, mask in ((4.0, 64, None, 0), (1.0, 16, None, 0), (9.0, 64, state, 2)):
        cnt, d2, idx = nn.FindNeighborCandidates(rec, slots, factor_sq, k, state=st, skip_mask=mask)
        for j, (c, od2, oidx) in enumerate(expect(factor_sq, k, st, mask)):
            assert cnt[j] == c, (j, cnt[j], c)
            if c:
                assert np.array_equal(idx[j, :c], oidx[:c]), j
                assert np.array_equal(d2[j, :c].view(np.uint32), od2[:c].view(np.uint32)), j
        if st is None:
            live = cnt > 0
            assert np.array_equal(idx[live, 0], slots[live])      # every live surfel finds itself first
            assert not merged[idx[live][:, 0]].any()
    # a narrower ball with the same K is a prefix of the wide list (what lets the mesher reuse one batch)
    cw, dw, iw = nn.FindNeighborCandidates(rec, slots, 4.0, 64)
    cn, dn, inn = nn.FindNeighborCandidates(rec, slots, 1.5, 64)
    for j, i in enumerate(slots):
        if cw[j] == 0:
            continue
        lim = np.float32(1.5) * t["radius_squared"][i]
        c = int(np.sum(dw[j, :cw[j]] <= lim))
        assert cn[j] == c and np.array_equal(inn[j, :c], iw[j, :c])
    # empty map and empty batch
    from surfelmeshing_amd import api
    empty = api.CUDASurfelReconstruction(1000, pg.reconstruction.depth_camera)
    nn2 = smx.SurfelNeighborIndex()
    nn2.BuildFromReconstruction(empty, 0.1)
    cnt, _, _ = nn2.FindNeighborCandidates(empty, np.array([0, 1], np.uint32), 4.0, 8)
    assert list(cnt) == [0, 0]
    cnt, _, _ = nn.FindNeighborCandidates(rec, np.zeros(0, np.uint32), 4.0, 8)
    assert cnt.size == 0
    with pytest.raises(smx.SmxError):
        nn.FindNeighborCandidates(rec, slots, 4.0, 65)


def test_check_triangles_against_oracle(smx):
    """CheckRemeshing's per-triangle tests (APP/surfel_meshing.cc:590-650) on the device-resident map vs
    the oracle's restatement over the oracle's own map after the same frames; flags bit-equal.  Triangles: each
    surfel with two of its regulariser neighbours (small, mostly consistent), nearest-neighbour triples in both
    windings, random triples (long edges), merged and out-of-range vertices."""
    from common import run_both, small_stream
    from test_gpu_parity import _pipes
    s = small_stream(obstacle_until=8, yaw_deg_per_frame=2.0)
    po, pg = _pipes(smx, s, 60000)
    run_both(po, pg, s, list(range(4, 18)), None)
    rec = pg.reconstruction
    t = po.recon.transfer_all()
    n = t["surfel_count"]
    rows = po.recon.surfels()
    nb = rows[19:23, :n].view(np.uint32)
    ok = (nb[0] < n) & (nb[1] < n)
    i = np.flatnonzero(ok).astype(np.uint32)
    tri_nb = np.stack([i, nb[0, i], nb[1, i]], 1)
    rng = np.random.default_rng(3)
    tri_rand = rng.integers(0, n, (3000, 3)).astype(np.uint32)
    merged = np.flatnonzero(t["radius_squared"] < 0).astype(np.uint32)
    tri_merged = np.stack([merged, np.roll(merged, 1), rng.integers(0, n, merged.size).astype(np.uint32)], 1)
    tri_bad = np.array([[0, 1, n], [n + 7, 0, 1], [2, 0xFFFFFFFF, 3]], np.uint32)
    tris = np.concatenate([tri_nb, tri_nb[:, [0, 2, 1]], tri_rand, tri_merged, tri_bad]).astype(np.uint32)
    args = [t[a] for a in ("x", "y", "z", "radius_squared", "normal_x", "normal_y", "normal_z")]
    seen = set()
    for factor_sq in (16.0, 1.0, 400.0):
        want = orc.check_triangles(*args, tris, factor_sq)
        got = rec.CheckTrianglesForRemeshing(None, tris, factor_sq)
        assert np.array_equal(got, want), (factor_sq, np.flatnonzero(got != want)[:10])
        seen |= set(np.unique(want).tolist())
    # the cases all occur: clean, long edge, each winding bit, merged, out of range
    assert {0, 1, 16}.issubset(seen) and any(f & 14 for f in seen) and any((f & 16) and f != 16 for f in seen)
    assert rec.CheckTrianglesForRemeshing(None, np.zeros((0, 3), np.uint32), 16.0).size == 0
