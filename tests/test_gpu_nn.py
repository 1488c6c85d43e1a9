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


def test_c5_scale_parity_against_oracle_grid(smx):
    """Config C5 at 5 M points (the full 50 M run with true brute force is tests/tools/c5_pin.py, its
    record is kept under profiles/): the room-surface cloud, index cell = 1.5 x spacing, self-queries with the surfel
    radius and with twice the radius (max search-range factor, main.cc:392), K = 64 -- counts, indices and squared
    distances equal the oracle's grid search (itself pinned to brute force, tests/test_nn_oracle.py) bit for bit."""
    from surfelmeshing_amd.synth import room_surface_points
    pts, spacing = room_surface_points(5_000_000)
    n = len(pts)
    assert n > 4_900_000
    r = np.float32(1.5 * spacing)
    rng = np.random.default_rng(55)
    sel = rng.choice(n, 4000, replace=False)
    nn = smx.SurfelNeighborIndex()
    nn.Build(pts[:, 0], pts[:, 1], pts[:, 2], float(r))
    info = nn.stats()
    assert info["n_indexed"] == n and info["cell_size"] == r        # cell = r at this extent: no dense-grid cap
    assert max(info["dim"]) > 700 and info["n_bricks"] > 100_000   # (a dense grid of these cells would have 2 * 10^8 entries)
    for factor in (1.0, 2.0):
        r2 = np.full(len(sel), (factor * r) ** 2, np.float32)
        nn.set_stats_enabled(True)
        cnt, d2, idx = nn.FindNearestSurfelsWithinRadius(pts[sel], r2, 64)
        st = nn.stats()
        nn.set_stats_enabled(False)
        ocnt, od2, oidx = orc.nn_grid_batch(pts[:, 0], pts[:, 1], pts[:, 2], 0.05, pts[sel, 0], pts[sel, 1], pts[sel, 2], r2, 64)   # (the oracle's grid is dense: a coarser cell, same answers)
        assert np.array_equal(cnt, ocnt)
        for q in range(len(sel)):
            k = cnt[q]
            assert idx[q, 0] == sel[q] and d2[q, 0] == 0            # a self-query returns the surfel first (surfel_meshing.cc:433-465)
            assert np.array_equal(idx[q, :k], oidx[q, :k]) and np.array_equal(d2[q, :k].view(np.uint32), od2[q, :k].view(np.uint32))
        assert st["results"] == int(cnt.sum()) and st["distance_tests"] >= st["results"]
    nn.close()


def test_far_from_origin_small_radius(smx):
    """Coordinates of tens of metres with millimetre radii: the conservative cell range has to absorb the rounding of
    q -+ r itself (half an ulp of the coordinate is comparable to the radius margin there)."""
    rng = np.random.default_rng(9)
    base = np.array([70.0, -45.0, 120.0], np.float32)
    pts = (base + rng.uniform(-0.05, 0.05, (20000, 3))).astype(np.float32)
    sel = rng.choice(len(pts), 200, replace=False)
    _check(smx, pts, pts[sel], np.float32(0.003 ** 2), 64, 0.003)
    _check(smx, pts, pts[sel[:50]] + np.float32(0.0015), np.float32(0.004 ** 2), 16, 0.002)


def test_workspace_is_reused_and_queries_of_any_size_and_k(smx):
    """One handle, several builds and batches of different sizes and K (the list capacity is a template parameter:
    K <= 16, <= 32, <= 64); more than 64 queries per brick (several tiles per brick), K smaller than the matches."""
    rng = np.random.default_rng(21)
    nn = smx.SurfelNeighborIndex()
    for n, cell in ((3000, 0.2), (50000, 0.05), (700, 0.5), (120000, 0.04)):
        pts = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        nn.Build(pts[:, 0], pts[:, 1], pts[:, 2], cell)
        for nq, k, rad in ((1, 1, 0.3), (70, 5, 0.3), (500, 16, 0.15), (333, 17, 0.2), (900, 33, 0.25), (64, 64, 0.4)):
            q = rng.uniform(-1.1, 1.1, (nq, 3)).astype(np.float32)
            q[: nq // 2] = q[0]                                       # many queries in one brick
            r2 = (rng.uniform(0.5, 1.0, nq) * rad).astype(np.float32) ** 2
            cnt, d2, idx = nn.FindNearestSurfelsWithinRadius(q, r2, k)
            for j in range(0, nq, max(1, nq // 40)):
                c, od2, oidx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], q[j], float(r2[j]), k)
                assert cnt[j] == c and np.array_equal(idx[j, :c], oidx[:c]) and np.array_equal(d2[j, :c].view(np.uint32), od2[:c].view(np.uint32)), (n, nq, k, j)
        # the self queries walk tiles in KEY order that every build cuts anew (round 6): on a handle that has held larger and
        # smaller indexes they give the rows of the batch entry point over the same points
        r2s = np.full(n, (0.8 * cell) ** 2, np.float32)
        cs, ds, is_ = nn.FindNearestOfIndexedPoints(n, 64, radius_squared=r2s, factor=1.0)
        cb, db, ib = nn.FindNearestSurfelsWithinRadius(pts, r2s, 64)
        m = np.arange(64)[None, :] < cb[:, None]
        assert np.array_equal(cs, cb) and np.array_equal(is_[m], ib[m]) and np.array_equal(ds[m].view(np.uint32), db[m].view(np.uint32)), n
        assert cs.min() >= 1
    nn.close()


def test_self_queries_in_a_sparse_cloud(smx):
    """Self-queries over a SPARSE cloud (isolated points and small clusters, bricks far apart: most tiles hold one or two
    queries): rows identical to the wavefront-per-query kernel and to brute force."""
    rng = np.random.default_rng(77)
    n = 20_000
    pts = rng.uniform(-4, 4, (n, 3)).astype(np.float32)              # ~0.04 points per 5 cm cell: isolated points
    clusters = rng.uniform(-4, 4, (300, 3)).astype(np.float32)       # ... and small clusters, so that rows are not empty
    pts[:6000] = (clusters[rng.integers(0, 300, 6000)] + rng.normal(0, 0.03, (6000, 3))).astype(np.float32)
    r2 = (rng.uniform(0.02, 0.06, n).astype(np.float32)) ** 2
    nn = smx.SurfelNeighborIndex()
    nn.Build(pts[:, 0], pts[:, 1], pts[:, 2], 0.05)
    cs, ds, is_ = nn.FindNearestOfIndexedPoints(n, 32, radius_squared=r2, factor=1.0)
    nn.set_query_mode(0)
    c0, d0, i0 = nn.FindNearestOfIndexedPoints(n, 32, radius_squared=r2, factor=1.0)
    nn.set_query_mode(2)
    m = np.arange(32)[None, :] < cs[:, None]
    assert np.array_equal(c0, cs) and np.array_equal(i0[m], is_[m]) and np.array_equal(d0[m].view(np.uint32), ds[m].view(np.uint32))
    assert cs.min() >= 1 and (cs > 4).sum() > 1000 and (cs == 1).sum() > 1000      # every point finds itself; clusters and loners
    for j in rng.choice(n, 80, replace=False):
        c, od2, oidx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], pts[j], float(r2[j]), 32)
        assert cs[j] == c and np.array_equal(is_[j, :c], oidx[:c]) and np.array_equal(ds[j, :c].view(np.uint32), od2[:c].view(np.uint32))
    nn.close()


def test_self_queries_and_kernel_modes_agree(smx):
    """smx_nn_query_self (every indexed point asks for its own neighbourhood; no query keys / sort) and both query
    kernels of smx_nn_query_batch give the same rows; points that are not indexed get count 0; bricks with more than 64
    points (several sub-tiles) and a state mask are covered."""
    rng = np.random.default_rng(31)
    n = 60_000
    u = rng.uniform(-1, 1, n).astype(np.float32)
    v = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    pts = np.stack([u, v, (0.05 * np.sin(4 * u) * np.cos(3 * v)).astype(np.float32)], 1)
    pts[:3000, :2] *= 0.02                                              # a dense clump: > 64 points per brick
    nanrows = rng.choice(n, 500, replace=False)
    pts[nanrows, 0] = np.nan
    r2 = (rng.uniform(0.004, 0.012, n).astype(np.float32)) ** 2
    state = (rng.random(n) < 0.3).astype(np.uint8)
    nn = smx.SurfelNeighborIndex()
    nn.Build(pts[:, 0], pts[:, 1], pts[:, 2], 0.01)
    for st, mask in ((None, 0), (state, 1)):
        cs, ds, is_ = nn.FindNearestOfIndexedPoints(n, 64, radius_squared=r2, factor=1.0, state=st, skip_mask=mask)
        assert np.all(cs[nanrows] == 0)
        rows = {}
        for mode in (0, 1, 2):
            nn.set_query_mode(mode)
            rows[mode] = nn.FindNearestSurfelsWithinRadius(pts, r2, 64, state=st, skip_mask=mask)
        # the self-query entry point under the other tile kernel as well (the first call ran the default, mode 2:
        # one lane per query, with the dense clump's queries -- more than 32 matches -- redone by the tile kernel)
        m0 = np.arange(64)[None, :] < cs[:, None]
        nn.set_query_mode(0)
        c0, d0, i0 = nn.FindNearestOfIndexedPoints(n, 64, radius_squared=r2, factor=1.0, state=st, skip_mask=mask)
        assert np.array_equal(c0, cs) and np.array_equal(i0[m0], is_[m0]) and np.array_equal(d0[m0].view(np.uint32), ds[m0].view(np.uint32))
        nn.set_query_mode(2)
        assert (cs > 32).sum() > 100 and (cs < 32).sum() > 1000        # both paths of mode 2 were taken
        for mode in (0, 1, 2):
            cb, db, ib = rows[mode]
            assert np.all(cb[nanrows] == 0)                             # NaN query: empty ball
            assert np.array_equal(cb, cs), mode
            m = np.arange(64)[None, :] < cb[:, None]
            assert np.array_equal(ib[m], is_[m]) and np.array_equal(db[m].view(np.uint32), ds[m].view(np.uint32)), mode
        assert cs.max() == 64 and cs.min() == 0
        for j in rng.choice(n, 60, replace=False):                      # and against brute force
            if np.isnan(pts[j, 0]):
                continue
            c, od2, oidx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], pts[j], float(r2[j]), 64, state=st, skip_mask=mask)
            assert cs[j] == c and np.array_equal(is_[j, :c], oidx[:c]) and np.array_equal(ds[j, :c].view(np.uint32), od2[:c].view(np.uint32))
    cu, du, iu = nn.FindNearestOfIndexedPoints(n, 16, radius_squared=None, factor=float(np.float32(0.008) ** 2))
    cb, db, ib = nn.FindNearestSurfelsWithinRadius(pts, np.float32(0.008) ** 2, 16)
    m = np.arange(16)[None, :] < cb[:, None]
    assert np.array_equal(cu, cb) and np.array_equal(iu[m], ib[m])
    nn.close()
