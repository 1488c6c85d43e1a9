"""Neighbor-search oracle pinned the way the reference pins its octree: exact equality with brute force
(APP/test/test_octree.cc:369-495: 100 trials x 100 points in a +-10 cube, K = 10, radius 3)."""
import numpy as np


def test_grid_search_equals_bruteforce_reference_protocol(orc):
    rng = np.random.default_rng(0)
    for trial in range(100):
        pts = rng.uniform(-10, 10, (100, 3)).astype(np.float32)
        q = rng.uniform(-10, 10, 3).astype(np.float32)
        n, d2, idx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], q, 9.0, 10)
        # independent numpy restatement of FindNearestSurfelsWithinRadiusBruteForce (test_octree.cc:116-143)
        dd = ((pts[:, 0] - q[0]) ** 2 + (pts[:, 1] - q[1]) ** 2) + (pts[:, 2] - q[2]) ** 2
        cand = np.nonzero(dd <= np.float32(9.0))[0]
        order = cand[np.lexsort((cand, dd[cand]))][:10]
        assert n == len(order)
        assert np.array_equal(idx[:n], order.astype(np.uint32))
        assert np.array_equal(d2[:n], dd[order])
        cnt, gd2, gidx = orc.nn_grid_batch(pts[:, 0], pts[:, 1], pts[:, 2], 3.0, q[0:1], q[1:2], q[2:3],
                                           np.array([9.0], np.float32), 10)
        assert cnt[0] == n and np.array_equal(gidx[0, :n], idx[:n]) and np.array_equal(gd2[0, :n], d2[:n])


def test_self_query_returns_self_first_and_state_filter(orc):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    state = (rng.random(500) < 0.3).astype(np.uint8)     # 1 = e.g. kFree / kCompleted, skipped (octree.cc:330-335)
    for i in (0, 17, 499):
        n, d2, idx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], pts[i], 0.25, 64)
        assert idx[0] == i and d2[0] == 0.0               # callers rely on it, surfel_meshing.cc:433-465
        assert np.all(np.diff(d2[:n]) >= 0)
        n2, d22, idx2 = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], pts[i], 0.25, 64, state=state, skip_mask=1)
        assert np.all(state[idx2[:n2]] == 0)
        assert n2 == int((state[idx[:n]] == 0).sum()) or n == 64


def test_empty_and_k_truncation(orc):
    pts = np.zeros((5, 3), np.float32)
    pts[:, 0] = np.arange(5)
    n, d2, idx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], np.array([10, 0, 0], np.float32), 1.0, 4)
    assert n == 0
    n, d2, idx = orc.nn_bruteforce(pts[:, 0], pts[:, 1], pts[:, 2], np.array([2, 0, 0], np.float32), 100.0, 3)
    assert n == 3 and list(idx[:3]) == [2, 1, 3]          # ties (d2 = 1) ordered by index
