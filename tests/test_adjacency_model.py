"""NumPy model of the device Delaunay rule (adjacency_kernel): edge (u, v) exists iff the interval of circle
parameters left open by all other points is non-empty, tau_p = ((p-u).(p-v)) / cross(v-u, p-u).  Checked against
scipy/Qhull (the reference's library, func_vpr.py:1318-1321) on generic point sets, including mask-centroid-like
rational coordinates; the kernel itself is checked against Qhull on the GPU."""
import numpy as np
import pytest
from scipy.spatial import Delaunay


def delaunay_edges_interval_rule(P):
    S = len(P)
    A = np.eye(S, dtype=bool)
    for u in range(S):
        for v in range(u + 1, S):
            e = P[v] - P[u]
            others = np.array([p for p in range(S) if p != u and p != v])
            a, b = P[others] - P[u], P[others] - P[v]
            num = a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]
            den = e[0] * a[:, 1] - e[1] * a[:, 0]
            tau = np.divide(num, den, out=np.zeros_like(num), where=den != 0)
            tmin = tau[den > 0].min() if (den > 0).any() else np.inf
            tmax = tau[den < 0].max() if (den < 0).any() else -np.inf
            blocked = ((den == 0) & (num < 0)).any()
            if not blocked and tmax <= tmin:
                A[u, v] = A[v, u] = True
    return A


def qhull_adjacency(P):
    tri = Delaunay(P)
    S = len(P)
    A = np.eye(S, dtype=bool)
    indptr, indices = tri.vertex_neighbor_vertices
    for v in range(S):
        A[v, indices[indptr[v]:indptr[v + 1]]] = True
    return A


@pytest.mark.parametrize("S,seed", [(4, 1), (7, 2), (20, 3), (50, 4), (80, 5)])
def test_interval_rule_equals_qhull_on_generic_points(S, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    P = rng.random((S, 2)) * np.array([320.0, 240.0])
    assert np.array_equal(delaunay_edges_interval_rule(P), qhull_adjacency(P))


def test_interval_rule_on_mask_centroid_like_points():
    """Centroids are means of integer pixel coordinates (rationals with small denominators)."""
    rng = np.random.Generator(np.random.PCG64(9))
    for _ in range(5):
        pts = []
        for _s in range(30):
            h, w = rng.integers(3, 40, 2)
            y0, x0 = rng.integers(0, 200), rng.integers(0, 280)
            m = rng.random((h, w)) < 0.7
            if not m.any():
                m[0, 0] = True
            ys, xs = np.nonzero(m)
            pts.append([x0 + xs.mean(), y0 + ys.mean()])
        P = np.array(pts)
        assert np.array_equal(delaunay_edges_interval_rule(P), qhull_adjacency(P))


def test_order_power_matches_the_reference_rule():
    """A <- (A^order) > 0 (func_vpr.py:1343-1345) equals `order` rounds of neighbourhood OR-expansion."""
    rng = np.random.Generator(np.random.PCG64(11))
    P = rng.random((25, 2)) * 100
    A = delaunay_edges_interval_rule(P)
    for order in (1, 2, 3):
        ref = np.linalg.matrix_power(A.astype(np.float64), order) > 0
        cur = A.copy()
        for _ in range(order - 1):
            cur = (cur.astype(np.int64) @ A.astype(np.int64)) > 0
        assert np.array_equal(cur, ref)
