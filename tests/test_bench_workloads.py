"""bench.py's synthetic workloads (CPU): the C5b refinement keeps the neuron's surface intact."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _edge_counts(faces):
    c = collections.Counter()
    for a, b, cc in faces:
        for e in ((a, b), (b, cc), (cc, a)):
            c[(min(e), max(e))] += 1
    return collections.Counter(c.values())


def _area(v, f):
    A, B, C = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    return 0.5 * np.linalg.norm(np.cross(B - A, C - A), axis=1).sum()


def _volume(v, f):
    A, B, C = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    return np.einsum("ij,ij->i", A, np.cross(B, C)).sum() / 6


def test_longest_edge_refinement_preserves_the_surface():
    m = np.load(os.path.join(ROOT, "tests", "golden", "neuron_mesh.npz"))
    v0, f0 = m["vertices"].astype(np.float64), m["faces"].astype(np.int64)
    v, f = bench.split_longest_edges(v0, f0, 33000)
    assert len(f) == 33000 and f.max() < len(v)
    # same manifold structure: every edge still has the same number of adjacent triangles
    assert set(_edge_counts(f)) == set(_edge_counts(f0))
    np.testing.assert_allclose(_area(v, f), _area(v0, f0), rtol=1e-9)
    np.testing.assert_allclose(_volume(v, f), _volume(v0, f0), rtol=1e-9)  # winding kept
    longest = lambda vv, ff: max(np.linalg.norm(vv[ff[:, i]] - vv[ff[:, (i + 1) % 3]], axis=1).max() for i in range(3))  # noqa: E731
    assert longest(v, f) < longest(v0, f0)


def test_fib_dirs_are_unit_vectors_and_packed_cylinders_have_4096_triangles():
    d = bench.fib_dirs(32)
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-12)
    v, f = bench.packed_cylinders()
    assert f.shape == (4096, 3) and f.max() < len(v)
