"""Adversarial inputs for the mesh traversal (helper of test_mesh_adversarial.py / test_gpu_mesh_adversarial.py).

The reference keeps the FIRST triangle that attains the smallest positive distance (``d < min_d`` is strict,
simulations.py:981) while it walks the subvoxels of the segment's range in x, y, z order and every subvoxel's list
in list order.  Random rays never produce two different triangles with bit-equal ``d``, so the tie-break of a
traversal that visits the candidates in any other order is only exercised on purpose.  This module builds

* "crumpled sheets": height fields over a square lattice of pitch ``a = 2**-20`` m whose heights are small integer
  multiples of ``a``, every lattice square cut into two right triangles.  Neighbouring triangles are not coplanar
  (different normals: a wrong tie-break changes the reflected direction), all coordinates are dyadic, and for a ray
  along the sheet's axis the Moller-Trumbore arithmetic (simulations.py:234-275) is EXACT: det = a**2 is a power of
  two and every product is a small integer times a power of two.  A ray through a lattice vertex therefore returns
  bit-equal ``d`` from all (up to 8) triangles around the vertex, a ray through an edge from both its triangles;
* subvoxel grids whose boundaries are lattice planes (``linspace`` with a dyadic step is exact), so that shared
  edges and vertices lie IN subvoxel boundary planes and a triangle is listed in many subvoxels (a ray that runs inside
  a boundary plane has an EMPTY subvoxel range in the reference - ll == ul - and must see nothing), and grids that are
  not aligned with the lattice;
* ray families: through vertices / edge midpoints / diagonal points / interior points, starts on boundary planes,
  rays inside boundary planes, ``step_l`` equal to (and one ulp either side of) the hit distance, starts in
  periodic images up to +-31 voxels (and beyond: the serial fallback), oblique and random rays.
"""
import numpy as np

A = 2.0 ** -20  # lattice pitch (~0.95 um)


def crumpled_sheet(n_cells, axis, level, seed):
    """Height field over an (n_cells+1)^2 lattice, normal to ``axis``, mean position ``level`` lattice units.
    Returns (vertices in lattice units (float), faces)."""
    rng = np.random.RandomState(seed)
    h = rng.randint(0, 3, size=(n_cells + 1, n_cells + 1)).astype(float)
    # periodic seam: opposite borders get the same heights so that the sheet closes across the voxel
    h[-1, :] = h[0, :]
    h[:, -1] = h[:, 0]
    ii, jj = np.meshgrid(np.arange(n_cells + 1), np.arange(n_cells + 1), indexing="ij")
    uvw = np.stack([ii.ravel().astype(float), jj.ravel().astype(float), level + h.ravel()], axis=1)
    perm = {0: [2, 0, 1], 1: [1, 2, 0], 2: [0, 1, 2]}[axis]  # which of (u, v, w) supplies x, y, z
    vertices = uvw[:, perm]
    idx = lambda i, j: i * (n_cells + 1) + j  # noqa: E731
    faces = []
    for i in range(n_cells):
        for j in range(n_cells):
            faces.append([idx(i, j), idx(i + 1, j), idx(i, j + 1)])
            faces.append([idx(i + 1, j + 1), idx(i, j + 1), idx(i + 1, j)])
    return vertices, np.array(faces, dtype=np.int64)


def build_mesh(lib, n_cells=8, n_sv=(8, 8, 8), sheets=((2, 3, 11), (0, 4, 12), (1, 2, 13))):
    """Sheets (axis, level, seed) in a cubic voxel of n_cells lattice units.  ``lib`` = disimpy_amd._lib (its host-only
    mesh_subdivide builds the reference's index; membership is pinned by tests/test_host_api.py)."""
    vs, fs, off = [], [], 0
    for axis, level, seed in sheets:
        v, f = crumpled_sheet(n_cells, axis, level, seed)
        vs.append(v)
        fs.append(f + off)
        off += len(v)
    vertices = np.ascontiguousarray(np.vstack(vs) * A)
    faces = np.ascontiguousarray(np.vstack(fs))
    L = n_cells * A
    n_sv = np.array(n_sv, dtype=np.int64)
    xs, ys, zs = (np.linspace(0, L, int(k) + 1) for k in n_sv)
    for b, k in zip((xs, ys, zs), n_sv):  # power-of-two subdivisions: exact dyadic boundaries
        assert (int(k) & (int(k) - 1)) or np.array_equal(b, np.arange(int(k) + 1) * (L / int(k)))
    tri, sub = lib.mesh_subdivide(vertices, faces, xs, ys, zs, n_sv)
    return dict(vertices=vertices, faces=faces, xs=xs, ys=ys, zs=zs, n_sv=n_sv,
                subvoxel_indices=np.ascontiguousarray(sub, dtype=np.int64),
                triangle_indices=np.ascontiguousarray(tri, dtype=np.int64), L=L, n_cells=n_cells, sheets=sheets)


def _axis_frame(axis):
    perm = {0: [2, 0, 1], 1: [1, 2, 0], 2: [0, 1, 2]}[axis]
    return perm


def rays(mesh, seed=0, n_random=20000):
    """Returns (r0 (n,3), step (n,3), step_l (n,), tag (n,) int) - tag names the family of each ray."""
    rng = np.random.RandomState(seed)
    n_cells, L = mesh["n_cells"], mesh["L"]
    r0s, steps, sls, tags = [], [], [], []

    def add(r0, st, sl, tag):
        r0 = np.atleast_2d(np.asarray(r0, dtype=float))
        st = np.broadcast_to(np.asarray(st, dtype=float), r0.shape)
        sl = np.broadcast_to(np.asarray(sl, dtype=float), (len(r0),))
        r0s.append(r0)
        steps.append(np.array(st))
        sls.append(np.array(sl))
        tags.append(np.full(len(r0), tag))

    for axis, level, seed_s in mesh["sheets"]:
        perm = _axis_frame(axis)
        # lattice points in (u, v): vertices (integers), edge midpoints (half integers), points on the diagonal of a
        # square (u + v integer, quarter offsets), interior points
        g = np.arange(0, n_cells + 1)
        uu, vv = np.meshgrid(g, g, indexing="ij")
        pts = {
            1: np.stack([uu.ravel(), vv.ravel()], 1).astype(float),                     # vertices: up to 8-fold ties
            2: np.stack([uu[:-1].ravel() + 0.5, vv[:-1].ravel()], 1).astype(float),      # edges along u
            3: np.stack([uu[:, :-1].ravel(), vv[:, :-1].ravel() + 0.5], 1).astype(float), # edges along v
            4: np.stack([uu[:-1, :-1].ravel() + 0.25, vv[:-1, :-1].ravel() + 0.75], 1),   # the diagonal of a square
            5: np.stack([uu[:-1, :-1].ravel() + 0.25, vv[:-1, :-1].ravel() + 0.375], 1),  # interior of a triangle
        }
        for fam, uv in pts.items():
            for sign in (1.0, -1.0):
                for start in (level + 4.0, level - 2.0, level + 2.5, float(int(level) + 3)):  # incl. starts ON boundary planes
                    w0 = start if sign < 0 else 2 * level + 2 - start
                    uvw = np.column_stack([uv, np.full(len(uv), w0)])
                    r0 = uvw[:, perm] * A
                    d = np.zeros(3)
                    d[axis] = sign
                    for sl in (6.0 * A, 2.5 * A, 1.0 * A):
                        add(r0, d, sl, fam)
                    # far periodic images of the same start (exact: the voxel length is dyadic); +-31 is the last image the
                    # cooperative queues encode, beyond it the walker is done serially
                    for img in ((31, 0, 0), (0, -31, 0), (0, 0, 31), (-31, 31, -31), (5, -7, 3), (40, 0, 0), (0, 0, -33)):
                        add(r0 + np.array(img, dtype=float) * L, d, 6.0 * A, 10 + fam)
        # hits at exactly d == step_l (and one ulp either side): a ray down the axis from w0 onto the height at (u, v)
        rs = np.random.RandomState(seed_s)
        h = rs.randint(0, 3, size=(n_cells + 1, n_cells + 1)).astype(float)
        h[-1, :] = h[0, :]
        h[:, -1] = h[:, 0]
        i = rng.randint(0, n_cells + 1, 400)
        j = rng.randint(0, n_cells + 1, 400)
        w0 = level + 3.5
        dist = (w0 - (level + h[i, j])) * A
        uvw = np.column_stack([i.astype(float), j.astype(float), np.full(400, w0)])
        d = np.zeros(3)
        d[axis] = -1.0
        for sl in (dist, np.nextafter(dist, 0), np.nextafter(dist, 1)):
            add(uvw[:, perm] * A, d, sl, 20)
        # rays INSIDE a subvoxel boundary plane (one coordinate constant and equal to a boundary): the reference's
        # range is empty on that axis (ll == ul), nothing is visited
        k = rng.randint(0, n_cells + 1, 400).astype(float)
        other = rng.uniform(0, n_cells, size=(400, 2))
        uvw = np.column_stack([k, other[:, 0], np.full(400, level + 3.0)])
        add(uvw[:, perm] * A, d, 5.0 * A, 21)
        # oblique rays through lattice vertices (inexact arithmetic: near-ties)
        for dirv in ((1, 1, -2), (1, -1, -1), (-2, 1, -3), (0, 1, -1)):
            st = np.array(dirv, dtype=float)[perm]  # (u, v, w) components -> x, y, z
            st = st / np.linalg.norm(st)
            uv = pts[1]
            target = np.column_stack([uv, level + h[uv[:, 0].astype(int), uv[:, 1].astype(int)]])[:, perm] * A
            add(target - 2.0 * A * st, st, 5.0 * A, 22)
    # random rays everywhere (the non-adversarial baseline), a tenth of them in random periodic images
    r0 = rng.uniform(0, L, size=(n_random, 3))
    img = np.where(rng.uniform(size=(n_random, 1)) < 0.1, rng.randint(-31, 32, size=(n_random, 3)), 0)
    st = rng.normal(size=(n_random, 3))
    st /= np.linalg.norm(st, axis=1)[:, None]
    add(r0 + img * L, st, rng.uniform(0.2, 4.0, n_random) * A, 30)
    # degenerate: zero and negative remaining lengths, NaN coordinates
    add(r0[:64], st[:64], 0.0, 31)
    add(r0[:64], st[:64], -1.0 * A, 31)
    bad = r0[:64].copy()
    bad[::2, 0] = np.nan
    bad[1::2, 2] = np.inf
    add(bad, st[:64], 2.0 * A, 32)
    r0 = np.ascontiguousarray(np.vstack(r0s))
    step = np.ascontiguousarray(np.vstack(steps))
    step_l = np.ascontiguousarray(np.concatenate(sls))
    tag = np.concatenate(tags)
    return r0, step, step_l, tag
