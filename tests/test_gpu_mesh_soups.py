"""Random triangle soups (``-m gpu``): the mesh step (disimpy/simulations.py:884-1013) on substrates that look nothing
like the smooth closed surfaces of the other tests - open sheets of random triangles at every angle, slivers, triangles
far larger and far smaller than the subvoxels, walkers crossing them from both sides - over the regimes of the index the
library picks by itself: subvoxels as the cells (lk = 0) and 2 / 4 / 8 fine cells per subvoxel edge (overlapping along x
and y), boxes of a segment that span one unit or several subvoxels, periodic images, the appended wall of a non-periodic
voxel, permeable triangles.  Both traversals (per lane, wavefront-cooperative), every walker bit for bit against the
exact oracle: positions, phases, RNG states, iteration flags."""
import numpy as np
import pytest

from test_gpu_parity import MESH, _assert_identical, _gpu_run, _mesh_arrays, _oracle_run, _random_gradient, lib  # noqa: F401

pytestmark = pytest.mark.gpu

# seed, triangles, triangle size, voxel edge, n_sv, step length (all lengths in um), periodic, perm_prob, fine cells expected
CASES = [
    (1, 3000, 0.8, 12.0, (3, 3, 3), 0.15, True, 0.0, 8),
    (2, 3000, 0.5, 10.0, (2, 4, 3), 0.30, True, 0.3, 4),
    (3, 6000, 0.4, 8.0, (4, 4, 4), 0.08, True, 0.0, 8),
    (4, 1500, 1.5, 15.0, (5, 2, 7), 0.60, False, 0.0, 2),
    (5, 4000, 0.6, 9.0, (1, 1, 1), 0.10, True, 0.5, 8),
    (6, 800, 3.0, 20.0, (6, 6, 6), 2.5, True, 0.0, 1),  # a step nearly a subvoxel long: several subvoxels per segment
    (7, 5000, 0.3, 6.0, (3, 5, 2), 0.05, False, 0.2, 8),
    (8, 2000, 1.0, 10.0, (12, 12, 12), 0.2, True, 0.0, 2),
    (9, 2500, 0.7, 10.0, (5, 5, 5), 0.45, True, 0.1, 2),  # boxes wider than a unit: several columns of cells per pass
    (10, 20000, 0.5, 6.0, (4, 4, 4), 0.10, True, 0.0, 8),  # dense: a hit in most steps, long lists per cell
]


def _soup(seed, n_tri, size, voxel):
    rng = np.random.RandomState(seed)
    centre = rng.random_sample((n_tri, 1, 3)) * voxel
    shape = rng.standard_normal((n_tri, 3, 3)) * size * rng.choice([0.05, 0.3, 1.0], size=(n_tri, 1, 1))  # slivers among them
    shape[:, :, rng.randint(3)] *= 0.2  # flattened along one axis: many nearly parallel sheets
    v = np.clip(centre + shape, 0.0, voxel).reshape(-1, 3) * 1e-6
    v[0], v[1] = 0.0, voxel * 1e-6  # the voxel is the bounding box of the vertices
    return v, np.arange(3 * n_tri, dtype=np.int64).reshape(-1, 3)


@pytest.mark.parametrize("traversal", [1, 2])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "soup%d" % c[0])
def test_triangle_soup_bit_identical(lib, oracle_exact, case, traversal):  # noqa: F811
    from disimpy_amd import substrates

    seed, n_tri, size, voxel, n_sv, step_um, periodic, perm_prob, cells = case
    v, f = _soup(seed, n_tri, size, voxel)
    sub = substrates.mesh(v, f, periodic, n_sv=np.array(n_sv), quiet=True, perm_prob=perm_prob)
    step_l = step_um * 1e-6
    fg = lib.mesh_fine_grid(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices, step_l)
    assert 1 << fg["lk"] == cells  # the regime this case is here for
    n, T, dt = 4096 + 37, 24, 1e-4
    g = _random_gradient(2, T)
    pos0 = np.random.RandomState(100 + seed).random_sample((n, 3)) * sub.voxel_size
    kw = dict(max_iter=1000, epsilon=1e-13, perm_prob=perm_prob)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 11 + seed, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 11 + seed, kw, mesh=ma, tuning=dict(mesh_traversal=traversal, resort_steps=5),
                   chunks=[0, 9, T])
    _assert_identical(gpu, ref)
    moved = np.linalg.norm(gpu["pos"] - pos0, axis=1)
    assert moved.max() > step_l  # they did walk


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[7]], ids=lambda c: "soup%d" % c[0])
def test_two_level_look_up_without_the_dense_cell_table(lib, oracle_exact, case, monkeypatch):  # noqa: F811
    """The dense cell table (a slot block for every subvoxel) is an optimisation: when it is not built - it would pass
    DSIM_DENSE_CELLS_MB, or its memory is not to be had (then the allocation error is cleared and the shard goes on,
    ADVICE round 5) - the cooperative traversal finds a run's records through sv_base -> cell_start.  DSIM_DENSE_CELLS_MB=0
    takes that path on purpose: every walker bit for bit against the exact oracle, as with the table"""
    from disimpy_amd import substrates

    monkeypatch.setenv("DSIM_DENSE_CELLS_MB", "0")
    seed, n_tri, size, voxel, n_sv, step_um, periodic, perm_prob, cells = case
    assert cells > 1  # meshes with fine cells: without them cell_start is indexed by the subvoxel itself
    v, f = _soup(seed, n_tri, size, voxel)
    sub = substrates.mesh(v, f, periodic, n_sv=np.array(n_sv), quiet=True, perm_prob=perm_prob)
    step_l = step_um * 1e-6
    n, T, dt = 4096 + 37, 24, 1e-4
    g = _random_gradient(2, T)
    pos0 = np.random.RandomState(100 + seed).random_sample((n, 3)) * sub.voxel_size
    kw = dict(max_iter=1000, epsilon=1e-13, perm_prob=perm_prob)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 11 + seed, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 11 + seed, kw, mesh=ma, tuning=dict(mesh_traversal=2, resort_steps=5), chunks=[0, 9, T])
    _assert_identical(gpu, ref)


@pytest.mark.parametrize("intra", [True, False])
@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[4], CASES[6], CASES[7], CASES[9]], ids=lambda c: "soup%d" % c[0])
def test_fill_of_a_triangle_soup_identical_to_oracle(lib, oracle_exact, case, intra):  # noqa: F811
    """`_fill_mesh` (disimpy/simulations.py:421-579) where "inside" is nothing but the parity of the triangles a +x ray
    meets: open sheets, slivers whose (y, z) projection is a line, triangles that span many buckets of the fill's own
    index and many that share one - accepted points and their order against the oracle's loop, bit for bit."""
    from disimpy_amd import simulations, substrates

    seed, n_tri, size, voxel, n_sv, _, periodic, _, _ = case
    v, f = _soup(seed, n_tri, size, voxel)
    sub = substrates.mesh(v, f, periodic, n_sv=np.array(n_sv), quiet=True)
    n = 6000
    got = simulations._fill_mesh(n, sub, intra, 40 + seed)
    if periodic:
        vv, ff, tri, svi = sub.vertices, sub.faces, sub.triangle_indices, sub.subvoxel_indices
    else:
        vv, ff, tri, svi = simulations._strip_wall_triangles(sub)
    g = oracle_exact.geom(MESH, mesh=dict(vertices=vv, faces=ff, xs=sub.xs, ys=sub.ys, zs=sub.zs, subvoxel_indices=svi,
                                          triangle_indices=tri, n_sv=sub.n_sv))
    want = oracle_exact.fill_mesh(n, g, sub.voxel_size, intra, 40 + seed)
    np.testing.assert_array_equal(got, want)
