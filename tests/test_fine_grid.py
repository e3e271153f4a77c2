"""The fine grid of the mesh index (built on the host by dsim_set_mesh, handed out by dsim_mesh_fine_grid)
checked on the CPU against the reference's subvoxel lists: structure, and the completeness the exactness argument of
the traversal rests on - a triangle listed in subvoxel s is found, through the cells of s, at every one of its points
(border cells of a subvoxel extend to infinity outwards, as the traversal's clamped cell range assumes)."""
import numpy as np
import pytest

from conftest import golden


@pytest.fixture(scope="module")
def lib():
    from disimpy_amd import _lib

    return _lib


def _substrates():
    from disimpy_amd import substrates

    m = golden("sphere_mesh.npz")
    yield "sphere 6^3", substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, n_sv=np.array([6, 6, 6]), quiet=True), 3e-8
    m = golden("cylinder_mesh_closed.npz")
    yield "cylinder 10^3", substrates.mesh(m["vertices"], m["faces"], True, n_sv=np.array([10, 10, 10]), quiet=True), 1e-7
    n = golden("neuron_mesh.npz")
    idx = golden("neuron_index_50.npz")
    sub = substrates.mesh(n["vertices"], n["faces"].astype(np.int64), True, n_sv=np.array([2, 2, 2]), quiet=True)  # index replaced below
    sub.n_sv = np.array([50, 50, 50])
    sub.xs, sub.ys, sub.zs = (np.linspace(0, sub.voxel_size[k], 51) for k in range(3))
    ends = np.cumsum(idx["subvoxel_counts"].astype(np.int64))
    sub.triangle_indices = idx["triangle_indices"].astype(np.int64)
    sub.subvoxel_indices = np.stack([ends - idx["subvoxel_counts"], ends], axis=1).astype(np.int64)
    yield "neuron 50^3", sub, 2.9e-7


@pytest.mark.parametrize("name,sub,step_l", list(_substrates()), ids=lambda v: v if isinstance(v, str) else "")
def test_fine_grid_is_complete(lib, name, sub, step_l):
    fg = lib.mesh_fine_grid(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices, step_l)
    lk, k = fg["lk"], 1 << fg["lk"]
    # k > 1: along x and y a subvoxel has 2k half-cell units and a cell of two units at every unit but the last (the
    # cells overlap, dsim_mesh_host.h); the slots of a subvoxel are (2k)^2 k, z fastest
    ku = 2 * k if lk > 0 else 1
    slots = ku * ku * k
    cs, tri, pos, box = fg["cell_start"].astype(np.int64), fg["tri"], fg["pos"], fg["box"]
    svi, refs = np.asarray(sub.subvoxel_indices), np.asarray(sub.triangle_indices)
    n_sub = int(np.prod(sub.n_sv))
    # structure
    assert np.all(np.diff(cs) >= 0) and cs[0] == 0 and cs[-1] == len(tri)
    np.testing.assert_array_equal(tri, refs[pos])  # a record names the triangle of its list entry
    empty = svi[:, 1] <= svi[:, 0]
    if lk > 0:
        assert lk >= 1 and np.array_equal(fg["sv_base"] < 0, empty)
        base = fg["sv_base"].astype(np.int64)
        assert len(cs) - 1 == (~empty).sum() * slots
    else:
        base = np.arange(n_sub, dtype=np.int64)
        assert len(cs) - 1 == n_sub
    # every record of subvoxel s belongs to the list of s, and every list entry of s is in at least one cell of s
    rec_cell = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
    owner = np.full(len(cs) - 1, -1, dtype=np.int64)
    for s in np.flatnonzero(~empty):
        owner[base[s] : base[s] + slots] = s
    rec_sv = owner[rec_cell]
    assert np.all((pos >= svi[rec_sv, 0]) & (pos < svi[rec_sv, 1]))
    seen = np.zeros(len(refs), dtype=bool)
    seen[pos] = True
    assert seen.all(), "%d list entries are in no cell" % (~seen).sum()
    # completeness at sampled points: point p of triangle T, T listed in subvoxel s -> the cell of s that p maps to
    # (clamped into s) lists T, and that record's box holds p when p lies inside s
    rng = np.random.RandomState(1)
    V, F = np.asarray(sub.vertices), np.asarray(sub.faces)
    h = np.array([sub.xs[1] - sub.xs[0], sub.ys[1] - sub.ys[0], sub.zs[1] - sub.zs[0]])
    ns = np.asarray(sub.n_sv)
    cell_sets = {}
    checked = inside = 0
    take = rng.choice(len(refs), size=min(len(refs), 6000), replace=False)
    sv_of_ref = np.zeros(len(refs), dtype=np.int64)
    for s in np.flatnonzero(~empty):
        sv_of_ref[svi[s, 0] : svi[s, 1]] = s
    for i in take:
        s, t = sv_of_ref[i], refs[i]
        sx, sy, sz = s // (ns[1] * ns[2]), (s // ns[2]) % ns[1], s % ns[2]
        org = np.array([sub.xs[sx], sub.ys[sy], sub.zs[sz]])
        w = rng.dirichlet(np.ones(3), size=8)
        pts = w @ V[F[t]]
        per_axis = np.array([ku, ku, k])
        loc = np.clip(np.floor((pts - org) / h * per_axis).astype(np.int64), 0, per_axis - 1)
        for p, (ua, ub, c) in zip(pts, loc):
            # EVERY cell that holds the point's unit must list the triangle: the traversal may come through any of them
            cells_a = [q for q in (ua - 1, ua) if 0 <= q <= ku - 2] if lk > 0 else [0]
            cells_b = [q for q in (ub - 1, ub) if 0 <= q <= ku - 2] if lk > 0 else [0]
            for a in cells_a:
                for b in cells_b:
                    cell = base[s] + (a * ku + b) * k + c
                    key = int(cell)
                    if key not in cell_sets:
                        cell_sets[key] = {int(q): r for r, q in zip(range(cs[cell], cs[cell + 1]), pos[cs[cell] : cs[cell + 1]])}
                    assert int(i) in cell_sets[key], "%s: list entry %d (triangle %d, subvoxel %d) missing from cell %s of its subvoxel" % (name, i, t, s, (a, b, c))
                    checked += 1
                    if np.all(p >= org) and np.all(p <= org + h):
                        bx = box[cell_sets[key][int(i)]].astype(np.float64)
                        tol = 1e-6 * h.max()
                        assert np.all(p >= bx[:3] - tol) and np.all(p <= bx[3:] + tol), (name, i, p, bx)
                        inside += 1
    if lk > 0:  # the unused slots (a cell cannot start at the last unit of an axis) are empty
        n_rec = np.diff(cs).reshape(-1, ku, ku, k)
        assert n_rec[:, ku - 1].sum() == 0 and n_rec[:, :, ku - 1].sum() == 0
    assert checked > 1000 and inside > 100


@pytest.mark.parametrize("name,sub,step_l", list(_substrates()), ids=lambda v: v if isinstance(v, str) else "")
def test_packed_records_hold_their_float32_boxes(lib, name, sub, step_l):
    """The records the kernels read (MeshDev::mrec - 16 bytes: box and triangle - and MeshDev::planes, the plane by the
    triangle; dsim_mesh_fine_grid hands out both per record): the fixed-point box is the float32 box
    rounded OUTWARDS (never tighter, at most two quanta wider), so whatever overlapped the float32 box overlaps it."""
    fg = lib.mesh_fine_grid(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices, step_l)
    pk, box, qs = fg["packed"], fg["box"].astype(np.float64), np.float64(np.float32(fg["quantum"]))
    assert qs > 0 and abs(qs * max(sub.xs[-1], sub.ys[-1], sub.zs[-1]) - 32767.0) < 0.01
    lo = np.stack([pk[:, 0] & 0xFFFF, pk[:, 0] >> 16, pk[:, 1] & 0xFFFF], axis=1).astype(np.float64)
    nhi = np.stack([pk[:, 1] >> 16, pk[:, 2] & 0xFFFF, pk[:, 2] >> 16], axis=1).astype(np.float64)
    hi = 32767.0 - nhi
    assert lo.max() <= 32767 and nhi.max() <= 32767
    v_lo, v_hi = box[:, :3] * qs, box[:, 3:] * qs
    inside = (v_lo >= 0) & (v_hi <= 32767)
    assert np.all(lo <= np.maximum(v_lo, 0)) and np.all(hi >= np.minimum(v_hi, 32767))
    assert np.all((v_lo - lo)[inside] < 1.001) and np.all((hi - v_hi)[inside] < 1.001)
    np.testing.assert_array_equal(pk[:, 3].astype(np.int64), fg["tri"].astype(np.int64))
    # the plane: unit normal and offset of the record's triangle in float32
    V, F = np.asarray(sub.vertices), np.asarray(sub.faces)
    t = fg["tri"][:: max(1, len(pk) // 2000)]
    n = pk[:: max(1, len(pk) // 2000), 4:7].copy().view(np.float32).astype(np.float64)
    c = pk[:: max(1, len(pk) // 2000), 7].copy().view(np.float32).astype(np.float64)
    a, b, cc = V[F[t, 0]], V[F[t, 1]], V[F[t, 2]]
    want = np.cross(b - a, cc - a)
    want /= np.linalg.norm(want, axis=1)[:, None]
    assert np.abs(n - want).max() < 1e-6 and np.abs(np.einsum("ij,ij->i", n, a) - c).max() < 1e-6 * np.abs(V).max()
