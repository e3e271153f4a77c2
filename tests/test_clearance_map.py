"""The walk kernel's clearance map (include/dsim.h, dsim_mesh_clearance_map; CPU, no GPU needed).

A clear bit lets the kernel end a walker's time step without looking at a single triangle, so the
map has to be CONSERVATIVE: whenever some point of some triangle (of any periodic image - the
reference's mesh look-ups wrap around the voxel, simulations.py:654-679) lies within one step length
of some point of a fine cell, the cell's bit must be set.  Checked directly: points sampled ON the
triangles must have every cell whose box, inflated by the step length, contains them marked."""
import numpy as np
import pytest

import os
import sys

from conftest import ROOT, golden

sys.path.insert(0, ROOT)


def _bit(idx, bits, F, n_sv, ijk):
    lf = int(np.log2(F))
    words = max(1, F**3 // 32)
    sv = ((ijk[:, 0] >> lf) * n_sv[1] + (ijk[:, 1] >> lf)) * n_sv[2] + (ijk[:, 2] >> lf)
    blk = idx[sv]
    m = F - 1
    bit = ((((ijk[:, 0] & m) << lf) | (ijk[:, 1] & m)) << lf) | (ijk[:, 2] & m)
    w = bits[np.maximum(blk, 0) * words + (bit >> 5)]
    return (blk >= 0) & (((w >> (bit & 31).astype(np.uint32)) & 1) == 1)


def _surface_points(v, f, per_edge):
    ij = np.array([(i / per_edge, j / per_edge) for i in range(per_edge + 1) for j in range(per_edge + 1 - i)])
    A, B, C = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    p = A[:, None] + ij[None, :, 0, None] * (B - A)[:, None] + ij[None, :, 1, None] * (C - A)[:, None]
    return p.reshape(-1, 3)


@pytest.mark.parametrize("name,n_sv,step_l,periodic", [
    ("sphere_mesh.npz", [4, 5, 6], 4e-7, True),
    ("sphere_mesh.npz", [10, 10, 10], 1.5e-6, True),   # step longer than a fine cell: cells >= step / 2
    ("cylinder_mesh_closed.npz", [6, 6, 12], 2.4e-6, False),  # wall triangles spanning the voxel
    ("cylinder_mesh_closed.npz", [1, 1, 1], 1e-7, True),
    ("packed_cylinders", [50, 50, 50], 2.9e-7, True),  # BASELINE C4: every triangle spans the whole z axis
])
def test_clear_cells_are_farther_than_a_step_from_every_triangle(name, n_sv, step_l, periodic):
    from disimpy_amd import _lib, substrates

    if name == "packed_cylinders":
        import bench

        v, f = bench.packed_cylinders()
        sub = substrates.mesh(v, f, True, n_sv=np.array(n_sv), quiet=True)
    else:
        m = golden(name)
        sub = substrates.mesh(m["vertices"], m["faces"], periodic, padding=np.array([1e-6, 0.5e-6, 2e-6]),
                              n_sv=np.array(n_sv), quiet=True)
    got = _lib.mesh_clearance_map(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, step_l)
    assert got is not None
    idx, bits, F = got
    n_sv = np.asarray(sub.n_sv)
    nf = n_sv * F
    hf = sub.voxel_size / nf
    assert (hf >= 0.5 * step_l * (1 - 1e-12)).all() or F == 1
    q = _surface_points(sub.vertices, sub.faces, 40 if name == "packed_cylinders" else 12)
    q = q[np.random.RandomState(0).permutation(len(q))[:60000]]
    lo = np.ceil((q - step_l) / hf).astype(np.int64) - 1  # cells whose inflated box holds q
    hi = np.floor((q + step_l) / hf).astype(np.int64)
    span = int((hi - lo).max()) + 1
    checked = 0
    for a in range(span):
        for b in range(span):
            for c in range(span):
                ijk = lo + np.array([a, b, c])
                ok = (ijk <= hi).all(axis=1)
                if not ok.any():
                    continue
                w = np.mod(ijk[ok], nf)  # the map is periodic whatever the substrate says
                assert _bit(idx, bits, F, n_sv, w).all()
                checked += int(ok.sum())
    assert checked > len(q)
    # and it is not trivially "everything set": the padded voxel has open space
    g = np.stack(np.meshgrid(*[np.arange(k) for k in nf], indexing="ij"), axis=-1).reshape(-1, 3)
    frac_set = _bit(idx, bits, F, n_sv, g).mean()
    print(name, n_sv, "F", F, "set fraction", frac_set)
    assert frac_set < 0.95 or not periodic


def test_no_map_for_degenerate_input():
    from disimpy_amd import _lib

    m = golden("sphere_mesh.npz")
    v, f = m["vertices"], m["faces"]
    xs = np.linspace(0, 1e-5, 3)
    assert _lib.mesh_clearance_map(v, f, xs, xs, np.zeros(3), [2, 2, 2], 1e-7) is None  # zero-size voxel
    assert _lib.mesh_clearance_map(v, f, xs, xs, xs, [2, 2, 2], float("nan")) is None
