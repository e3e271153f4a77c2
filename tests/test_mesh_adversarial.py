"""CPU side of the adversarial traversal tests: the oracle's one-pass twin (dso_u_mesh_pass) is pinned to the
oracle's own mesh step, and the adversarial ray set really attacks what it claims to (tests/adversarial_mesh.py)."""
import numpy as np
import numpy.testing as npt
import pytest

import adversarial_mesh as adv
from oracle.oracle import MESH


@pytest.fixture(scope="module")
def lib():
    from disimpy_amd import _lib

    return _lib


def _geom(o, mesh, **kw):
    return o.geom(MESH, mesh=mesh, **kw)


def test_one_pass_twin_equals_the_oracles_mesh_step(oracle_exact, lib):
    """random step + whole reflect loop + advance through dso_u_mesh_pass == one time step of dso_run"""
    o = oracle_exact
    mesh = adv.build_mesh(lib)
    n, seed = 4000, 7
    rng = np.random.RandomState(1)
    pos = np.ascontiguousarray(rng.uniform(0, mesh["L"], size=(n, 3)))
    step_l = 1.7 * adv.A
    for perm_prob in (0.0, 0.4):
        g = _geom(o, mesh, perm_prob=perm_prob, max_iter=50)
        st = o.rng_states(n, seed)
        p_ref = pos.copy()
        st_ref = st.copy()
        _, exc, _ = o.run(g, np.zeros((1, 1, 3)), 1e-3, step_l, p_ref, st_ref)
        st2 = st.copy()
        steps = o.random_steps(st2)  # advances st2 like the kernel's random step does
        r0, _, _, st3, flags = o.mesh_pass(g, pos, steps, np.full(n, step_l), st2, n_pass=0)
        npt.assert_array_equal(r0, p_ref)
        npt.assert_array_equal(st3, st_ref)
        npt.assert_array_equal((flags & 2) != 0, exc.astype(bool))
        # pass by pass == all at once
        r, s, sl, stp = pos.copy(), steps.copy(), np.full(n, step_l), st2.copy()
        over = np.zeros(n, dtype=bool)
        for _ in range(60):
            r_n, s_n, sl_n, st_n, fl = o.mesh_pass(g, r, s, sl, stp, n_pass=1)
            todo = ~over
            r[todo], s[todo], sl[todo], stp[todo] = r_n[todo], s_n[todo], sl_n[todo], st_n[todo]
            over |= (fl & 1) != 0
            if over.all():
                break
        assert over.all()
        r_end, _, _, _, _ = o.mesh_pass(g, r, s, sl, stp, n_pass=0)  # loop already over: only the advance is left
        npt.assert_array_equal(r_end, r0)
        npt.assert_array_equal(stp, st3)


GRIDS = ((8, 8, 8), (4, 16, 32), (2, 2, 2), (3, 5, 7))


def test_the_adversarial_rays_attack_ties_boundaries_and_images(oracle_exact, lib):
    o = oracle_exact
    n_vertex_ties = n_edge_ties = n_image_ties = n_big = n_exact = 0
    for n_sv in GRIDS:
        mesh = adv.build_mesh(lib, n_sv=n_sv)
        g = _geom(o, mesh)
        r0, step, step_l, tag = adv.rays(mesh, n_random=2000)
        ties, lists, subs = o.mesh_pass_stats(g, r0, step, step_l)
        assert len(step_l) > 30000
        n_vertex_ties += (ties[tag == 1] >= 3).sum()
        n_edge_ties += (ties[(tag == 2) | (tag == 3) | (tag == 4)] == 2).sum()
        n_image_ties += (ties[(tag >= 11) & (tag < 20)] >= 2).sum()
        n_big = max(n_big, ties.max())
        # rays inside a boundary plane visit nothing (on the 8 x 8 x 8 grid every lattice plane is a boundary)
        if n_sv == (8, 8, 8):
            assert (subs[tag == 21] == 0).all()
        # hits at exactly d == step_l: found with step_l = d, not with one ulp less (where no other sheet is in the way;
        # a vertex IN a boundary plane is not listed in the subvoxel above it - the reference walks through)
        idx = np.flatnonzero(tag == 20).reshape(len(mesh["sheets"]), 3, -1)
        n_exact += ((ties[idx[:, 0]] >= 1) & (ties[idx[:, 1]] == 0) & (ties[idx[:, 2]] >= 1)).sum()
        if n_sv == (4, 16, 32):
            # triangles listed in many subvoxels: a pass looks at the same triangle through several lists
            per_tri = np.bincount(mesh["triangle_indices"], minlength=len(mesh["faces"]))
            assert per_tri.max() >= 6 and lists.max() > 50
    # exact ties between different triangles (different normals): thousands, up to 6-8 triangles around a vertex, also in
    # far periodic images
    assert n_vertex_ties > 1000 and n_edge_ties > 3000 and n_image_ties > 3000 and n_big >= 6 and n_exact > 100, (
        n_vertex_ties, n_edge_ties, n_image_ties, n_big, n_exact)
