"""Adversarial traversal tests on the GPU (``-m gpu``): ONE pass of the wavefront-cooperative traversal
(coop_iteration), its whole reflect loop, and the per-lane traversal (step_mesh) on caller-supplied rays, through
the test-only wrapper ``dsu_mesh_pass`` of libdsim_unit.so, against the oracle's twin ``dso_u_mesh_pass`` (one
iteration of simulations.py:924-997) - bit for bit: position, direction, remaining length, RNG state, flags.

The rays (tests/adversarial_mesh.py) go through shared edges and vertices of non-coplanar triangles on dyadic
coordinates (bit-equal distances from up to 8 different triangles with different normals: the reference's "first
visited wins", simulations.py:981), along and onto subvoxel boundary planes, end exactly on a triangle
(d == step_l), start in periodic images up to +-31 voxels and beyond, and meet triangles that are listed in dozens
of subvoxels; test_mesh_adversarial.py (CPU) counts the ties the set produces."""
import ctypes as C
import os

import numpy as np
import numpy.testing as npt
import pytest

import adversarial_mesh as adv
from oracle.oracle import MESH, STATE_DTYPE

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNIT_LIB = os.path.join(ROOT, "disimpy_amd", "csrc", "libdsim_unit.so")
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)


@pytest.fixture(scope="module")
def lib():
    from disimpy_amd import _lib

    assert _lib.device_count() > 0, "these tests need an AMD GPU"
    return _lib


@pytest.fixture(scope="module")
def unit():
    assert os.path.exists(UNIT_LIB), "build it with `make -C disimpy_amd/csrc libdsim_unit.so` (__graft_entry__.build() does)"
    u = C.CDLL(UNIT_LIB)
    u.dsu_last_error.restype = C.c_char_p
    u.dsu_mesh_pass.argtypes = [_f64p, C.c_longlong, _i64p, C.c_longlong, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p,
                                C.c_longlong, C.c_double, C.c_longlong, C.c_double, C.c_double, C.c_int, C.c_int,
                                C.c_longlong, _f64p, _f64p, _f64p, _u64p, _f64p, _f64p, _f64p, _u64p, _i32p]
    u.dsu_mesh_pass.restype = C.c_int
    return u


def gpu_pass(u, mesh, index_step_l, r0, step, step_l, states, mode, n_pass, max_iter=1000, eps=1e-13, perm_prob=0.0):
    n = len(step_l)
    r0 = np.ascontiguousarray(r0, dtype=np.float64)
    step = np.ascontiguousarray(step, dtype=np.float64)
    step_l = np.ascontiguousarray(step_l, dtype=np.float64)
    st = np.ascontiguousarray(np.stack([states["s0"], states["s1"]], 1), dtype=np.uint64)
    out_r, out_s, out_l = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    out_st, flags = np.zeros((n, 2), dtype=np.uint64), np.zeros(n, dtype=np.int32)
    v, f = mesh["vertices"], mesh["faces"]
    n_sv = np.ascontiguousarray(mesh["n_sv"], dtype=np.int64)
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    rc = u.dsu_mesh_pass(p(v, _f64p), len(v), p(f, _i64p), len(f), p(mesh["xs"], _f64p), p(mesh["ys"], _f64p), p(mesh["zs"], _f64p),
                         p(n_sv, _i64p), p(mesh["subvoxel_indices"], _i64p), p(mesh["triangle_indices"], _i64p),
                         len(mesh["triangle_indices"]), index_step_l, max_iter, eps, perm_prob, mode, n_pass, n, p(r0, _f64p),
                         p(step, _f64p), p(step_l, _f64p), p(st, _u64p), p(out_r, _f64p), p(out_s, _f64p), p(out_l, _f64p),
                         p(out_st, _u64p), p(flags, _i32p))
    assert rc == 0, u.dsu_last_error().decode()
    sto = np.zeros(n, dtype=STATE_DTYPE)
    sto["s0"], sto["s1"] = out_st[:, 0], out_st[:, 1]
    return out_r, out_s, out_l, sto, flags


def _same(a, b, what):
    """bit for bit; NaN == NaN"""
    a, b = np.asarray(a), np.asarray(b)
    bad = ~((a == b) | ((a != a) & (b != b)))
    assert not bad.any(), "%s: %d of %d differ, first at %s: %r vs %r" % (
        what, bad.sum(), bad.size, np.argwhere(bad)[0], a[tuple(np.argwhere(bad)[0])], b[tuple(np.argwhere(bad)[0])])


GRIDS = ((8, 8, 8), (4, 16, 32), (2, 2, 2), (3, 5, 7))


@pytest.mark.parametrize("n_sv", GRIDS)
@pytest.mark.parametrize("perm_prob", (0.0, 0.5))
def test_traversals_on_adversarial_rays(unit, lib, oracle_exact, n_sv, perm_prob):
    o = oracle_exact
    mesh = adv.build_mesh(lib, n_sv=n_sv)
    r0, step, step_l, tag = adv.rays(mesh, seed=3 + len(n_sv) + n_sv[0])
    n = len(step_l)
    states = o.rng_states(n, 99)
    g = o.geom(MESH, mesh=mesh, perm_prob=perm_prob, max_iter=1000, epsilon=1e-13)
    want1 = o.mesh_pass(g, r0, step, step_l, states, n_pass=1)
    want_all = o.mesh_pass(g, r0, step, step_l, states, n_pass=0)
    # the index is laid out for a step length: 0.05 lattice units -> finest micro-grid, 50 -> one cell per subvoxel
    for index_step_l in (0.05 * adv.A, 0.7 * adv.A, 50 * adv.A):
        tagstr = "n_sv=%s perm=%g index_step_l=%g" % (n_sv, perm_prob, index_step_l / adv.A)
        got = gpu_pass(unit, mesh, index_step_l, r0, step, step_l, states, 0, 1, perm_prob=perm_prob)
        for k, name in enumerate(("r0", "step", "step_l")):
            _same(got[k], want1[k], "one cooperative pass, %s, %s" % (name, tagstr))
        npt.assert_array_equal(got[3], want1[3])
        npt.assert_array_equal(got[4] & 1, want1[4] & 1)
        got = gpu_pass(unit, mesh, index_step_l, r0, step, step_l, states, 0, 0, perm_prob=perm_prob)
        for k, name in enumerate(("r0", "step", "step_l")):
            _same(got[k], want_all[k], "cooperative reflect loop, %s, %s" % (name, tagstr))
        npt.assert_array_equal(got[3], want_all[3])
        npt.assert_array_equal(got[4], want_all[4])
        got = gpu_pass(unit, mesh, index_step_l, r0, step, step_l, states, 1, 0, perm_prob=perm_prob)
        _same(got[0], want_all[0], "per-lane traversal, r0, %s" % tagstr)
        npt.assert_array_equal(got[3], want_all[3])
        npt.assert_array_equal(got[4], want_all[4])


def test_iteration_limit_on_adversarial_rays(unit, lib, oracle_exact):
    """max_iter = 2: the loop stops after two passes whatever is left; flags carry iter_exc"""
    o = oracle_exact
    mesh = adv.build_mesh(lib, n_sv=(3, 5, 7))
    r0, step, step_l, _ = adv.rays(mesh, seed=11, n_random=30000)
    states = o.rng_states(len(step_l), 5)
    g = o.geom(MESH, mesh=mesh, perm_prob=0.0, max_iter=2, epsilon=1e-13)
    want = o.mesh_pass(g, r0, step, step_l * 3, states, n_pass=0)
    assert (want[4] & 2).sum() > 100
    for mode in (0, 1):
        got = gpu_pass(unit, mesh, 0.3 * adv.A, r0, step, step_l * 3, states, mode, 0, max_iter=2)
        _same(got[0], want[0], "max_iter, mode %d" % mode)
        npt.assert_array_equal(got[3], want[3])
        npt.assert_array_equal(got[4], want[4])
