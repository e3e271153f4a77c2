"""Pins the CPU oracle (oracle/dsim_oracle.c) against the reference's own
golden vectors and known answers, before anything is compared with it.

Sources of the expected values:
* tests/golden/traj_free_seed123.npy = disimpy/tests/test_traj.txt (reference
  output of simulation(10 walkers, free, seed=123, traj=...))
* disimpy/tests/test_simulations.py:23-360 (device-function known answers)
"""
import ctypes as C

import numpy as np
import numpy.testing as npt
import pytest

from conftest import golden
from oracle.oracle import FREE, _f64p, _i64p

STEP_L = 9.84731927834662e-07  # sqrt(6 * 2e-9 * 8.0808e-05), constant step length of the fixture


def _p(a):
    return a.ctypes.data_as(_f64p)


@pytest.mark.parametrize("which", ["libm", "exact"])
def test_golden_trajectory(which, oracle_libm, oracle_exact):
    """10 walkers x 999 steps of free diffusion, seed 123, walker w = RNG state w."""
    o = oracle_libm if which == "libm" else oracle_exact
    traj = golden("traj_free_seed123.npy").reshape(1000, 10, 3)
    assert np.all(traj[0] == 0)
    st = o.rng_states(128, 123)
    pos = np.zeros((10, 3))
    _, _, tr = o.run(o.geom(FREE), np.zeros((1, 999, 3)), 1e-3, STEP_L, pos, st, want_traj=True)
    full = np.concatenate([np.zeros((1, 10, 3)), tr])
    err = np.linalg.norm((np.diff(traj, axis=0) - np.diff(full, axis=0)) / STEP_L, axis=2)
    # CUDA logf vs glibc/own logf differ by 1 ulp(float32) on a few % of draws
    assert err.max() < 6e-8
    assert np.median(err) < 1e-12
    assert (err < 1e-12).mean() > 0.5
    assert np.abs(full - traj).max() < 1e-12  # metres, after 999 steps
    npt.assert_allclose(np.linalg.norm(np.diff(full, axis=0), axis=2), STEP_L, rtol=1e-12)


def test_other_seeds_do_not_match(oracle_libm):
    traj = golden("traj_free_seed123.npy").reshape(1000, 10, 3)
    for seed in (0, 1, 12, 42):
        st = oracle_libm.rng_states(16, seed)
        pos = np.zeros((10, 3))
        oracle_libm.run(oracle_libm.geom(FREE), np.zeros((1, 1, 3)), 1e-3, STEP_L, pos, st)
        assert np.abs(pos - traj[1]).max() > 1e-8


def test_cudasim_log_mode_is_a_different_rounding(oracle_libm):
    """float64 log of the float32 uniform (NUMBA_ENABLE_CUDASIM) differs ~1e-8."""
    traj = golden("traj_free_seed123.npy").reshape(1000, 10, 3)
    try:
        oracle_libm.set_log_mode(True)
        st = oracle_libm.rng_states(16, 123)
        pos = np.zeros((10, 3))
        _, _, tr = oracle_libm.run(oracle_libm.geom(FREE), np.zeros((1, 999, 3)), 1e-3, STEP_L, pos, st,
                                   want_traj=True)
    finally:
        oracle_libm.set_log_mode(False)
    full = np.concatenate([np.zeros((1, 10, 3)), tr])
    err = np.linalg.norm((np.diff(traj, axis=0) - np.diff(full, axis=0)) / STEP_L, axis=2)
    assert 1e-9 < err.max() < 1e-7
    assert (err < 1e-12).mean() < 0.05


def test_rng_splitmix_and_jump(oracle_libm):
    """state[0] = splitmix64(seed) in both words; state[i] = jump(state[i-1])."""
    st = oracle_libm.rng_states(4, 123)
    z = (123 + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    z ^= z >> 31
    assert int(st["s0"][0]) == z and int(st["s1"][0]) == z

    # pure-Python jump of the published xoroshiro128+ (55, 14, 36)
    def nxt(s):
        s0, s1 = s
        s1 ^= s0
        m = 2**64 - 1
        return [(((s0 << 55) | (s0 >> 9)) & m) ^ s1 ^ ((s1 << 14) & m), ((s1 << 36) | (s1 >> 28)) & m]

    s = [z, z]
    a = c = 0
    for word in (0xBEAC0467EBA5FACB, 0xD86B048B86AA9922):
        for b in range(64):
            if word >> b & 1:
                a ^= s[0]
                c ^= s[1]
            s = nxt(s)
    assert (int(st["s0"][1]), int(st["s1"][1])) == (a, c)
    # subsequence_start advances state 0 by whole jumps
    st2 = oracle_libm.rng_states(2, 123, subsequence_start=2)
    assert st2[0] == st[2] and st2[1] == st[3]


def test_random_step_statistics(oracle_libm):
    """disimpy/tests/test_simulations.py:112-139."""
    n = 100000
    steps = [oracle_libm.random_steps(oracle_libm.rng_states(n, s)) for s in (1, 1, 12)]
    npt.assert_equal(steps[0], steps[1])
    assert np.all(steps[0] != steps[2])
    npt.assert_almost_equal(np.mean(np.sum(steps[1], axis=0) / n), 0, 3)
    npt.assert_almost_equal(np.linalg.norm(np.array(steps), axis=2), np.ones((3, n)))


def test_vector_helpers(oracle_libm):
    lib = oracle_libm.lib
    rng = np.random.RandomState(0)
    for _ in range(50):
        a, b = rng.random(3) - 0.5, rng.random(3) - 0.5
        npt.assert_almost_equal(lib.dso_u_dot(_p(a), _p(b)), np.dot(a, b))
        c = np.zeros(3)
        lib.dso_u_cross(_p(a), _p(b), _p(c))
        npt.assert_almost_equal(c, np.cross(a, b))
        v = a.copy()
        lib.dso_u_normalize(_p(v))
        npt.assert_almost_equal(v, a / np.linalg.norm(a))
        tri = rng.random((3, 3)) - 0.5
        n = np.zeros(3)
        lib.dso_u_triangle_normal(_p(tri), _p(n))
        want = np.cross(tri[0] - tri[1], tri[0] - tri[2])
        npt.assert_almost_equal(n, want / np.linalg.norm(want))
        R = rng.random((3, 3)) - 0.5
        v = a.copy()
        lib.dso_u_mat_mul(_p(R), _p(v))
        npt.assert_almost_equal(v, R @ a)


def test_line_quadric_known_answers(oracle_libm):
    """d = 1.1414213562373097 (test_simulations.py:181, :204, :227)."""
    lib = oracle_libm.lib
    s2 = np.array([1.0, 1.0]) / np.sqrt(2)
    npt.assert_almost_equal(lib.dso_u_line_circle(_p(np.array([-0.1, -0.1])), _p(s2), 1.0), 1.1414213562373097)
    r0 = np.array([-0.1, -0.1, 0.0])
    s3 = np.array([1.0, 1.0, 0.0]) / np.sqrt(2)
    npt.assert_almost_equal(lib.dso_u_line_sphere(_p(r0), _p(s3), 1.0), 1.1414213562373097)
    npt.assert_almost_equal(lib.dso_u_line_ellipsoid(_p(r0), _p(s3), _p(np.ones(3))), 1.1414213562373097)


def test_ray_triangle_known_answers(oracle_libm):
    """[1, -1, 10, nan, nan] (test_simulations.py:242-259)."""
    lib = oracle_libm.lib
    tri = np.array([[2.0, 0, 0], [0, 2.0, 0], [0.0, 0, 0]])
    r0s = np.array([[0.1, 0.1, 1.0]] * 4 + [[10.0, 10, 0]])
    steps = np.array([[0, 0, -1.0], [0, 0, 1], [0, 0, -0.1], [1.0, 1.0, 0], [0, 0, 1.0]])
    got = [lib.dso_u_ray_triangle(_p(tri), _p(r0s[i].copy()), _p(steps[i].copy())) for i in range(5)]
    npt.assert_almost_equal(got, [1, -1, 10, np.nan, np.nan])


def test_reflection_and_crossing_known_answers(oracle_libm):
    """test_simulations.py:278-295, :312-325, :346-359."""
    lib = oracle_libm.lib
    normal0 = np.array([0.0, 1.0, 1.0]) / np.sqrt(2)
    for eps in (0.0, 0.5):
        r0, step, normal = np.zeros(3), np.array([0.0, 0.0, 1.0]), normal0.copy()
        lib.dso_u_reflection(_p(r0), _p(step), 0.5, _p(normal), eps)
        npt.assert_almost_equal(step, [0.0, -1.0, 0.0])
        # the normal is flipped in place to point against the step
        npt.assert_almost_equal(normal, -normal0)
        npt.assert_almost_equal(r0, np.array([0, 0, 0.5]) + eps * normal)
    # triangle in the z = 0 plane hit from above
    tri = np.zeros((3, 3))
    tri[1, 0] = 1
    tri[2, 1] = 1
    r0, step = np.array([0, 0, 0.5]), np.array([0, 0, -1.0])
    d = lib.dso_u_ray_triangle(_p(tri), _p(r0), _p(step))
    assert 0 < d < 1.0
    n = np.zeros(3)
    lib.dso_u_triangle_normal(_p(tri), _p(n))
    lib.dso_u_reflection(_p(r0), _p(step), d, _p(n), 1e-10)
    npt.assert_almost_equal(step, [0.0, 0.0, 1.0])
    npt.assert_almost_equal(r0, [0.0, 0.0, 1e-10])
    # crossing
    tri = np.zeros((3, 3))
    tri[0, 2] = 1
    tri[1, 0] = 1
    tri[1, 2] = 1
    tri[2, 1] = 1
    tri[2, 2] = 1
    r0, step = np.zeros(3), np.array([0, 0, 1.0])
    d = lib.dso_u_ray_triangle(_p(tri), _p(r0), _p(step))
    lib.dso_u_triangle_normal(_p(tri), _p(n))
    lib.dso_u_crossing(_p(r0), _p(step), d, _p(n), 1e-10)
    npt.assert_almost_equal(r0, [0.0, 0, 1 + 1e-10])


def test_subvoxel_ranges(oracle_libm):
    """_ll/_ul_subvoxel_overlap(_periodic), simulations.py:616-679: boundary-equality rules."""
    lib = oracle_libm.lib
    xs = np.linspace(0, 1.0, 11)

    def rng(x1, x2, periodic):
        ll, ul = C.c_int64(), C.c_int64()
        lib.dso_u_subvoxel_range(_p(xs), len(xs), x1, x2, periodic, C.byref(ll), C.byref(ul))
        return ll.value, ul.value

    assert rng(0.25, 0.35, 0) == (2, 4)
    assert rng(0.35, 0.25, 0) == (2, 4)
    assert rng(-1.0, 0.05, 0) == (0, 1)
    assert rng(0.95, 7.0, 0) == (9, 10)
    assert rng(xs[3], xs[3], 0) == (3, 3)  # a point exactly on a boundary gives an EMPTY range
    assert rng(xs[3], xs[4], 0) == (3, 4)
    assert rng(0.25, 0.26, 0) == (2, 3)
    # periodic images
    assert rng(-0.05, 0.05, 1) == (-1, 1)
    assert rng(0.95, 1.05, 1) == (9, 11)
    assert rng(2.31, 2.32, 1) == (23, 24)
    assert rng(-1.25, -1.15, 1) == (-13, -11)


def test_exact_build_close_to_libm_build(oracle_libm, oracle_exact):
    """Same steps up to 1 ulp(float32) of the log: directions agree to 6e-8."""
    a = oracle_libm.random_steps(oracle_libm.rng_states(20000, 7))
    b = oracle_exact.random_steps(oracle_exact.rng_states(20000, 7))
    d = np.linalg.norm(a - b, axis=1)
    assert d.max() < 1.2e-7
    assert (d < 1e-14).mean() > 0.95
