"""Analytic substrates over random shapes (``-m gpu``): spheres, cylinders and ellipsoids (disimpy/simulations.py:704-881)
with random radii / semi-axes (0.3 - 20 um, aspect ratios up to 30), random orientations and step lengths from a hundredth
of the size to several times it (many reflections per step, the iteration limit among them) - every walker bit for bit
against the exact oracle.  The line-quadric intersections end in a square root the kernels compute with their own
correctly rounded sequence; this is where a wrong last bit would show."""
import numpy as np
import pytest

from test_gpu_parity import CYLINDER, ELLIPSOID, SPHERE, _assert_identical, _gpu_run, _oracle_run, _random_gradient, _rot, lib  # noqa: F401

pytestmark = pytest.mark.gpu


def _shape(seed):
    rng = np.random.RandomState(1000 + seed)
    size = 10 ** rng.uniform(np.log10(3e-7), np.log10(2e-5))
    ratio = 10 ** rng.uniform(-2, 0.7)  # step length / size
    return rng, size, ratio * size


@pytest.mark.parametrize("seed", range(6))
def test_random_spheres(lib, oracle_exact, seed):  # noqa: F811
    from disimpy_amd import simulations

    rng, radius, step_l = _shape(seed)
    n, T = 1500, 60
    g = _random_gradient(1, T, seed=seed)
    pos0 = simulations._fill_sphere(n, radius, simulations._host_rng(seed))
    kw = dict(radius=radius, max_iter=int(rng.choice([5, 1000])), epsilon=1e-13)
    ref = _oracle_run(oracle_exact, SPHERE, n, g, 1e-4, step_l, pos0, 50 + seed, kw)
    gpu = _gpu_run(lib, SPHERE, n, g, 1e-4, step_l, pos0, 50 + seed, kw, chunks=[0, 17, T])
    _assert_identical(gpu, ref)


@pytest.mark.parametrize("seed", range(8))
def test_random_cylinders(lib, oracle_exact, seed):  # noqa: F811
    from disimpy_amd import simulations, utils

    rng, radius, step_l = _shape(100 + seed)
    n, T = 1500, 60
    g = _random_gradient(2, T, seed=seed)
    orientation = rng.normal(size=3) if seed else np.array([1.0, 0.0, 0.0])  # once along the axis the rotation maps to
    orientation /= np.linalg.norm(orientation)
    R = utils.vec2vec_rotmat(orientation, np.array([1.0, 0, 0]))
    R_inv = np.linalg.inv(R)
    pos0 = simulations._initial_positions_cylinder(n, radius, R_inv, simulations._host_rng(seed))
    kw = dict(radius=radius, R=R, R_inv=R_inv, max_iter=int(rng.choice([5, 1000])), epsilon=1e-13)
    ref = _oracle_run(oracle_exact, CYLINDER, n, g, 1e-4, step_l, pos0, 60 + seed, kw)
    gpu = _gpu_run(lib, CYLINDER, n, g, 1e-4, step_l, pos0, 60 + seed, kw, chunks=[0, 31, T])
    _assert_identical(gpu, ref)


@pytest.mark.parametrize("seed", range(8))
def test_random_ellipsoids(lib, oracle_exact, seed):  # noqa: F811
    from disimpy_amd import simulations

    rng, size, step_l = _shape(200 + seed)
    n, T = 1500, 60
    g = _random_gradient(1, T, seed=seed)
    semiaxes = size * 10 ** rng.uniform(-1.5, 0, size=3)  # aspect ratios up to 30
    semiaxes[rng.randint(3)] = size
    R_inv, R = _rot(300 + seed)
    pos0 = simulations._initial_positions_ellipsoid(n, semiaxes, R_inv, simulations._host_rng(seed))
    kw = dict(semiaxes=semiaxes, R=R, R_inv=R_inv, max_iter=int(rng.choice([5, 1000])), epsilon=1e-13)
    ref = _oracle_run(oracle_exact, ELLIPSOID, n, g, 1e-4, step_l, pos0, 70 + seed, kw)
    gpu = _gpu_run(lib, ELLIPSOID, n, g, 1e-4, step_l, pos0, 70 + seed, kw, chunks=[0, 9, T])
    _assert_identical(gpu, ref)
