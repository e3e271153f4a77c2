"""Gradient-array helpers (mirror of disimpy/gradients.py).

A gradient array is a ``numpy.ndarray`` of shape (number of measurements,
number of time points, 3) in T/m; ``dt`` is the duration of one time step in
seconds.  These are host-side numpy routines; the arrays they produce are what
``simulations.simulation`` uploads to the GPU.
"""
import numpy as np

from . import utils

GAMMA = 267.513e6  # gyromagnetic ratio of the simulated spins (disimpy/gradients.py:13)

_trapezoid = getattr(np, "trapezoid", None) or np.trapz


def interpolate_gradient(gradient, dt, n_t):
    """Linearly resample every waveform to ``n_t`` time points (disimpy/gradients.py:16-45).

    Returns
    -------
    interp_g : numpy.ndarray
        Array of shape (number of measurements, n_t, 3).
    dt : float
        Duration of a time step after resampling.
    """
    n_old = gradient.shape[1]
    T = dt * (n_old - 1)
    new_dt = T / (n_t - 1)
    t_new = np.linspace(0, T, n_t)
    t_old = np.linspace(0, T, n_old)
    interp_g = np.zeros((gradient.shape[0], n_t, 3))
    for m, waveform in enumerate(gradient):
        for axis in range(3):
            interp_g[m, :, axis] = np.interp(t_new, t_old, waveform[:, axis])
    return interp_g, new_dt


def calc_q(gradient, dt):
    """q(t) = GAMMA * integral of g (trapezoidal cumulative sum), disimpy/gradients.py:48-70."""
    increments = dt * (gradient[:, 1::, :] + gradient[:, 0:-1, :]) / 2
    q = np.zeros(gradient.shape[0:1] + (gradient.shape[1], 3))
    q[:, 1:, :] = np.cumsum(increments, axis=1)
    return GAMMA * q


def calc_b(gradient, dt):
    """b = integral of |q|^2 dt for every measurement, disimpy/gradients.py:73-90."""
    q = calc_q(gradient, dt)
    return _trapezoid(np.linalg.norm(q, axis=2) ** 2, axis=1, dx=dt)


def set_b(gradient, dt, b):
    """Rescale every waveform so that its b-value equals ``b`` (disimpy/gradients.py:93-115)."""
    b = np.asarray(b)
    current = calc_b(gradient, dt)
    if np.any(np.isclose(current, 0)):
        raise Exception("b-value can not be changed for measurements with b = 0")
    ratio = b / current
    return gradient * np.sqrt(ratio)[:, np.newaxis, np.newaxis]


def rotate_gradient(gradient, Rs):
    """Apply rotation matrix ``Rs[i]`` to measurement i (disimpy/gradients.py:118-141)."""
    g = np.zeros(gradient.shape)
    for i, R in enumerate(Rs):
        if not np.isclose(np.linalg.det(R), 1) or not np.all(np.isclose(R.T, np.linalg.inv(R))):
            raise ValueError(f"Rs[{i}] ({R}) is not a valid rotation matrix")
        g[i, :, :] = np.matmul(R, gradient[i, :, :].T).T
    return g


def pgse(delta, DELTA, n_t, bvals, bvecs):
    """Pulsed-gradient spin-echo gradient array (disimpy/gradients.py:144-179).

    A +1/-1 template of 10^6 points along x is resampled to ``n_t`` points,
    replicated per b-value, scaled with :func:`set_b` and rotated onto each b-vector.

    Returns
    -------
    gradient : numpy.ndarray
    dt : float
    """
    n_template = int(1e6)
    template = np.zeros((1, n_template, 3))
    dt = (delta + DELTA) / (n_template - 1)
    n_on = np.round(delta / dt).astype(int)
    template[0, 1:n_on, 0] = 1
    template[0, -n_on:-1, 0] = -1
    one, dt = interpolate_gradient(template, dt, n_t)
    gradient = np.concatenate([one for _ in bvals], axis=0)
    gradient = set_b(gradient, dt, bvals)
    x_axis = np.array([1.0, 0.0, 0.0])
    Rs = np.zeros((len(bvals), 3, 3))
    for i, bvec in enumerate(bvecs):
        Rs[i] = utils.vec2vec_rotmat(x_axis, bvec)
    return rotate_gradient(gradient, Rs), dt


def load_camino_scheme_file(path):
    """Gradient array from a Camino GRADIENT_WAVEFORM scheme file (disimpy/gradients.py:182-214)."""
    with open(path, "r") as f:
        if f.readline().strip() != "VERSION: GRADIENT_WAVEFORM":
            raise Exception("The scheme file does not start with 'VERSION: GRADIENT_WAVEFORM'")
    scheme = np.loadtxt(path, skiprows=1, ndmin=2)
    dts = scheme[:, 1]
    if len(set(dts)) != 1:
        raise Exception(
            "Not all rows of the scheme file have the same time step duration. "
            "Disimpy does not support scheme files with multiple time step durations."
        )
    gradient = scheme[:, 2::].reshape(len(scheme), -1, 3)
    return gradient, dts[0]
