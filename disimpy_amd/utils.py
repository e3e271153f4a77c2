"""General helpers (mirror of disimpy/utils.py; the matplotlib viewers are out of scope)."""
import numpy as np


def vec2vec_rotmat(v, k):
    """Rotation matrix that turns the direction of ``v`` into the direction of ``k``.

    Same contract as disimpy/utils.py:11-42: Rodrigues' formula about v x k;
    parallel vectors give the identity and antiparallel vectors give ``-eye(3)``
    (the reference's convention, kept although it is a point reflection).

    Parameters
    ----------
    v, k : numpy.ndarray
        1D arrays with length 3.

    Returns
    -------
    R : numpy.ndarray
        3 by 3 matrix.
    """
    v = v / np.linalg.norm(v)
    k = k / np.linalg.norm(k)
    axis = np.cross(v, k)
    s = np.linalg.norm(axis)
    if s < np.finfo(float).eps:
        return -np.eye(3) if np.linalg.norm(v - k) > np.linalg.norm(v) else np.eye(3)
    axis /= s
    angle = np.arccos(np.dot(v, k))
    ax, ay, az = axis
    K = np.array([[0, -az, ay], [az, 0, -ax], [-ay, ax, 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * np.matmul(K, K)
