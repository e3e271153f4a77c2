"""Substrate objects (mirror of disimpy/substrates.py).

A substrate describes the simulated microstructure.  Constructors validate
their arguments exactly like the reference (same ``ValueError`` messages) and
return a ``_Substrate`` attribute bag that ``simulations.simulation`` consumes.
For triangular meshes the voxelised triangle index (``xs, ys, zs,
triangle_indices, subvoxel_indices``) is built by the C routine
``dsim_mesh_subdivide`` (disimpy_amd/csrc/dsim_hip.hip), a host-only
restatement of ``_mesh_space_subdivision`` (disimpy/substrates.py:467-536)
with identical membership.
"""
import numpy as np

from . import _lib


class _Substrate:
    """Attribute bag describing the microstructure (disimpy/substrates.py:11-44)."""

    def __init__(self, substrate_type, **kwargs):
        self.type = substrate_type
        if substrate_type == "sphere":
            self.radius = kwargs["radius"]
        elif substrate_type == "cylinder":
            self.radius = kwargs["radius"]
            self.orientation = kwargs["orientation"]
        elif substrate_type == "ellipsoid":
            self.semiaxes = kwargs["semiaxes"]
            self.R = kwargs["R"]
        elif substrate_type == "mesh":
            for name in ("vertices", "faces", "voxel_size", "periodic", "init_pos", "n_sv", "perm_prob"):
                setattr(self, name, kwargs[name])
            quiet = kwargs["quiet"]
            if not quiet:
                print("Dividing the mesh into subvoxels")
            self.xs, self.ys, self.zs, self.triangle_indices, self.subvoxel_indices = _mesh_space_subdivision(
                self.vertices, self.faces, self.voxel_size, self.n_sv
            )
            if not quiet:
                print("Finished dividing the mesh into subvoxels")


def free():
    """Substrate for free diffusion."""
    return _Substrate("free")


def _is_float_array(a, shape=None, ndim=None):
    if not isinstance(a, np.ndarray) or not np.issubdtype(a.dtype, np.floating):
        return False
    if shape is not None and a.shape != shape:
        return False
    if ndim is not None and a.ndim != ndim:
        return False
    return True


def sphere(radius):
    """Substrate for diffusion inside a sphere of the given radius (float, > 0)."""
    if not isinstance(radius, float) or radius <= 0:
        raise ValueError(f"Incorrect value ({radius}) for radius")
    return _Substrate("sphere", radius=radius)


def cylinder(radius, orientation):
    """Substrate for diffusion inside an infinite cylinder.

    Parameters
    ----------
    radius : float
    orientation : numpy.ndarray
        Floating-point array with shape (3,); stored normalised.
    """
    if not isinstance(radius, float) or radius <= 0:
        raise ValueError(f"Incorrect value ({radius}) for radius")
    if not _is_float_array(orientation, shape=(3,)):
        raise ValueError(f"Incorrect value ({orientation}) for orientation")
    return _Substrate("cylinder", radius=radius, orientation=orientation / np.linalg.norm(orientation))


def ellipsoid(semiaxes, R=np.eye(3)):
    """Substrate for diffusion inside an ellipsoid.

    Parameters
    ----------
    semiaxes : numpy.ndarray
        Floating-point array with shape (3,).
    R : numpy.ndarray, optional
        Rotation matrix (3, 3) applied to the axis-aligned ellipsoid.
    """
    if not _is_float_array(semiaxes, shape=(3,)):
        raise ValueError(f"Incorrect value ({semiaxes}) for semiaxes")
    if not _is_float_array(R, shape=(3, 3)):
        raise ValueError(f"Incorrect value ({R}) for R")
    if not np.isclose(np.linalg.det(R), 1) or not np.all(np.isclose(R.T, np.linalg.inv(R))):
        raise ValueError(f"R ({R}) is not a valid rotation matrix")
    return _Substrate("ellipsoid", semiaxes=semiaxes, R=R)


def mesh(
    vertices,
    faces,
    periodic,
    padding=np.zeros(3),
    init_pos="uniform",
    n_sv=np.array([50, 50, 50]),
    quiet=False,
    perm_prob=0,
):
    """Substrate for diffusion restricted by a triangular mesh (disimpy/substrates.py:143-269).

    The simulated voxel is the axis-aligned bounding box of the triangles plus
    ``padding``; the vertices are shifted so that its lower corner is the
    origin.  With ``periodic=False`` the 12 triangles of the voxel boundary are
    appended as an impermeable wall.

    Parameters
    ----------
    vertices : numpy.ndarray
        Floating-point array (number of vertices, 3).
    faces : numpy.ndarray
        Integer array (number of triangles, 3).
    periodic : bool
    padding : numpy.ndarray, optional
        Floating-point array (3,).
    init_pos : numpy.ndarray or str, optional
        Array (number of walkers, 3) or one of 'uniform', 'intra', 'extra'.
    n_sv : numpy.ndarray, optional
        Integer array (3,): number of subvoxels along each axis.
    quiet : bool, optional
    perm_prob : float, optional
        Probability that a spin passes through a triangle.
    """
    if not _is_float_array(vertices, ndim=2) or vertices.shape[1] != 3:
        raise ValueError(f"Incorrect value ({vertices}) for vertices.")
    if (
        not isinstance(faces, np.ndarray)
        or faces.ndim != 2
        or faces.shape[1] != 3
        or not np.issubdtype(faces.dtype, np.integer)
    ):
        raise ValueError(f"Incorrect value ({faces}) for faces.")
    if not isinstance(periodic, bool):
        raise ValueError(f"Incorrect value ({periodic}) for periodic")
    if not _is_float_array(padding, shape=(3,)):
        raise ValueError(f"Incorrect value ({padding}) for padding")
    if isinstance(init_pos, np.ndarray):
        if not _is_float_array(init_pos, ndim=2) or init_pos.shape[1] != 3:
            raise ValueError(f"Incorrect value ({init_pos}) for init_pos")
    elif isinstance(init_pos, str):
        if init_pos not in ("uniform", "intra", "extra"):
            raise ValueError(f"Incorrect value ({init_pos}) for init_pos")
    else:
        raise ValueError(f"Incorrect value ({init_pos}) for init_pos")
    if not isinstance(n_sv, np.ndarray) or n_sv.shape != (3,) or not np.issubdtype(n_sv.dtype, np.integer):
        raise ValueError(f"Incorrect value ({n_sv}) for n_sv")
    if (perm_prob != 0 and not isinstance(perm_prob, float)) or perm_prob < 0 or perm_prob > 1:
        raise ValueError(f"Incorrect value ({perm_prob}) for perm_prob.")
    if not quiet:
        print("Aligning the corner of the simulated voxel with the origin")
    shift = -np.min(vertices, axis=0) + padding
    vertices = vertices + shift
    if not quiet:
        print(f"Moved the vertices by {shift}")
    voxel_size = np.max(vertices, axis=0) + padding
    if not periodic:
        wall_vertices, wall_faces = _aabb_to_mesh(np.zeros(3), voxel_size)
        faces = np.vstack((faces, wall_faces + len(vertices)))
        vertices = np.vstack((vertices, wall_vertices))
    return _Substrate(
        "mesh",
        vertices=vertices,
        faces=faces,
        voxel_size=voxel_size,
        n_sv=n_sv,
        periodic=periodic,
        init_pos=init_pos,
        quiet=quiet,
        perm_prob=perm_prob,
    )


def _triangle_box_overlap(triangle, box):
    """Akenine-Moller triangle/box separating-axis test (disimpy/substrates.py:291-368), in C."""
    return _lib.triangle_box_overlap(triangle, box)


def _interval_sv_overlap(xs, x1, x2):
    """Subvoxel index range overlapped by [x1, x2] (disimpy/substrates.py:371-419), in C."""
    return _lib.interval_sv_overlap(xs, x1, x2)


def _triangle_aabb(triangle):
    """Corners of the axis-aligned bounding box of a (3, 3) triangle (disimpy/substrates.py:423-441)."""
    return np.stack([np.min(triangle, axis=0), np.max(triangle, axis=0)]).astype(float)


def _box_subvoxel_overlap(box, xs, ys, zs):
    """Lowest/highest overlapped subvoxel index along each axis (disimpy/substrates.py:445-464)."""
    return np.array(
        [_interval_sv_overlap(b, box[0, i], box[1, i]) for i, b in enumerate((xs, ys, zs))], dtype=np.int32
    )


def _mesh_space_subdivision(vertices, faces, voxel_size, n_sv):
    """Voxelised triangle index (disimpy/substrates.py:467-536).

    Returns
    -------
    xs, ys, zs : numpy.ndarray
        Subvoxel boundaries, ``linspace(0, voxel_size[k], n_sv[k] + 1)``.
    triangle_indices : numpy.ndarray
        1D int array: triangle ids of all subvoxels, concatenated.
    subvoxel_indices : numpy.ndarray
        (prod(n_sv), 2) int array: [start, end) of every subvoxel in ``triangle_indices``.
    """
    xs = np.linspace(0, voxel_size[0], n_sv[0] + 1)
    ys = np.linspace(0, voxel_size[1], n_sv[1] + 1)
    zs = np.linspace(0, voxel_size[2], n_sv[2] + 1)
    triangle_indices, subvoxel_indices = _lib.mesh_subdivide(vertices, faces, xs, ys, zs, n_sv)
    return xs, ys, zs, triangle_indices.astype(int), subvoxel_indices.astype(int)


def _aabb_to_mesh(a, b):
    """Triangulated surface (8 vertices, 12 triangles) of the box with corners ``a`` and ``b``.

    Vertex and face numbering are part of the contract (disimpy/substrates.py:539-570): the
    wall triangles are appended to the user's mesh and identified later by index.
    """
    corner = np.array([a, b])
    pick = np.array(  # which of (a, b) supplies x, y, z of every vertex
        [[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1], [0, 1, 1], [0, 0, 1], [0, 1, 0], [1, 0, 1]]
    )
    vertices = corner[pick, np.arange(3)]
    faces = np.array(
        [[0, 1, 2], [0, 6, 2], [5, 7, 3], [5, 4, 3], [1, 2, 3], [1, 7, 3],
         [0, 6, 4], [0, 5, 4], [0, 1, 7], [0, 5, 7], [6, 2, 3], [6, 4, 3]]
    )  # fmt: skip
    return vertices, faces
