"""ctypes front end of the CPU oracle (oracle/dsim_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py - never by the ``disimpy_amd`` package.

Two builds of the same C file are exposed:

* ``Oracle(exact=False)``  glibc ``logf``/``cos``; this is the build pinned
  against the reference's golden trajectory (tests/golden/traj_free_seed123.npy).
* ``Oracle(exact=True)``   ``log``/``cos`` taken from
  disimpy_amd/csrc/dsim_math.h; bit-identical to the HIP kernels, used to check
  them walker by walker on every substrate (reflecting geometries are chaotic,
  so a tolerance comparison is meaningless there).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FREE, SPHERE, CYLINDER, ELLIPSOID, MESH = range(5)

_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)

STATE_DTYPE = np.dtype([("s0", np.uint64), ("s1", np.uint64)], align=True)


class Geom(C.Structure):
    """Mirror of ``dso_geom``."""

    _fields_ = [
        ("type", C.c_int),
        ("radius", C.c_double),
        ("semiaxes", C.c_double * 3),
        ("R", C.c_double * 9),
        ("R_inv", C.c_double * 9),
        ("vertices", _f64p),
        ("faces", _i64p),
        ("xs", _f64p),
        ("ys", _f64p),
        ("zs", _f64p),
        ("n_xs", C.c_int64),
        ("n_ys", C.c_int64),
        ("n_zs", C.c_int64),
        ("subvoxel_indices", _i64p),
        ("triangle_indices", _i64p),
        ("n_sv", C.c_int64 * 3),
        ("perm_prob", C.c_double),
        ("max_iter", C.c_int64),
        ("epsilon", C.c_double),
    ]


def build(force=False):
    """Compile both oracle builds with the committed Makefile (gcc only)."""
    targets = [os.path.join(_HERE, n) for n in ("libdsim_oracle.so", "libdsim_oracle_exact.so")]
    if force or not all(os.path.exists(t) for t in targets):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return targets


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class Oracle:
    def __init__(self, exact=False):
        name = "libdsim_oracle_exact.so" if exact else "libdsim_oracle.so"
        other = os.environ.get("DSIM_ORACLE_DIR")  # tools/asan_cpu.sh: a sanitizer build of the same source elsewhere
        if not other:
            build()
        self.lib = lib = C.CDLL(os.path.join(other or _HERE, name))
        self.exact = exact
        assert bool(lib.dso_shared_math()) == exact
        assert lib.dso_sizeof_geom() == C.sizeof(Geom)
        lib.dso_rng_init_states.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]
        lib.dso_run.argtypes = [
            C.POINTER(Geom), C.c_int64, C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_double, C.c_double,
            _f64p, _f64p, C.c_void_p, _u8p, C.c_int, C.c_int, _f64p, C.c_int,
        ]  # fmt: skip
        lib.dso_signal.argtypes = [_f64p, _u8p, C.c_int, C.c_int64, _f64p, _f64p]
        lib.dso_fill_mesh.argtypes = [
            C.c_int64, C.POINTER(Geom), _f64p, C.c_int, C.c_uint64, C.c_int64, _f64p, C.c_int,
        ]  # fmt: skip
        lib.dso_fill_mesh.restype = C.c_int
        lib.dso_u_dot.restype = C.c_double
        lib.dso_u_line_circle.restype = C.c_double
        lib.dso_u_line_sphere.restype = C.c_double
        lib.dso_u_line_ellipsoid.restype = C.c_double
        lib.dso_u_ray_triangle.restype = C.c_double
        lib.dso_u_line_circle.argtypes = [_f64p, _f64p, C.c_double]
        lib.dso_u_line_sphere.argtypes = [_f64p, _f64p, C.c_double]
        lib.dso_u_reflection.argtypes = [_f64p, _f64p, C.c_double, _f64p, C.c_double]
        lib.dso_u_crossing.argtypes = [_f64p, _f64p, C.c_double, _f64p, C.c_double]
        lib.dso_u_random_steps.argtypes = [C.c_void_p, C.c_int64, _f64p]
        lib.dso_u_subvoxel_range.argtypes = [_f64p, C.c_int64, C.c_double, C.c_double, C.c_int, _i64p, _i64p]
        lib.dso_u_next.argtypes = [C.c_void_p]
        lib.dso_u_next.restype = C.c_uint64
        lib.dso_u_uniform_f64.argtypes = [C.c_void_p]
        lib.dso_u_uniform_f64.restype = C.c_double
        lib.dso_u_logf.argtypes = [C.c_float]
        lib.dso_u_logf.restype = C.c_float
        lib.dso_u_cos.argtypes = [C.c_double]
        lib.dso_u_cos.restype = C.c_double
        lib.dso_u_logf_sweep.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.dso_u_logf_sweep.restype = C.c_int64
        lib.dso_u_cos_sweep.argtypes = [C.c_uint32, C.c_uint32, _f64p, _f64p]
        lib.dso_u_cos_sweep.restype = None
        lib.dso_u_mesh_pass.argtypes = [C.POINTER(Geom), C.c_int64, C.c_int64, _f64p, _f64p, _f64p, C.c_void_p,
                                        C.POINTER(C.c_int32)]
        lib.dso_u_mesh_pass.restype = None
        _i32p = C.POINTER(C.c_int32)
        lib.dso_u_mesh_pass_stats.argtypes = [C.POINTER(Geom), C.c_int64, _f64p, _f64p, _f64p, _i32p, _i32p, _i32p]
        lib.dso_u_mesh_pass_stats.restype = None

    # ------------------------------------------------------------------ RNG
    def rng_states(self, n, seed, subsequence_start=0):
        """create_xoroshiro128p_states(n, seed) as a host structured array."""
        st = np.zeros(n, dtype=STATE_DTYPE)
        self.lib.dso_rng_init_states(st.ctypes.data, n, seed, subsequence_start)
        return st

    def set_log_mode(self, cudasim):
        self.lib.dso_set_log_mode(1 if cudasim else 0)

    def random_steps(self, states):
        steps = np.zeros((len(states), 3))
        self.lib.dso_u_random_steps(states.ctypes.data, len(states), _ptr(steps, _f64p))
        return steps

    # ------------------------------------------------------------- geometry
    def geom(self, kind, *, radius=0.0, semiaxes=None, R=None, R_inv=None, mesh=None, perm_prob=0.0,
             max_iter=1000, epsilon=1e-13):
        """Build a Geom.  ``mesh`` = dict(vertices, faces, xs, ys, zs, subvoxel_indices,
        triangle_indices, n_sv) of numpy arrays; they are kept alive on the returned object."""
        g = Geom()
        g.type = kind
        g.radius = radius
        g.max_iter = max_iter
        g.epsilon = epsilon
        g.perm_prob = perm_prob
        keep = []
        if semiaxes is not None:
            g.semiaxes = (C.c_double * 3)(*np.asarray(semiaxes, dtype=np.float64))
        if R is not None:
            g.R = (C.c_double * 9)(*np.asarray(R, dtype=np.float64).ravel())
            g.R_inv = (C.c_double * 9)(*np.asarray(R_inv, dtype=np.float64).ravel())
        if mesh is not None:
            v = np.ascontiguousarray(mesh["vertices"], dtype=np.float64)
            f = np.ascontiguousarray(mesh["faces"], dtype=np.int64)
            xs = np.ascontiguousarray(mesh["xs"], dtype=np.float64)
            ys = np.ascontiguousarray(mesh["ys"], dtype=np.float64)
            zs = np.ascontiguousarray(mesh["zs"], dtype=np.float64)
            svi = np.ascontiguousarray(mesh["subvoxel_indices"], dtype=np.int64)
            tri = np.ascontiguousarray(mesh["triangle_indices"], dtype=np.int64)
            keep += [v, f, xs, ys, zs, svi, tri]
            g.vertices, g.faces = _ptr(v, _f64p), _ptr(f, _i64p)
            g.xs, g.ys, g.zs = _ptr(xs, _f64p), _ptr(ys, _f64p), _ptr(zs, _f64p)
            g.n_xs, g.n_ys, g.n_zs = len(xs), len(ys), len(zs)
            g.subvoxel_indices, g.triangle_indices = _ptr(svi, _i64p), _ptr(tri, _i64p)
            g.n_sv = (C.c_int64 * 3)(*[int(k) for k in mesh["n_sv"]])
        g._keep = keep
        return g

    # ------------------------------------------------------------ time loop
    def run(self, g, gradient, dt, step_l, positions, states, phases=None, iter_exc=None, t_begin=0,
            t_end=None, want_traj=False, n_threads=1):
        """Advance ``positions`` (N,3), ``states`` (N,), ``phases`` (M,N) and
        ``iter_exc`` (N,) in place over time steps [t_begin, t_end)."""
        gradient = np.asarray(gradient, dtype=np.float64)
        M, T = gradient.shape[:2]
        N = positions.shape[0]
        gx = np.ascontiguousarray(gradient[:, :, 0])
        gy = np.ascontiguousarray(gradient[:, :, 1])
        gz = np.ascontiguousarray(gradient[:, :, 2])
        if phases is None:
            phases = np.zeros((M, N))
        if iter_exc is None:
            iter_exc = np.zeros(N, dtype=np.uint8)
        if t_end is None:
            t_end = T
        assert positions.flags.c_contiguous and positions.dtype == np.float64
        assert states.dtype == STATE_DTYPE and len(states) >= N
        traj = np.zeros((t_end - t_begin, N, 3)) if want_traj else None
        self.lib.dso_run(
            C.byref(g), N, M, T, _ptr(gx, _f64p), _ptr(gy, _f64p), _ptr(gz, _f64p), dt, step_l,
            _ptr(positions, _f64p), _ptr(phases, _f64p), states.ctypes.data, _ptr(iter_exc, _u8p),
            t_begin, t_end, _ptr(traj, _f64p) if want_traj else None, n_threads,
        )  # fmt: skip
        return phases, iter_exc, traj

    def mesh_pass(self, g, r0, step, step_l, states, n_pass=1):
        """The reflect loop of the mesh kernel (simulations.py:924-997) on caller-supplied rays: ``n_pass`` > 0
        iterations at most, or (0) the whole loop plus the final advance.  Returns copies
        (r0, step, step_l, states, flags); flags bit 0 = loop over, bit 1 = iter_exc (n_pass = 0 only)."""
        r0 = np.array(r0, dtype=np.float64, order="C")
        step = np.array(step, dtype=np.float64, order="C")
        step_l = np.array(step_l, dtype=np.float64, order="C")
        st = np.array(states, dtype=STATE_DTYPE, order="C")
        n = len(step_l)
        assert r0.shape == (n, 3) and step.shape == (n, 3) and len(st) == n
        flags = np.zeros(n, dtype=np.int32)
        self.lib.dso_u_mesh_pass(C.byref(g), n, n_pass, _ptr(r0, _f64p), _ptr(step, _f64p), _ptr(step_l, _f64p),
                                 st.ctypes.data, flags.ctypes.data_as(C.POINTER(C.c_int32)))
        return r0, step, step_l, st, flags

    def mesh_pass_stats(self, g, r0, step, step_l):
        """First iteration of the reflect loop: (distinct triangles tied at the minimum d of a hit, list entries
        visited, subvoxels in range) per ray."""
        r0 = np.ascontiguousarray(r0, dtype=np.float64)
        step = np.ascontiguousarray(step, dtype=np.float64)
        step_l = np.ascontiguousarray(step_l, dtype=np.float64)
        n = len(step_l)
        out = [np.zeros(n, dtype=np.int32) for _ in range(3)]
        i32p = C.POINTER(C.c_int32)
        self.lib.dso_u_mesh_pass_stats(C.byref(g), n, _ptr(r0, _f64p), _ptr(step, _f64p), _ptr(step_l, _f64p),
                                       *[o.ctypes.data_as(i32p) for o in out])
        return tuple(out)

    def signal(self, phases, iter_exc):
        M, N = phases.shape
        c = np.zeros(M)
        s = np.zeros(M)
        self.lib.dso_signal(_ptr(phases, _f64p), _ptr(iter_exc, _u8p), M, N, _ptr(c, _f64p), _ptr(s, _f64p))
        return c, s

    def fill_mesh(self, n_points, g, voxel_size, intra, seed, cuda_bs=128, max_launches=10000):
        out = np.zeros((n_points, 3))
        vs = np.ascontiguousarray(voxel_size, dtype=np.float64)
        n = self.lib.dso_fill_mesh(n_points, C.byref(g), _ptr(vs, _f64p), int(bool(intra)), seed, cuda_bs,
                                   _ptr(out, _f64p), max_launches)
        if n < 0:
            raise RuntimeError("oracle fill_mesh did not converge")
        return out
