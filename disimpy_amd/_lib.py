"""ctypes binding of the C ABI in include/dsim.h (libdsim_hip.so).

There is no CPU fallback: if the HIP library is missing, or no AMD GPU is
visible, every compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSIM_LIB") or os.path.join(_HERE, "csrc", "libdsim_hip.so")

# every symbol include/dsim.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "dsim_abi_version", "dsim_last_error", "dsim_device_count", "dsim_rng_init", "dsim_rng_init_device",
    "dsim_create", "dsim_destroy", "dsim_set_positions", "dsim_set_rng_states", "dsim_set_mesh",
    "dsim_run_free", "dsim_run_sphere", "dsim_run_cylinder", "dsim_run_ellipsoid", "dsim_run_mesh",
    "dsim_get_signal", "dsim_get_positions", "dsim_get_phases", "dsim_get_iter_exc", "dsim_get_rng_states",
    "dsim_fill_mesh", "dsim_mesh_subdivide", "dsim_mesh_clearance_map", "dsim_triangle_box_overlap", "dsim_interval_sv_overlap",
    "dsim_last_run_ms", "dsim_last_walk_ms", "dsim_last_walk_busy_ms", "dsim_set_tuning", "dsim_set_recording", "dsim_get_recorded",
    "dsim_write_traj", "dsim_write_traj_line", "dsim_get_all_signals", "dsim_fill_open", "dsim_fill_run", "dsim_fill_close", "dsim_mesh_fine_grid",
    "dsim_create_multi", "dsim_shard_count", "dsim_get_signal_limbs", "dsim_signal_from_limbs", "dsim_partition_by_position",
    "dsim_partition_walkers", "dsim_set_walker_ids", "dsim_fill_mesh_multi", "dsim_mesh_walker_cost",
)  # fmt: skip

_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_h = C.c_void_p

_lib = None


class DsimError(RuntimeError):
    """A call into libdsim_hip.so failed."""


def load():
    """Load libdsim_hip.so (built by ``__graft_entry__.build()`` / ``make -C disimpy_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsimError(
            "HIP extension %s is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C disimpy_amd/csrc` (there is no CPU fallback)" % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    missing = [name for name in SYMBOLS if not hasattr(lib, name)]
    if missing and not (os.environ.get("DSIM_LIB") and os.environ.get("DSIM_LIB_OLD_ABI")):
        # also for a build named by DSIM_LIB (A/B timing runs): a library that lacks a declared symbol is not loaded -
        # unless DSIM_LIB_OLD_ABI says that it is an OLDER build kept for a same-box comparison (tools/gpu_ab.sh), whose
        # missing entry points the run does not use
        raise DsimError("%s lacks %s: rebuild it (make -C disimpy_amd/csrc)" % (LIB_PATH, ", ".join(missing)))
    lib.dsim_last_error.restype = C.c_char_p
    lib.dsim_rng_init.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u64p]
    lib.dsim_rng_init_device.argtypes = [_h, C.c_uint64]
    lib.dsim_create.argtypes = [C.POINTER(_h), C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _f64p, _f64p,
                                _f64p, C.c_double, C.c_double]
    if missing:  # DSIM_LIB_OLD_ABI: the calls below that name a missing symbol are skipped
        class _Absent:
            argtypes = restype = None

        for name in missing:
            setattr(lib, name, _Absent())
    lib.dsim_create_multi.argtypes = [C.POINTER(_h), _i32p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _f64p, _f64p,
                                      _f64p, C.c_double, C.c_double]
    lib.dsim_shard_count.argtypes = [_h]
    lib.dsim_destroy.argtypes = [_h]
    lib.dsim_set_positions.argtypes = [_h, _f64p]
    lib.dsim_partition_by_position.argtypes = [_h, _f64p]
    lib.dsim_partition_walkers.argtypes = [_f64p, C.c_int64, C.c_int32, _f64p, _i64p, _i64p]
    lib.dsim_mesh_walker_cost.argtypes = [_f64p, C.c_int64, _i64p, C.c_int64, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, C.c_int64, C.c_double,
                                          _f64p, C.c_int64, _f64p]
    lib.dsim_set_walker_ids.argtypes = [_h, _i64p]
    lib.dsim_set_rng_states.argtypes = [_h, _u64p]
    lib.dsim_set_mesh.argtypes = [_h, _f64p, C.c_int64, _i64p, C.c_int64, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p,
                                  C.c_int64]
    lib.dsim_run_free.argtypes = [_h, C.c_int32, C.c_int32]
    lib.dsim_run_sphere.argtypes = [_h, C.c_double, C.c_int64, C.c_double, C.c_int32, C.c_int32]
    lib.dsim_run_cylinder.argtypes = [_h, C.c_double, _f64p, _f64p, C.c_int64, C.c_double, C.c_int32, C.c_int32]
    lib.dsim_run_ellipsoid.argtypes = [_h, _f64p, _f64p, _f64p, C.c_int64, C.c_double, C.c_int32, C.c_int32]
    lib.dsim_run_mesh.argtypes = [_h, C.c_int64, C.c_double, C.c_double, C.c_int32, C.c_int32]
    lib.dsim_get_signal.argtypes = [_h, _f64p, _f64p, _i64p]
    lib.dsim_get_signal_limbs.argtypes = [_h, _i64p, _i64p]
    lib.dsim_signal_from_limbs.argtypes = [C.c_int32, _i64p, _f64p, _f64p]
    lib.dsim_get_positions.argtypes = [_h, _f64p]
    lib.dsim_get_phases.argtypes = [_h, _f64p]
    lib.dsim_get_iter_exc.argtypes = [_h, _u8p]
    lib.dsim_get_all_signals.argtypes = [_h, _f64p]
    lib.dsim_get_rng_states.argtypes = [_h, _u64p]
    lib.dsim_fill_mesh.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_uint64, _f64p, C.c_int64, _i64p, C.c_int64,
                                   _f64p, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, C.c_int64, _f64p]
    lib.dsim_fill_mesh_multi.argtypes = [_i32p, C.c_int32, C.c_int64, C.c_int, C.c_uint64, _f64p, C.c_int64, _i64p, C.c_int64,
                                         _f64p, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, C.c_int64, _f64p]
    lib.dsim_fill_open.argtypes = [C.POINTER(_h), C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_uint64, _f64p, C.c_int64, _i64p,
                                   C.c_int64, _f64p, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, C.c_int64]
    lib.dsim_fill_run.argtypes = [_h, C.c_int64, _i64p, _f64p, C.c_int64, _i64p]
    lib.dsim_fill_close.argtypes = [_h]
    lib.dsim_mesh_subdivide.argtypes = [_f64p, C.c_int64, _i64p, C.c_int64, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p,
                                        _i64p]
    lib.dsim_mesh_clearance_map.argtypes = [_f64p, C.c_int64, _i64p, C.c_int64, _f64p, _f64p, _f64p, _i64p, C.c_double,
                                            _i32p, C.POINTER(C.c_uint32), _i64p, _i32p]
    lib.dsim_mesh_fine_grid.argtypes = [_f64p, C.c_int64, _i64p, C.c_int64, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, C.c_int64, C.c_double,
                                        _i32p, _i64p, _i64p, _i32p, C.POINTER(C.c_uint32), _i32p, _i32p, C.POINTER(C.c_float),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    lib.dsim_triangle_box_overlap.argtypes = [_f64p, _f64p]
    lib.dsim_interval_sv_overlap.argtypes = [_f64p, C.c_int64, C.c_double, C.c_double, _i64p, _i64p]
    lib.dsim_last_run_ms.argtypes = [_h, _f64p, _i32p]
    lib.dsim_last_walk_ms.argtypes = [_h, _f64p, _i32p]
    lib.dsim_last_walk_busy_ms.argtypes = [_h, _f64p, _i32p]
    lib.dsim_set_tuning.argtypes = [_h, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.dsim_set_recording.argtypes = [_h, C.c_int32]
    lib.dsim_get_recorded.argtypes = [_h, C.c_int32, C.c_int32, _f64p]
    lib.dsim_write_traj.argtypes = [_h, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
    lib.dsim_write_traj_line.argtypes = [C.c_char_p, _f64p, C.c_int64, C.c_int32]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise DsimError("libdsim_hip: %s (code %d)" % (load().dsim_last_error().decode(), rc))


def device_count():
    return int(load().dsim_device_count())


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def p_f64(a):
    return a.ctypes.data_as(_f64p)


def p_i64(a):
    return a.ctypes.data_as(_i64p)


def p_u64(a):
    return a.ctypes.data_as(_u64p)


def rng_init(seed, first_index, n):
    """create_xoroshiro128p_states restated on the host: (n, 2) uint64 of {s0, s1}."""
    out = np.zeros((n, 2), dtype=np.uint64)
    check(load().dsim_rng_init(seed, first_index, n, p_u64(out)))
    return out


def mesh_subdivide(vertices, faces, xs, ys, zs, n_sv):
    """substrates._mesh_space_subdivision in C: returns (triangle_indices, subvoxel_indices) int64."""
    lib = load()
    v = _c(vertices, np.float64)
    f = _c(faces, np.int64)
    xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
    nsv = _c(n_sv, np.int64)
    n_sub = int(np.prod(nsv))
    svi = np.zeros((n_sub, 2), dtype=np.int64)
    n_refs = C.c_int64(0)
    check(lib.dsim_mesh_subdivide(p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv),
                                  p_i64(svi), None, C.byref(n_refs)))
    tri = np.zeros(max(n_refs.value, 0), dtype=np.int64)
    check(lib.dsim_mesh_subdivide(p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv),
                                  p_i64(svi), p_i64(tri) if len(tri) else None, C.byref(n_refs)))
    return tri, svi


def mesh_clearance_map(vertices, faces, xs, ys, zs, n_sv, step_l):
    """The walk kernel's clearance map: (clear_idx (prod n_sv,) int32, bits uint32, F) or None when no map is used."""
    lib = load()
    v, f = _c(vertices, np.float64), _c(faces, np.int64)
    xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
    nsv = _c(n_sv, np.int64)
    n_words, lf = C.c_int64(0), C.c_int32(0)
    check(lib.dsim_mesh_clearance_map(p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv),
                                      step_l, None, None, C.byref(n_words), C.byref(lf)))
    if n_words.value == 0:
        return None
    idx = np.zeros(int(np.prod(nsv)), dtype=np.int32)
    bits = np.zeros(n_words.value, dtype=np.uint32)
    check(lib.dsim_mesh_clearance_map(p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv),
                                      step_l, idx.ctypes.data_as(_i32p), bits.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      C.byref(n_words), C.byref(lf)))
    return idx, bits, 1 << lf.value


def mesh_fine_grid(vertices, faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, step_l):
    """The fine grid of dsim_set_mesh (host only): dict(lk, sv_base or None, cell_start, tri, pos, box (n, 6))."""
    lib = load()
    v, f = _c(vertices, np.float64), _c(faces, np.int64)
    xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
    nsv = _c(n_sv, np.int64)
    svi, tri = _c(subvoxel_indices, np.int64), _c(triangle_indices, np.int64)
    lk, nc, nr = C.c_int32(0), C.c_int64(0), C.c_int64(0)
    args = (p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv), p_i64(svi), p_i64(tri), len(tri), float(step_l))
    check(lib.dsim_mesh_fine_grid(*args, C.byref(lk), C.byref(nc), C.byref(nr), None, None, None, None, None, None, None))
    sv_base = np.full(int(np.prod(nsv)), -1, dtype=np.int32)
    cell_start = np.zeros(nc.value + 1, dtype=np.uint32)
    rt, rp = np.zeros(max(nr.value, 1), dtype=np.int32), np.zeros(max(nr.value, 1), dtype=np.int32)
    box = np.zeros((max(nr.value, 1), 6), dtype=np.float32)
    packed = np.zeros((max(nr.value, 1), 8), dtype=np.uint32)
    quantum = np.zeros(1, dtype=np.float32)
    check(lib.dsim_mesh_fine_grid(*args, C.byref(lk), C.byref(nc), C.byref(nr), sv_base.ctypes.data_as(_i32p),
                                  cell_start.ctypes.data_as(C.POINTER(C.c_uint32)), rt.ctypes.data_as(_i32p), rp.ctypes.data_as(_i32p),
                                  box.ctypes.data_as(C.POINTER(C.c_float)), packed.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  quantum.ctypes.data_as(C.POINTER(C.c_float))))
    return dict(lk=lk.value, sv_base=sv_base if lk.value > 0 else None, cell_start=cell_start, tri=rt[: nr.value], pos=rp[: nr.value],
                box=box[: nr.value], packed=packed[: nr.value], quantum=float(quantum[0]))


def partition_walkers(positions, n_parts, weights=None):
    """The walkers cut into n_parts by space (include/dsim.h, dsim_partition_walkers; host only).  Returns the list of
    the parts' walker indices (each ascending) - Morton order of the positions, balanced counts or, with ``weights``,
    equal total weight."""
    pos = _c(positions, np.float64)
    n = len(pos)
    ids, start = np.zeros(max(n, 1), dtype=np.int64), np.zeros(n_parts + 1, dtype=np.int64)
    w = None if weights is None else _c(weights, np.float64)
    assert w is None or w.shape == (n,)
    check(load().dsim_partition_walkers(p_f64(pos), n, int(n_parts), None if w is None else p_f64(w), p_i64(ids), p_i64(start)))
    return [ids[start[k]: start[k + 1]].copy() for k in range(n_parts)]


def mesh_walker_cost(vertices, faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, step_l, positions):
    """Estimated relative cost of a walker-step at every position of a mesh substrate (include/dsim.h,
    dsim_mesh_walker_cost; host only): the weights of the spatial cut."""
    v, f = _c(vertices, np.float64), _c(faces, np.int64)
    xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
    nsv = _c(n_sv, np.int64)
    svi, tri = _c(subvoxel_indices, np.int64), _c(triangle_indices, np.int64)
    pos = _c(positions, np.float64)
    out = np.zeros(len(pos))
    check(load().dsim_mesh_walker_cost(p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv), p_i64(svi), p_i64(tri),
                                       len(tri), float(step_l), p_f64(pos), len(pos), p_f64(out)))
    return out


def triangle_box_overlap(triangle, box):
    t, b = _c(triangle, np.float64), _c(box, np.float64)
    rc = load().dsim_triangle_box_overlap(p_f64(t), p_f64(b))
    if rc < 0:
        check(rc)
    return bool(rc)


def interval_sv_overlap(xs, x1, x2):
    xs = _c(xs, np.float64)
    ll, ul = C.c_int64(0), C.c_int64(0)
    check(load().dsim_interval_sv_overlap(p_f64(xs), len(xs), x1, x2, C.byref(ll), C.byref(ul)))
    return ll.value, ul.value


class Sim:
    """Owner of one ``dsim_handle``: a range of walkers on one GPU or, with ``devices=[d0, d1, ...]``, cut into balanced
    shards over several devices of this node behind the one handle (dsim_create_multi; a device may be named twice)."""

    def __init__(self, n_walkers, gradient, dt, step_l, device=0, first_walker=0, devices=None):
        lib = load()
        if lib.dsim_device_count() <= 0:
            raise DsimError("no HIP device visible (need an AMD GPU with ROCm); there is no CPU fallback")
        g = np.asarray(gradient, dtype=np.float64)
        self.M, self.T = int(g.shape[0]), int(g.shape[1])
        self.n = int(n_walkers)
        gx, gy, gz = (np.ascontiguousarray(g[:, :, k]) for k in range(3))
        self._h = _h()
        if devices is not None and len(devices) > 1:
            ids = np.ascontiguousarray(devices, dtype=np.int32)
            check(lib.dsim_create_multi(C.byref(self._h), ids.ctypes.data_as(_i32p), len(ids), self.n, first_walker, self.M, self.T,
                                        p_f64(gx), p_f64(gy), p_f64(gz), dt, step_l))
        else:
            if devices is not None and len(devices) == 1:
                device = int(devices[0])
            check(lib.dsim_create(C.byref(self._h), device, self.n, first_walker, self.M, self.T, p_f64(gx), p_f64(gy),
                                  p_f64(gz), dt, step_l))
        self._lib = lib

    def shard_count(self):
        return int(self._lib.dsim_shard_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dsim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # state ------------------------------------------------------------------
    def set_positions(self, pos):
        pos = _c(pos, np.float64)
        assert pos.shape == (self.n, 3)
        check(self._lib.dsim_set_positions(self._h, p_f64(pos)))

    def partition_by_position(self, pos):
        """set_positions that also cuts a multi-device handle's walkers into its shards by space (include/dsim.h)"""
        pos = _c(pos, np.float64)
        assert pos.shape == (self.n, 3)
        check(self._lib.dsim_partition_by_position(self._h, p_f64(pos)))

    def set_walker_ids(self, ids):
        """single-device handle: slot i holds walker first_walker + ids[i] (ascending); None: the range again"""
        if ids is None:
            check(self._lib.dsim_set_walker_ids(self._h, None))
            return
        ids = _c(ids, np.int64)
        assert ids.shape == (self.n,)
        check(self._lib.dsim_set_walker_ids(self._h, p_i64(ids)))

    def set_rng_states(self, states):
        st = _c(states, np.uint64)
        assert st.shape == (self.n, 2)
        check(self._lib.dsim_set_rng_states(self._h, p_u64(st)))

    def rng_init_device(self, seed):
        check(self._lib.dsim_rng_init_device(self._h, seed))

    def set_mesh(self, vertices, faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices):
        v, f = _c(vertices, np.float64), _c(faces, np.int64)
        xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
        nsv = _c(n_sv, np.int64)
        svi, tri = _c(subvoxel_indices, np.int64), _c(triangle_indices, np.int64)
        check(self._lib.dsim_set_mesh(self._h, p_f64(v), len(v), p_i64(f), len(f), p_f64(xs), p_f64(ys), p_f64(zs),
                                      p_i64(nsv), p_i64(svi), p_i64(tri), len(tri)))

    def set_tuning(self, block_size=0, tile_steps=0, meas_per_pass=0, resort_steps=0, gradient_source=0,
                   mesh_traversal=0):
        check(self._lib.dsim_set_tuning(self._h, block_size, tile_steps, meas_per_pass, resort_steps,
                                        gradient_source, mesh_traversal))

    # time loop ----------------------------------------------------------------
    def run_free(self, t0, t1):
        check(self._lib.dsim_run_free(self._h, t0, t1))

    def run_sphere(self, radius, max_iter, epsilon, t0, t1):
        check(self._lib.dsim_run_sphere(self._h, radius, max_iter, epsilon, t0, t1))

    def run_cylinder(self, radius, R, R_inv, max_iter, epsilon, t0, t1):
        R, R_inv = _c(R, np.float64), _c(R_inv, np.float64)
        check(self._lib.dsim_run_cylinder(self._h, radius, p_f64(R), p_f64(R_inv), max_iter, epsilon, t0, t1))

    def run_ellipsoid(self, semiaxes, R, R_inv, max_iter, epsilon, t0, t1):
        ax, R, R_inv = _c(semiaxes, np.float64), _c(R, np.float64), _c(R_inv, np.float64)
        check(self._lib.dsim_run_ellipsoid(self._h, p_f64(ax), p_f64(R), p_f64(R_inv), max_iter, epsilon, t0, t1))

    def run_mesh(self, max_iter, epsilon, perm_prob, t0, t1):
        check(self._lib.dsim_run_mesh(self._h, max_iter, epsilon, float(perm_prob), t0, t1))

    def set_recording(self, enable=True):
        """Keep the positions after every step of the following run_* calls on the device."""
        check(self._lib.dsim_set_recording(self._h, 1 if enable else 0))

    def recorded(self, t0, t1):
        """(t1 - t0, n, 3) positions after steps t0 .. t1-1 of the last (recorded) run_* call."""
        out = np.empty((t1 - t0, 3, self.n), dtype=np.float64)
        check(self._lib.dsim_get_recorded(self._h, t0, t1, p_f64(out)))
        return np.ascontiguousarray(out.transpose(0, 2, 1))

    def write_traj(self, path, t0, t1, append=True):
        """Lines of steps t0 .. t1-1 of the last recorded run_* call -> the trajectories file."""
        check(self._lib.dsim_write_traj(self._h, os.fsencode(path), t0, t1, 1 if append else 0))

    def last_run_ms(self):
        ms, nl = C.c_double(0), C.c_int32(0)
        check(self._lib.dsim_last_run_ms(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def last_walk_ms(self):
        """(ms inside the walk kernel alone, number of its launches) of the last run_* call."""
        ms, nl = C.c_double(0), C.c_int32(0)
        check(self._lib.dsim_last_walk_ms(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def last_walk_busy_ms(self):
        """(ms during which at least one walk kernel was running, streams used) of the last run_* call: a large
        mesh shard is run as two halves on two streams, whose launches overlap."""
        ms, ns = C.c_double(0), C.c_int32(0)
        check(self._lib.dsim_last_walk_busy_ms(self._h, C.byref(ms), C.byref(ns)))
        return ms.value, ns.value

    # outputs ------------------------------------------------------------------
    def signal(self):
        c, s = np.zeros(self.M), np.zeros(self.M)
        nexc = C.c_int64(0)
        check(self._lib.dsim_get_signal(self._h, p_f64(c), p_f64(s), C.byref(nexc)))
        return c, s, nexc.value

    def signal_limbs(self):
        """(limbs (4 M,) int64, n_exceeded): the exact integer sums behind ``signal()`` - limbs of shards / ranks add"""
        limbs = np.zeros(4 * self.M, dtype=np.int64)
        nexc = C.c_int64(0)
        check(self._lib.dsim_get_signal_limbs(self._h, p_i64(limbs), C.byref(nexc)))
        return limbs, nexc.value

    def positions(self):
        out = np.zeros((self.n, 3))
        check(self._lib.dsim_get_positions(self._h, p_f64(out)))
        return out

    def phases(self):
        out = np.zeros((self.M, self.n))
        check(self._lib.dsim_get_phases(self._h, p_f64(out)))
        return out

    def all_signals(self):
        """(M, n) cos(phase), NaN for walkers with iter_exc set - evaluated on the device (simulations.py:1414-1417)"""
        out = np.empty((self.M, self.n))
        check(self._lib.dsim_get_all_signals(self._h, p_f64(out)))
        return out

    def iter_exc(self):
        out = np.zeros(self.n, dtype=np.uint8)
        check(self._lib.dsim_get_iter_exc(self._h, out.ctypes.data_as(_u8p)))
        return out.astype(bool)

    def rng_states(self):
        out = np.zeros((self.n, 2), dtype=np.uint64)
        check(self._lib.dsim_get_rng_states(self._h, p_u64(out)))
        return out


def signal_from_limbs(limbs):
    """(cos (M,), sin (M,)) from summed limbs (dsim_signal_from_limbs: exact sum, rounded once; needs no GPU)"""
    limbs = _c(limbs, np.int64)
    M = len(limbs) // 4
    c, s = np.zeros(M), np.zeros(M)
    check(load().dsim_signal_from_limbs(M, p_i64(limbs), p_f64(c), p_f64(s)))
    return c, s


def write_traj_line(path, positions, append=False):
    """One line of the trajectories file from a host array (n, 3): ``str(v) + " "`` per value."""
    pos = _c(positions, np.float64)
    assert pos.ndim == 2 and pos.shape[1] == 3
    check(load().dsim_write_traj_line(os.fsencode(path), p_f64(pos), len(pos), 1 if append else 0))


def fill_mesh(n_points, intra, seed, vertices, faces, voxel_size, xs, ys, zs, n_sv, subvoxel_indices,
              triangle_indices, device=0, devices=None):
    """disimpy/simulations.py:505-579 on the device; ``devices`` (a list, entries may repeat): the threads of every launch
    split over them (dsim_fill_mesh_multi) - the same rows"""
    lib = load()
    if lib.dsim_device_count() <= 0:
        raise DsimError("no HIP device visible (need an AMD GPU with ROCm); there is no CPU fallback")
    v, f = _c(vertices, np.float64), _c(faces, np.int64)
    vs = _c(voxel_size, np.float64)
    xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
    nsv = _c(n_sv, np.int64)
    svi, tri = _c(subvoxel_indices, np.int64), _c(triangle_indices, np.int64)
    out = np.zeros((n_points, 3))
    if devices is not None and len(devices) > 1:
        ids = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib.dsim_fill_mesh_multi(ids.ctypes.data_as(_i32p), len(ids), n_points, int(bool(intra)), seed, p_f64(v), len(v),
                                       p_i64(f), len(f), p_f64(vs), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv), p_i64(svi),
                                       p_i64(tri), len(tri), p_f64(out)))
        return out
    check(lib.dsim_fill_mesh(device, n_points, int(bool(intra)), seed, p_f64(v), len(v), p_i64(f), len(f), p_f64(vs),
                             p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv), p_i64(svi), p_i64(tri), len(tri),
                             p_f64(out)))
    return out


class FillSession:
    """The _fill_mesh loop (disimpy/simulations.py:505-579) for the threads [thread_begin, thread_end) of its launches
    (dsim_fill_open / dsim_fill_run / dsim_fill_close); see disimpy_amd.sharding.fill_mesh_sharded."""

    def __init__(self, thread_begin, thread_end, intra, seed, vertices, faces, voxel_size, xs, ys, zs, n_sv, subvoxel_indices,
                 triangle_indices, device=0):
        lib = load()
        if lib.dsim_device_count() <= 0:
            raise DsimError("no HIP device visible (need an AMD GPU with ROCm); there is no CPU fallback")
        v, f = _c(vertices, np.float64), _c(faces, np.int64)
        vs = _c(voxel_size, np.float64)
        xs, ys, zs = _c(xs, np.float64), _c(ys, np.float64), _c(zs, np.float64)
        nsv = _c(n_sv, np.int64)
        svi, tri = _c(subvoxel_indices, np.int64), _c(triangle_indices, np.int64)
        self._lib, self._h = lib, _h()
        self.n_threads = thread_end - thread_begin
        check(lib.dsim_fill_open(C.byref(self._h), device, thread_begin, thread_end, int(bool(intra)), seed, p_f64(v), len(v),
                                 p_i64(f), len(f), p_f64(vs), p_f64(xs), p_f64(ys), p_f64(zs), p_i64(nsv), p_i64(svi), p_i64(tri),
                                 len(tri)))

    def run(self, n_launches, capacity):
        """-> (counts (n_launches,), points (rows, 3)): accepted candidates per launch and, in (launch, thread) order, the
        first ``capacity`` of them"""
        counts = np.zeros(n_launches, dtype=np.int64)
        pts = np.empty((max(capacity, 1), 3))
        rows = C.c_int64(0)
        check(self._lib.dsim_fill_run(self._h, n_launches, p_i64(counts), p_f64(pts), capacity, C.byref(rows)))
        return counts, pts[: rows.value]

    def close(self):
        if self._h:
            self._lib.dsim_fill_close(self._h)
            self._h = _h()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
