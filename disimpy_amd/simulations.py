"""Diffusion-weighted MR simulation driver (mirror of disimpy/simulations.py).

``simulation()`` keeps the reference's signature, validation, messages,
warnings and return shapes (disimpy/simulations.py:1051-1429) but runs the
whole time loop in fused HIP kernels on an MI355X through the C ABI of
``include/dsim.h``: one launch per range of time steps instead of one launch +
synchronisation per step, and the signal is reduced on the device instead of
copying the (M, N) phase matrix to the host.  There is no CPU fallback.
"""
import math
import os
import sys
import warnings

import numpy as np

from . import _lib, substrates, utils
from .gradients import GAMMA  # noqa: F401  (re-exported like the reference)

_NO_GPU_MESSAGE = (
    "disimpy_amd was unable to detect an AMD GPU. To run the simulation, check that ROCm is installed, "
    "that /dev/kfd is accessible and that disimpy_amd/csrc/libdsim_hip.so has been built for gfx950."
)


# ----------------------------------------------------------------------------
# initial positions (host; disimpy/simulations.py:346-418)
# ----------------------------------------------------------------------------


def _rejection_sample(rng, n, dim, scale, inside):
    """First ``n`` accepted points of the stream ``p = (u - 0.5) * 2 * scale`` with
    ``u = rng.random_sample(dim)`` per attempt (attempt k consumes draws k*dim ..)."""
    out = np.zeros((n, dim))
    have = 0
    while have < n:
        batch = max(1024, int(2.2 * (n - have)))
        p = (rng.random_sample((batch, dim)) - 0.5) * 2 * scale
        p = p[inside(p)][: n - have]
        out[have : have + len(p)] = p
        have += len(p)
    return out


def _host_rng(seed):
    """The reference seeds numba's CPU Mersenne Twister with ``np.random.seed(seed)``
    inside a jitted function (disimpy/simulations.py:346-350) and draws with
    ``np.random.random(k)``.  Read from numba's source (tests/test_r16_numba_mt19937.py
    restates it line by line): the seed goes through MT19937 ``init_genrand``
    (numba/_random.c:58-71 via numba/cpython/randomimpl.py:207-219) and a double is
    ``(a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53`` (randomimpl.py:133-146) - the same
    stream as NumPy's legacy ``RandomState(seed).random_sample``."""
    return np.random.RandomState(seed)


def _fill_circle(n, radius, rng=None):
    """n points uniformly inside a circle (disimpy/simulations.py:353-366)."""
    rng = rng or _host_rng(None)
    return _rejection_sample(rng, n, 2, radius, lambda p: np.linalg.norm(p, axis=1) < radius)


def _fill_sphere(n, radius, rng=None):
    """n points uniformly inside a sphere (disimpy/simulations.py:369-382)."""
    rng = rng or _host_rng(None)
    return _rejection_sample(rng, n, 3, radius, lambda p: np.linalg.norm(p, axis=1) < radius)


def _fill_ellipsoid(n, semiaxes, rng=None):
    """n points uniformly inside an axis-aligned ellipsoid (disimpy/simulations.py:385-399)."""
    rng = rng or _host_rng(None)
    return _rejection_sample(rng, n, 3, semiaxes, lambda p: np.sum((p / semiaxes) ** 2, axis=1) < 1)


def _initial_positions_cylinder(n_walkers, radius, R, rng=None):
    """Points in the (y, z) disc of the cylinder frame, rotated by R (disimpy/simulations.py:402-409)."""
    positions = np.zeros((n_walkers, 3))
    positions[:, 1:3] = _fill_circle(n_walkers, radius, rng)
    return np.matmul(R, positions.T).T


def _initial_positions_ellipsoid(n_walkers, semiaxes, R, rng=None):
    """Points in the ellipsoid frame rotated by R (disimpy/simulations.py:412-418)."""
    return np.matmul(R, _fill_ellipsoid(n_walkers, semiaxes, rng).T).T


def _strip_wall_triangles(substrate):
    """Mesh arrays without the 12 voxel-boundary triangles of a non-periodic
    substrate, as ``_fill_mesh`` builds them (disimpy/simulations.py:531-542).

    The reference deletes the wall references one by one and, for every
    deletion at (current) position q, decrements BOTH ends of every subvoxel
    whose end is > q.  That is equivalent to shifting the whole range of
    subvoxel j down by d_j = number of deleted references located before its
    original end, then clamping negatives to 0 (a subvoxel that contained a
    deleted reference therefore also starts one reference early - a quirk of
    the reference that is reproduced here because it decides which candidate
    triangles the fill kernel visits)."""
    vertices = np.copy(substrate.vertices[0:-8])
    faces = np.copy(substrate.faces[0:-12])
    tri = np.asarray(substrate.triangle_indices)
    svi = np.array(substrate.subvoxel_indices, dtype=np.int64)
    deleted = np.where(tri >= len(faces))[0]
    d = np.searchsorted(deleted, svi[:, 1], side="left")
    svi = svi - d[:, np.newaxis]
    svi[svi < 0] = 0
    return vertices, faces, np.delete(tri, deleted), svi


def _fill_mesh(n_points, substrate, intra, seed, cuda_bs=128, device=0, dist=None, walker_range=None, devices=None):
    """Sample points uniformly inside (``intra``) or outside the closed surface of
    a triangular mesh on the GPU (disimpy/simulations.py:505-579).  With an initialised
    process group ``dist`` the launches' threads are split over the ranks
    (sharding.fill_mesh_sharded) and only the rows ``walker_range`` come back; ``devices``: the in-process twin - one
    fill session per listed device on host threads (dsim_fill_mesh_multi).  The rows are the same either way."""
    if substrate.periodic:
        vertices, faces = substrate.vertices, substrate.faces
        tri, svi = substrate.triangle_indices, substrate.subvoxel_indices
    else:
        vertices, faces, tri, svi = _strip_wall_triangles(substrate)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        from . import sharding

        def open_session(t0, t1):
            return _lib.FillSession(t0, t1, intra, seed, vertices, faces, substrate.voxel_size, substrate.xs, substrate.ys,
                                    substrate.zs, substrate.n_sv, svi, tri, device=device)

        return sharding.fill_mesh_sharded(n_points, open_session, dist, device, walker_range)
    return _lib.fill_mesh(
        n_points, intra, seed, vertices, faces, substrate.voxel_size, substrate.xs, substrate.ys, substrate.zs,
        substrate.n_sv, svi, tri, device=device, devices=devices,
    )  # fmt: skip


# ----------------------------------------------------------------------------
# small host utilities (disimpy/simulations.py:1016-1048)
# ----------------------------------------------------------------------------


def add_noise_to_data(data, sigma, seed=None):
    """Add Rician noise to data.

    Parameters
    ----------
    data : numpy.ndarray
    sigma : float
        Standard deviation of noise in each channel.
    seed : int, optional

    Returns
    -------
    noisy_data : numpy.ndarray
    """
    if seed:
        np.random.seed(seed)
    real = np.random.normal(size=data.shape, scale=sigma, loc=0)
    imag = np.random.normal(size=data.shape, scale=sigma, loc=0)
    return np.abs(data + real + 1j * imag)


def _write_traj(traj, mode, positions):
    """One line of ``x y z x y z ...`` (``str(value) + " "`` per coordinate) to the trajectories
    file (disimpy/simulations.py:1043-1048), formatted by the native writer of the C ABI."""
    _lib.write_traj_line(traj, positions, append=(mode == "a"))


def _write_traj_python(traj, mode, positions):
    """The reference's own formulation of ``_write_traj`` - kept as the checker of the native writer
    (tests/test_host_api.py), not used by ``simulation()``."""
    with open(traj, mode) as f:
        f.write(" ".join(map(repr, positions.ravel().tolist())))
        f.write(" \n")


# ----------------------------------------------------------------------------
# the driver
# ----------------------------------------------------------------------------


def _validate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, cuda_bs, max_iter):
    """Argument checks of disimpy/simulations.py:1128-1153 (same messages)."""
    if not isinstance(n_walkers, int) or n_walkers <= 0:
        raise ValueError("Incorrect value (%s) for n_walkers" % n_walkers)
    if not isinstance(diffusivity, float) or diffusivity <= 0:
        raise ValueError("Incorrect value (%s) for diffusivity" % diffusivity)
    if (
        not isinstance(gradient, np.ndarray)
        or gradient.ndim != 3
        or gradient.shape[2] != 3
        or not np.issubdtype(gradient.dtype, np.floating)
    ):
        raise ValueError("Incorrect value (%s) for gradient" % gradient)
    if not isinstance(dt, float) or dt <= 0:
        raise ValueError("Incorrect value (%s) for dt" % dt)
    if not isinstance(substrate, substrates._Substrate):
        raise ValueError("Incorrect value (%s) for substrate" % substrate)
    if not isinstance(seed, int) or seed < 0:
        raise ValueError("Incorrect value (%s) for seed" % seed)
    if traj:
        if not isinstance(traj, str):
            raise ValueError("Incorrect value (%s) for traj" % traj)
    if not isinstance(quiet, bool):
        raise ValueError("Incorrect value (%s) for quiet" % quiet)
    if not isinstance(cuda_bs, int) or cuda_bs <= 0:
        raise ValueError("Incorrect value (%s) for cuda_bs" % cuda_bs)
    if not isinstance(max_iter, int) or max_iter < 1:
        raise ValueError("Incorrect value (%s) for max_iter" % max_iter)


def _initial_positions(n_walkers, substrate, seed, quiet, device, walker_range=None, dist=None, devices=None):
    """Initial positions of the walkers [a, b) = ``walker_range`` (default: all) and the
    substrate's rotation frames.  Returns (positions (b - a, 3) or None for zeros, frames).

    Walker i always starts at row i of the reference's (n_walkers, 3) array, whatever the
    range.  What a shard has to compute for that:

    * analytic substrates (host rejection streams, disimpy/simulations.py:353-418): the
      stream is sequential and acceptance is data dependent, so the first b points are
      drawn (a few tens of ns each) and rows [a, b) kept - nothing beyond b is generated;
    * mesh, ``init_pos="uniform"`` (:1347): the first 3 b draws of the global NumPy stream;
    * mesh, ``init_pos="intra"/"extra"`` (:505-579): every launch of the reference's fill
      kernel appends its accepted points in thread order, so row i depends on the outcome
      of all earlier threads and launches.  Thread t of every launch only ever needs RNG state t,
      though: with an initialised process group the ranks share the fill - each runs the launches
      over its range of the threads, the per-launch accepted counts are exchanged and every rank
      ends up with its rows of the single-process sequence (sharding.fill_mesh_sharded); a single
      process runs the whole fill on one device and keeps rows [a, b)."""
    a, b = (0, n_walkers) if walker_range is None else walker_range
    kind = substrate.type
    if kind == "free":
        return None, None
    if kind == "cylinder":
        R = utils.vec2vec_rotmat(substrate.orientation, np.array([1.0, 0, 0]))
        R_inv = np.linalg.inv(R)
        pos = _initial_positions_cylinder(b, substrate.radius, R_inv, _host_rng(seed))
        return pos[a:b], (R, R_inv)
    if kind == "sphere":
        return _fill_sphere(b, substrate.radius, _host_rng(seed))[a:b], None
    if kind == "ellipsoid":
        R_inv = substrate.R
        R = np.linalg.inv(R_inv)
        pos = _initial_positions_ellipsoid(b, substrate.semiaxes, R_inv, _host_rng(seed))
        return pos[a:b], (R, R_inv)
    if kind == "mesh":
        if isinstance(substrate.init_pos, np.ndarray):
            if n_walkers != substrate.init_pos.shape[0]:
                raise ValueError("n_walkers must be equal to the number of initial positions")
            return substrate.init_pos[a:b], None
        if not quiet:
            print("Calculating initial positions")
        if substrate.init_pos == "uniform":
            pos = (np.random.random((b, 3)) * substrate.voxel_size)[a:b]
        elif dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            # the device fill is shared by the ranks (threads of every launch split, sharding.fill_mesh_sharded)
            pos = _fill_mesh(n_walkers, substrate, substrate.init_pos == "intra", seed, device=device, dist=dist, walker_range=(a, b))
        elif substrate.init_pos == "intra":
            pos = _fill_mesh(n_walkers, substrate, True, seed, device=device, devices=devices)[a:b]
        else:
            pos = _fill_mesh(n_walkers, substrate, False, seed, device=device, devices=devices)[a:b]
        if not quiet:
            print("Finished calculating initial positions")
        return pos, None
    raise ValueError("Incorrect value (%s) for substrate" % substrate)


def _run_range(sim, substrate, frames, max_iter, epsilon, t0, t1):
    kind = substrate.type
    if kind == "free":
        sim.run_free(t0, t1)
    elif kind == "sphere":
        sim.run_sphere(substrate.radius, max_iter, epsilon, t0, t1)
    elif kind == "cylinder":
        sim.run_cylinder(substrate.radius, frames[0], frames[1], max_iter, epsilon, t0, t1)
    elif kind == "ellipsoid":
        sim.run_ellipsoid(substrate.semiaxes, frames[0], frames[1], max_iter, epsilon, t0, t1)
    else:
        sim.run_mesh(max_iter, epsilon, substrate.perm_prob, t0, t1)


_MIN_WALKERS_PER_DEVICE = 65536


def _default_devices(n_gpu):
    """The devices ``simulation()`` spreads its walkers over: every visible GPU, one balanced contiguous shard of
    the walkers each (dsim_create_multi), unless DSIM_DEVICES names a list (e.g. ``0,0``: two shards on device 0)."""
    env = os.environ.get("DSIM_DEVICES")
    if env:
        return [int(x) for x in env.split(",") if x.strip() != ""]
    # a job launched as one process per GPU (torchrun: RANK / LOCAL_RANK / WORLD_SIZE are set) must not drive every GPU
    # from every rank: such a process stays on its own device (sharding.simulation_sharded is the entry for those jobs)
    if any(os.environ.get(k) for k in ("LOCAL_RANK", "RANK", "WORLD_SIZE")):
        return [int(os.environ.get("LOCAL_RANK", "0")) % max(1, n_gpu)]
    return list(range(max(1, n_gpu)))


def _simulate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, max_iter, epsilon, device=0,
              walker_range=None, need_phases=False, need_positions=False, dist=None, devices=None, walker_ids=None,
              positions_all=None, traj_sink=None):
    """Run walkers [a, b) = ``walker_range`` (default: all) on one GPU, or - ``devices`` a list of more than one
    device - cut into shards over several GPUs behind one multi-device handle (include/dsim.h,
    dsim_create_multi: the shards run concurrently, per-walker outputs come back in order; a mesh run cuts them by
    space).  ``walker_ids`` (ascending) instead of a range: this process's part of a spatial cut of ALL walkers
    (sharding.simulation_sharded), ``positions_all`` then holds the initial positions of all of them.
    ``traj_sink(t0, t1, rows)``: called once with ``t0 = -1`` and this shard's (n_local, 3) initial positions, then per
    range of recorded steps with its positions (t1 - t0, n_local, 3), instead of writing the trajectories file here (the
    ranks of a sharded job gather their columns).

    Returns a dict with the device-reduced signal sums and, on request,
    per-walker outputs of this range.  Walker i always gets RNG state i and
    initial position i, so any partition of the walkers gives the same
    per-walker results."""
    n_t = gradient.shape[1]
    a, b = (0, n_walkers) if walker_range is None else walker_range
    if walker_ids is not None:
        walker_ids = np.ascontiguousarray(walker_ids, dtype=np.int64)
        a, b = 0, len(walker_ids)
    step_l = np.sqrt(6 * diffusivity * dt)
    if not quiet:
        print("Number of random walkers = %s" % n_walkers)
        print("Number of steps = %s" % n_t)
        print("Step length = %s m" % step_l)
        print("Step duration = %s s" % dt)
    if devices is not None and len(devices) > 1 and not os.environ.get("DSIM_DEVICES"):
        # a shard below ~65 000 walkers does not fill a GPU: a small run pays handle creations and mesh uploads for nothing
        devices = list(devices)[: max(1, (b - a) // _MIN_WALKERS_PER_DEVICE)]
    if devices is not None and len(devices) >= 1:
        device = int(devices[0])
    if walker_ids is not None:
        if substrate.type != "mesh" or positions_all is None:
            raise ValueError("walker_ids: a spatial cut of a mesh substrate (needs positions_all)")
        frames = None
        positions = np.ascontiguousarray(np.asarray(positions_all)[walker_ids], dtype=np.float64)
    else:
        positions, frames = _initial_positions(n_walkers, substrate, seed, quiet, device, (a, b), dist=dist, devices=devices)
    with _lib.Sim(b - a, gradient, dt, float(step_l), device=device, first_walker=a, devices=devices) as sim:
        if walker_ids is not None:
            sim.set_walker_ids(walker_ids)
        sim.rng_init_device(seed)
        if substrate.type == "mesh":  # before the positions: a multi-device handle cuts its walkers by the cost the index implies
            sim.set_mesh(substrate.vertices, substrate.faces, substrate.xs, substrate.ys, substrate.zs,
                         substrate.n_sv, substrate.subvoxel_indices, substrate.triangle_indices)
        if positions is not None:
            pos = np.ascontiguousarray(positions, dtype=np.float64)
            if substrate.type == "mesh":
                # several devices: every shard gets a REGION of the substrate (the Morton order of the initial positions
                # cut into parts of equal estimated cost), not a range of walker indices - the outputs are the same bytes
                sim.partition_by_position(pos)
            else:
                sim.set_positions(pos)
        if traj:  # the start line: the positions the device got (never computed a second time)
            start = np.zeros((b - a, 3)) if positions is None else np.asarray(positions, dtype=np.float64)
            if traj_sink is None:
                _write_traj(traj, "w", start)
            else:
                traj_sink(-1, 0, start)
        if traj:
            # the trajectories file needs the positions after every step: the device records them
            # and hands them over once per range of steps (<= 64 MB) instead of once per step
            sim.set_recording(True)
            # <= 8 GB of recorded positions per range (a mesh run re-sorts its walkers inside a range like any other mesh
            # run - the device records through the slot -> walker map - so the ranges need not be short)
            chunk = max(1, min(n_t, (1 << 33) // (24 * max(1, b - a))))
            if traj_sink is not None:  # the ranks gather whole lines: <= 1 GB of all walkers' positions per range
                chunk = max(1, min(n_t, (1 << 30) // (24 * max(1, n_walkers))))
        elif quiet:
            chunk = n_t
        else:
            chunk = max(1, int(math.ceil(n_t / 100)))  # progress is reported at range boundaries
            if substrate.type == "mesh":
                # every run_mesh call re-sorts the walkers by position and restores their order at
                # the end (about 20 time steps' worth of work): never cut a mesh run finer than the
                # library's own re-sort interval
                chunk = max(chunk, 1024)
        ms_total = 0.0
        for t0 in range(0, n_t, chunk):
            t1 = min(n_t, t0 + chunk)
            _run_range(sim, substrate, frames, max_iter, epsilon, t0, t1)
            ms_total += sim.last_run_ms()[0]
            if traj and traj_sink is not None:
                traj_sink(t0, t1, sim.recorded(t0, t1))
            elif traj:
                sim.write_traj(traj, t0, t1)
            if not quiet:
                sys.stdout.write(f"\r{np.round((t0 / n_t) * 100, 1)}%")
                sys.stdout.flush()
        out = {"device_ms": ms_total, "range": (a, b)}
        out["limbs"], out["n_exceeded"] = sim.signal_limbs()  # exact integer sums: limbs of shards / ranks add
        out["cos"], out["sin"] = _lib.signal_from_limbs(out["limbs"])
        out["iter_exc"] = sim.iter_exc() if out["n_exceeded"] else np.zeros(b - a, dtype=bool)
        if need_phases:  # all_signals: real(exp(1j * phases)) with the NaN mask, evaluated on the device (:1414-1417)
            out["all_signals"] = sim.all_signals()
        if need_positions:
            out["positions"] = sim.positions()
    return out


def simulation(
    n_walkers,
    diffusivity,
    gradient,
    dt,
    substrate,
    seed=123,
    traj=None,
    final_pos=False,
    all_signals=False,
    quiet=False,
    cuda_bs=128,
    max_iter=int(1e3),
    epsilon=1e-13,
):
    """Simulate a diffusion-weighted MR experiment and generate signal.

    Same contract as ``disimpy.simulations.simulation``.

    Parameters
    ----------
    n_walkers : int
        Number of random walkers.
    diffusivity : float
        Diffusivity in SI units (m^2/s).
    gradient : numpy.ndarray
        Floating-point array of shape (number of measurements, number of time
        points, 3), in T/m.
    dt : float
        Duration of a time step in the gradient array (s).
    substrate : disimpy_amd.substrates._Substrate
    seed : int, optional
        Seed for pseudorandom number generation.
    traj : str, optional
        Path of a file in which to save the walker trajectories (one line per
        time point: walker_1_x walker_1_y walker_1_z walker_2_x ...).
    final_pos : bool, optional
        If True, return ``(signal, final positions)``.
    all_signals : bool, optional
        If True, return the signal of every walker, shape (measurements, walkers).
    quiet : bool, optional
    cuda_bs : int, optional
        Accepted for compatibility.  (In the reference it only changes how many
        RNG states are created, never which state a walker uses.)
    max_iter : int, optional
        Maximum number of reflections checked per walker and time step.
    epsilon : float, optional
        Distance by which a walker is moved off a surface after a collision.

    Returns
    -------
    signal : numpy.ndarray
        Sum over walkers of cos(phase) per measurement (not divided by n_walkers).
    """
    try:
        n_gpu = _lib.device_count()
    except Exception as e:  # library missing / not loadable
        raise Exception(_NO_GPU_MESSAGE + " (%s)" % e)
    if n_gpu <= 0:
        raise Exception(_NO_GPU_MESSAGE)
    _validate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, cuda_bs, max_iter)
    if not quiet:
        print("Starting simulation")
        if traj:
            print("The trajectories file will be up to %s GB" % (gradient.shape[1] * n_walkers * 3 * 25 / 1e9))
    np.random.seed(seed)  # the reference seeds the global NumPy generator too (:1169)
    # the reference runs on one GPU (:1051); here the walkers are spread over every visible GPU behind one handle
    res = _simulate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, max_iter, epsilon,
                    need_phases=bool(all_signals), need_positions=bool(final_pos), devices=_default_devices(n_gpu))
    iter_exc = res["iter_exc"]
    if np.any(iter_exc):
        warnings.warn(
            "Maximum number of iterations was exceeded in the intersection "
            + "check algorithm for walkers %s" % np.where(iter_exc)[0]
        )
    if all_signals:
        signals = res["all_signals"]
    else:
        signals = res["cos"]
    if not quiet:
        sys.stdout.write("\rSimulation finished\n")
        sys.stdout.flush()
    if final_pos:
        return signals, res["positions"]
    return signals
