"""Multi-process path on CPU: world_size-2 `gloo` group.  The shard partition,
the global-index RNG convention and the signal all-reduce are exercised with
the CPU oracle standing in for the GPU kernels (tests may use the oracle as the
compute engine; the product never does)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_all_walkers():
    from disimpy_amd.sharding import shard_range

    for n in (1, 7, 1000, 10**7 + 3):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a0, b0), (a1, b1) in zip(ranges[:-1], ranges[1:]):
                assert b0 == a1 and a0 <= b0
            per = -(-n // world)
            assert all(b - a <= per for a, b in ranges)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disimpy_amd import gradients, sharding
    from oracle.oracle import FREE, Oracle

    gradient, dt = gradients.pgse(10e-3, 40e-3, 60, np.array([1e9, 5e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    o = Oracle(exact=False)
    a, b = sharding.shard_range(n, rank, world)
    st = o.rng_states(b, 123)[a:].copy()  # walker i <- RNG state i, whatever the partition
    pos = np.zeros((b - a, 3))
    ph, exc, _ = o.run(o.geom(FREE), gradient, dt, step_l, pos, st)
    c, s = o.signal(ph, exc)
    tc, ts, nexc = sharding.reduce_signal(c, s, int(exc.sum()), dist)
    q.put((rank, tc, ts, nexc))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduce_matches_single_process():
    from disimpy_amd import gradients
    from oracle.oracle import FREE, Oracle

    n, world = 1001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gradient, dt = gradients.pgse(10e-3, 40e-3, 60, np.array([1e9, 5e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    o = Oracle(exact=False)
    st = o.rng_states(n, 123)
    pos = np.zeros((n, 3))
    ph, exc, _ = o.run(o.geom(FREE), gradient, dt, step_l, pos, st)
    c, s = o.signal(ph, exc)
    for rank, tc, ts, nexc in got:
        np.testing.assert_allclose(tc, c, rtol=0, atol=1e-12 * n)  # summation order only
        np.testing.assert_allclose(ts, s, rtol=0, atol=1e-12 * n)
        assert nexc == 0
    np.testing.assert_array_equal(got[0][1], got[1][1])  # every rank holds the same total


def _limbs(ph, exc):
    """numpy restatement of the device's exact signal sums (dsim_kernels.h, signal_partial_kernel): every cos / sin
    term rounded to a multiple of 2^-60, summed as integers in two limbs of 30 bits"""
    out = []
    keep = ~np.asarray(exc, dtype=bool)
    for m in range(len(ph)):
        for f in (np.cos, np.sin):
            q = [int(v) for v in np.rint(f(ph[m][keep]) * 2.0**60)]
            out += [sum(v >> 30 for v in q), sum(v & (2**30 - 1) for v in q)]
    return np.array(out, dtype=np.int64)


def _worker_empty(rank, world, port, n, q):
    """More ranks than walkers: the rank with an empty range goes through simulation_sharded() itself (it must
    not touch the GPU and must still take part in the collective), the others stand in with the oracle."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disimpy_amd import gradients, sharding, substrates
    from oracle.oracle import FREE, Oracle

    gradient, dt = gradients.pgse(10e-3, 40e-3, 30, np.array([1e9, 5e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    a, b = sharding.shard_range(n, rank, world)
    if b == a:
        tc = sharding.simulation_sharded(n, 2e-9, gradient, dt, substrates.free(), seed=123, dist=dist)
    else:
        step_l = float(np.sqrt(6 * 2e-9 * dt))
        o = Oracle(exact=False)
        st = o.rng_states(b, 123)[a:].copy()
        pos = np.zeros((b - a, 3))
        ph, exc, _ = o.run(o.geom(FREE), gradient, dt, step_l, pos, st)
        tc, _, _ = sharding.reduce_signal_exact(_limbs(ph, exc), 0, dist)
    q.put((rank, (a, b), tc))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_with_an_empty_shard_still_joins_the_reduction():
    n, world = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_empty, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [(0, 1), (1, 2), (2, 2)]
    for g in got[1:]:
        np.testing.assert_array_equal(g[2], got[0][2])
    assert np.all(np.abs(got[0][2]) <= n)


class _FakeFill:
    """Stand-in for disimpy_amd._lib.FillSession on the CPU: thread t of launch l accepts with probability p, decided and
    placed by a hash of (t, l) - the same for every split of the threads, as on the device (thread t <- RNG state t)."""

    def __init__(self, t0, t1, p):
        self.t, self.launch, self.p = np.arange(t0, t1, dtype=np.uint64), 0, p

    def _draw(self, launch):
        h = (self.t * np.uint64(0x9E3779B97F4A7C15) + np.uint64(launch) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFF)
        h = (h ^ (h >> np.uint64(21))) * np.uint64(0x94D049BB133111EB) & np.uint64(0xFFFFFFFFFFFF)
        u = (h >> np.uint64(8)).astype(np.float64) / float(1 << 40)
        ok = u < self.p
        pts = np.stack([self.t.astype(np.float64), np.full(len(self.t), float(launch)), u], axis=1)
        return ok, pts

    def run(self, n_launches, capacity):
        counts, out = np.zeros(n_launches, dtype=np.int64), []
        for k in range(n_launches):
            ok, pts = self._draw(self.launch)
            self.launch += 1
            counts[k] = ok.sum()
            out.append(pts[ok])
        out = np.concatenate(out) if out else np.zeros((0, 3))
        return counts, out[:capacity]

    def close(self):
        pass


def _worker_fill(rank, world, port, n, p, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disimpy_amd import sharding

    rows = sharding.fill_mesh_sharded(n, lambda t0, t1: _FakeFill(t0, t1, p), dist)
    q.put((rank, sharding.shard_range(n, rank, world), rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,p", [(2, 5001, 0.03), (3, 4000, 0.4), (3, 2, 0.5)])
def test_sharded_fill_reproduces_the_single_process_sequence(world, n, p):
    """the threads of every launch split over the ranks, only per-launch counts exchanged: same rows, same order"""
    from disimpy_amd import sharding

    want = sharding.fill_mesh_sharded(n, lambda t0, t1: _FakeFill(t0, t1, p))  # one process, all threads
    # the single sequence itself: accepted candidates launch by launch in thread order
    seq, f = [], _FakeFill(0, n, p)
    while sum(len(x) for x in seq) < n:
        seq.append(f.run(1, n)[1])
    np.testing.assert_array_equal(want, np.concatenate(seq)[:n])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fill, args=(r, world, port, n, p, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, (a, b), rows in got:
        np.testing.assert_array_equal(rows, want[a:b])


class _OverflowingFill(_FakeFill):
    """a session whose buffer is too small on ONE rank: more points accepted than fit `capacity` while more are needed"""

    def run(self, n_launches, capacity):
        counts, pts = super().run(n_launches, capacity)
        return counts, pts[: max(0, min(len(pts), 3))]


def _worker_fill_fails(rank, world, port, n, mode, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disimpy_amd import sharding

    class _Raising(_FakeFill):
        def run(self, n_launches, capacity):
            raise RuntimeError("device lost")

    def open_session(t0, t1):
        if rank == 1:
            return _OverflowingFill(t0, t1, 0.5) if mode == "overflow" else _Raising(t0, t1, 0.5)
        return _FakeFill(t0, t1, 0.5)

    try:
        sharding.fill_mesh_sharded(n, open_session, dist)
        q.put((rank, "returned"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()  # every rank is still alive and in step: nobody was left behind in a collective
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["overflow", "raises"])
def test_a_failing_rank_ends_the_sharded_fill_on_every_rank(mode):
    """the overflow error was raised on one rank only, before the all-gather - the others hung in it"""
    world, n = 2, 4000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fill_fails, args=(r, world, port, n, mode, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all("rank 1 failed" in msg for msg in got.values()), got
    assert ("overflow" in got[1]) if mode == "overflow" else ("device lost" in got[1])


def _oracle_simulate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, max_iter, epsilon, device=0,
                     walker_range=None, need_phases=False, need_positions=False, dist=None, devices=None, walker_ids=None,
                     positions_all=None, traj_sink=None):
    """stand-in for simulations._simulate with the CPU oracle as the engine (sphere, or a mesh with given initial
    positions; walker i <- state i, position i - by range or, ``walker_ids``, by a list of walkers)"""
    from disimpy_amd import simulations
    from oracle.oracle import MESH, SPHERE, Oracle

    o = Oracle(exact=True)
    step_l = float(np.sqrt(6 * diffusivity * dt))
    ids = np.arange(*walker_range) if walker_ids is None else np.asarray(walker_ids)
    if substrate.type == "sphere":
        g = o.geom(SPHERE, radius=substrate.radius, max_iter=max_iter, epsilon=epsilon)
        pos = np.ascontiguousarray(simulations._fill_sphere(n_walkers, substrate.radius, simulations._host_rng(seed))[ids])
    else:
        ma = dict(vertices=substrate.vertices, faces=substrate.faces, xs=substrate.xs, ys=substrate.ys, zs=substrate.zs, n_sv=substrate.n_sv,
                  subvoxel_indices=substrate.subvoxel_indices, triangle_indices=substrate.triangle_indices)
        g = o.geom(MESH, mesh=ma, max_iter=max_iter, epsilon=epsilon)
        start = positions_all
        if start is None:  # like the real _simulate: given rows, or (init_pos="uniform") draws of the global NumPy stream
            start = substrate.init_pos if isinstance(substrate.init_pos, np.ndarray) else \
                simulations._initial_positions(n_walkers, substrate, seed, True, 0, (0, n_walkers))[0]
        pos = np.ascontiguousarray(np.asarray(start)[ids], dtype=np.float64)
    st = o.rng_states(n_walkers, seed)[ids].copy()
    pos0 = pos.copy()
    ph, exc, tr = o.run(g, gradient, dt, step_l, pos, st, want_traj=bool(traj))
    if traj and traj_sink is not None:
        traj_sink(-1, 0, pos0)  # the start line: the positions this run began with
        traj_sink(0, gradient.shape[1], tr)
    elif traj:
        simulations._write_traj_python(traj, "w", pos0)
        for row in tr:
            simulations._write_traj_python(traj, "a", row)
    c, s = o.signal(ph, exc)
    sig = np.cos(ph)
    sig[:, exc.astype(bool)] = np.nan
    # the exact integer sums the device returns (dsim_get_signal_limbs): terms rounded to multiples of 2^-60, q = hi * 2^30 + lo
    limbs = np.zeros(4 * ph.shape[0], dtype=np.int64)
    keep = ~exc.astype(bool)
    for m in range(ph.shape[0]):
        for k, f in enumerate((np.cos, np.sin)):
            terms = f(ph[m][keep & ~np.isnan(ph[m])])
            q = sum(int(v) for v in np.rint(terms * 2.0**60))
            limbs[4 * m + 2 * k], limbs[4 * m + 2 * k + 1] = q >> 30, q & ((1 << 30) - 1)
    return {"cos": c, "sin": s, "n_exceeded": int(exc.sum()), "iter_exc": exc.astype(bool), "all_signals": sig, "positions": pos,
            "limbs": limbs}


def _worker_outputs(rank, world, port, n, max_iter, q):
    import warnings

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disimpy_amd import gradients, sharding, simulations, substrates

    simulations._simulate = _oracle_simulate
    gradient, dt = gradients.pgse(10e-3, 40e-3, 40, np.array([1e9, 5e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sig, pos = sharding.simulation_sharded(n, 2e-9, gradient, dt, substrates.sphere(2e-6), seed=7, dist=dist, max_iter=max_iter,
                                               final_pos=True, all_signals=True)
        total = sharding.simulation_sharded(n, 2e-9, gradient, dt, substrates.sphere(2e-6), seed=7, dist=dist, max_iter=max_iter)
    q.put((rank, sig, pos, total, [str(x.message) for x in w]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,max_iter", [(2, 301, 1000), (3, 100, 1)])
def test_sharded_simulation_returns_every_per_walker_output(world, n, max_iter):
    """final_pos, all_signals and the iteration-limit warning of simulation() (disimpy/simulations.py:1406-1429) from a
    job of one process per GPU: per-walker outputs gathered in walker order, the warning with global indices"""
    from disimpy_amd import gradients, substrates

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_outputs, args=(r, world, port, n, max_iter, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    gradient, dt = gradients.pgse(10e-3, 40e-3, 40, np.array([1e9, 5e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    one = _oracle_simulate(n, 2e-9, gradient, dt, substrates.sphere(2e-6), 7, None, True, max_iter, 1e-13, walker_range=(0, n))
    for rank, sig, pos, total, warned in got:
        np.testing.assert_array_equal(sig, one["all_signals"])
        np.testing.assert_array_equal(pos, one["positions"])
        np.testing.assert_allclose(total, one["cos"], rtol=0, atol=1e-12 * n)
        if one["n_exceeded"]:
            want = "check algorithm for walkers %s" % np.where(one["iter_exc"])[0]
            assert len(warned) == 2 and all(want in m for m in warned), warned
        else:
            assert not warned
    assert max_iter > 1 or one["n_exceeded"] > 0  # the max_iter = 1 case does exercise the warning


def _small_mesh_substrate(n):
    """a closed sphere mesh with n given initial positions inside it (no device fill needed)"""
    from disimpy_amd import substrates

    m = np.load(os.path.join(ROOT, "tests", "golden", "sphere_mesh.npz"))
    rng = np.random.RandomState(11)
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, n_sv=np.array([4, 4, 4]), quiet=True)
    centre, radius = 0.5 * np.asarray(sub.voxel_size), 0.4 * (np.min(sub.voxel_size) - 2e-6)
    d = rng.standard_normal((n, 3))
    d *= (radius * rng.uniform(0, 1, (n, 1)) ** (1 / 3)) / np.linalg.norm(d, axis=1)[:, None]
    sub.init_pos = centre + d
    return sub


def _worker_traj(rank, world, port, n, kind, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disimpy_amd import gradients, sharding, simulations, substrates

    simulations._simulate = _oracle_simulate
    gradient, dt = gradients.pgse(10e-3, 40e-3, 25, np.array([1e9]), np.array([[1.0, 0, 0]]))
    sub = substrates.sphere(2e-6) if kind == "sphere" else _small_mesh_substrate(n)
    sig, pos = sharding.simulation_sharded(n, 2e-9, gradient, dt, sub, seed=7, dist=dist, final_pos=True, all_signals=True, traj=path)
    q.put((rank, sig, pos))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,kind", [(2, 57, "sphere"), (3, 64, "mesh"), (3, 2, "sphere")])
def test_sharded_simulation_writes_the_trajectories_file_and_cuts_a_mesh_by_space(world, n, kind, tmp_path):
    """traj= of a job of one process per GPU (disimpy/simulations.py:1043-1048, :1194, :1211-1213): rank 0 writes the
    single-process file, the ranks' columns gathered per range of steps; on a mesh the ranks own REGIONS (the Morton
    order of the initial positions in equal parts, dsim_partition_walkers) and every per-walker output still comes back
    in walker order"""
    from disimpy_amd import gradients, substrates

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    path = str(tmp_path / "traj_sharded.txt")
    procs = [ctx.Process(target=_worker_traj, args=(r, world, port, n, kind, path, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    gradient, dt = gradients.pgse(10e-3, 40e-3, 25, np.array([1e9]), np.array([[1.0, 0, 0]]))
    sub = substrates.sphere(2e-6) if kind == "sphere" else _small_mesh_substrate(n)
    want_path = str(tmp_path / "traj_one.txt")
    one = _oracle_simulate(n, 2e-9, gradient, dt, sub, 7, want_path, True, 1000, 1e-13, walker_range=(0, n))
    for rank, sig, pos in got:
        np.testing.assert_array_equal(sig, one["all_signals"])
        np.testing.assert_array_equal(pos, one["positions"])
    assert open(path, "rb").read() == open(want_path, "rb").read()
    assert len(open(path).read().splitlines()) == 26


def test_one_process_writes_the_positions_the_walkers_started_from(tmp_path, monkeypatch):
    """without a process group (world = 1) simulation_sharded(traj=...) takes the single-process writer,
    and the start line holds the positions the walkers were given - on a mesh with init_pos="uniform" those are draws of
    the global NumPy stream (disimpy/simulations.py:1347), which a second look would find advanced"""
    from disimpy_amd import gradients, sharding, simulations, substrates

    calls = []
    real = simulations._initial_positions

    def counted(*a, **k):
        calls.append(1)
        return real(*a, **k)

    monkeypatch.setattr(simulations, "_simulate", _oracle_simulate)
    monkeypatch.setattr(simulations, "_initial_positions", counted)
    m = np.load(os.path.join(ROOT, "tests", "golden", "sphere_mesh.npz"))
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="uniform", n_sv=np.array([4, 4, 4]), quiet=True)
    gradient, dt = gradients.pgse(10e-3, 40e-3, 25, np.array([1e9]), np.array([[1.0, 0, 0]]))
    n, seed = 6, 7
    path = str(tmp_path / "traj.txt")
    _, pos = sharding.simulation_sharded(n, 2e-9, gradient, dt, sub, seed=seed, dist=None, final_pos=True, traj=path)
    assert len(calls) == 1  # the initial positions are worked out once
    lines = [np.array(l.split(), dtype=np.float64).reshape(n, 3) for l in open(path).read().splitlines()]
    assert len(lines) == 26
    np.random.seed(seed)
    np.testing.assert_array_equal(lines[0], np.random.random((n, 3)) * sub.voxel_size)
    assert not np.array_equal(lines[1], lines[0])
    np.testing.assert_array_equal(lines[-1], pos)
