"""GPU parity at production shape (run with ``-m gpu``): the combinations the headline bench
actually exercises - the neuron with automatic Morton re-sorting over several re-sort intervals
and launches of the 4-waves/SIMD cooperative kernel, the packed-cylinder mesh of BASELINE config
C4, and the 50 000-triangle refinement (C5b) - each bit-identical, walker by walker, to the
exact CPU oracle (OpenMP over walkers; reference: disimpy/simulations.py:878-1013).
"""
import os
import sys

import numpy as np
import numpy.testing as npt
import pytest

from conftest import ROOT, golden
from oracle.oracle import FREE, MESH, SPHERE
from test_gpu_parity import D, _assert_identical, _gpu_run, _mesh_arrays, _random_gradient

sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
THREADS = os.cpu_count() or 1


@pytest.fixture(scope="module")
def lib():
    from disimpy_amd import _lib

    assert _lib.device_count() > 0, "these tests need an AMD GPU"
    return _lib


def _oracle_mesh(o, ma, n, g, dt, step_l, pos0, seed, perm_prob=0.0, want_traj=False):
    st = o.rng_states(n, seed)
    pos = np.array(pos0, dtype=np.float64, order="C")
    geom = o.geom(MESH, mesh=ma, max_iter=1000, epsilon=1e-13, perm_prob=perm_prob)
    ph, exc, traj = o.run(geom, g, dt, step_l, pos, st, n_threads=THREADS, want_traj=want_traj)
    c, s = o.signal(ph, exc)
    return dict(pos=pos, phases=ph, exc=exc.astype(bool), states=np.stack([st["s0"], st["s1"]], axis=1), cos=c, sin=s,
                traj=traj)


def test_neuron_production_shape_resorted_launches(lib, oracle_exact, monkeypatch):
    """Neuron (29 688 triangles, 50^3 index, walkers inside), the C5 step length, 8192 walkers x
    2200 steps: automatic re-sort interval (1024 steps => 3 sorts, 3 launches of
    walk_kernel_coop4, XCD-aware group mapping, walkers advancing at their own pace)."""
    from disimpy_amd import simulations, substrates

    m = golden("neuron_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
    n, T, dt = 8192, 2200, 7e-6
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(1, T)
    pos0 = simulations._fill_mesh(n, sub, True, 123)
    ma = _mesh_arrays(sub)
    ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 123)
    kw = dict(max_iter=1000, epsilon=1e-13)
    # default tuning = what bench.py runs: per-walker clocks (a launch is a budget of passes per wavefront; walkers
    # reach step 2200 in different launches); mesh_traversal 5 = every launch covers the same steps for all walkers;
    # and the default again in two calls (all walkers are synchronised at every call boundary); then the same with
    # the shard cut into two halves that are sorted and launched independently on two streams (what the library
    # does by itself from 2^18 walkers on; DSIM_GROUPS=2 asks for it here)
    for tuning, chunks, groups in ((None, [0, T], 1), (dict(mesh_traversal=5), [0, T], 1), (None, [0, 1500, T], 1),
                                   (None, [0, T], 2), (None, [0, 700, T], 2)):
        monkeypatch.setenv("DSIM_GROUPS", str(groups))
        with lib.Sim(n, g, dt, step_l) as sim:
            if tuning:
                sim.set_tuning(**tuning)
            sim.rng_init_device(123)
            sim.set_positions(pos0)
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
            for t0, t1 in zip(chunks[:-1], chunks[1:]):
                sim.run_mesh(1000, 1e-13, 0.0, t0, t1)
            if len(chunks) == 2:
                assert sim.last_run_ms()[1] >= 3 and sim.last_walk_ms()[1] >= 3  # >= 2 re-sort intervals, >= 2 launches
            busy, streams = sim.last_walk_busy_ms()
            assert streams == (2 if groups == 2 and not tuning else 1)
            assert 0 < busy <= sim.last_walk_ms()[0] * 1.001
            c, s, _ = sim.signal()
            gpu = dict(pos=sim.positions(), phases=sim.phases(), exc=sim.iter_exc(), states=sim.rng_states(), cos=c, sin=s)
        _assert_identical(gpu, ref)
    assert (gpu["pos"].max(axis=0) < sub.voxel_size).all() and (gpu["pos"].min(axis=0) > 0).all()
    del kw


def test_neuron_shard_large_enough_for_two_groups_by_default(lib, oracle_exact, monkeypatch):
    """270 000 walkers x 400 steps in the neuron with nothing forced: from 2^18 walkers on the library cuts the
    shard into two halves on two streams by itself, and a launch this small re-sorts every 256 steps."""
    from disimpy_amd import simulations, substrates

    monkeypatch.delenv("DSIM_GROUPS", raising=False)
    m = golden("neuron_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
    n, T, dt = 270_000, 400, 7e-6
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(1, T)
    pos0 = simulations._fill_mesh(n, sub, True, 123)
    ma = _mesh_arrays(sub)
    ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 123)
    with lib.Sim(n, g, dt, step_l) as sim:
        sim.rng_init_device(123)
        sim.set_positions(pos0)
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
        sim.run_mesh(1000, 1e-13, 0.0, 0, T)
        assert sim.last_walk_busy_ms()[1] == 2
        assert sim.last_walk_ms()[1] >= 4  # two groups x (a budget worth 256 steps + the rest)
        c, s, _ = sim.signal()
        gpu = dict(pos=sim.positions(), phases=sim.phases(), exc=sim.iter_exc(), states=sim.rng_states(), cos=c, sin=s)
    _assert_identical(gpu, ref)


def test_two_walker_groups_with_ragged_halves_and_permeable_walls(lib, oracle_exact, monkeypatch):
    """Two independently sorted halves (DSIM_GROUPS=2) of an odd number of walkers - 9001 = 4608 + 4393 - in the
    closed cylinder mesh with permeable walls (crossings draw an extra uniform), with one to four measurements
    (the rows of the phase matrix of a half keep the handle's row stride)."""
    from disimpy_amd import substrates

    m = golden("cylinder_mesh_closed.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, n_sv=np.array([10, 10, 10]), perm_prob=0.35, quiet=True)
    n, T, dt = 9001, 700, 1e-3
    step_l = float(np.sqrt(6 * D * dt))
    pos0 = np.random.RandomState(5).random_sample((n, 3)) * sub.voxel_size
    ma = _mesh_arrays(sub)
    monkeypatch.setenv("DSIM_GROUPS", "2")
    for M in (1, 2, 3, 4):  # up to four phase accumulators ride in the cooperative kernel's registers
        g = _random_gradient(M, T)
        ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 11, perm_prob=0.35)
        with lib.Sim(n, g, dt, step_l) as sim:
            sim.set_tuning(resort_steps=128)
            sim.rng_init_device(11)
            sim.set_positions(pos0)
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
            sim.run_mesh(1000, 1e-13, 0.35, 0, T)
            assert sim.last_walk_busy_ms()[1] == 2
            c, s, _ = sim.signal()
            gpu = dict(pos=sim.positions(), phases=sim.phases(), exc=sim.iter_exc(), states=sim.rng_states(), cos=c, sin=s)
        _assert_identical(gpu, ref)


@pytest.mark.parametrize("groups", [1, 2, 3])
def test_many_measurements_on_a_mesh_through_the_group_driver(lib, oracle_exact, monkeypatch, groups):
    """M = 5 and M = 40 on a mesh with re-sorted walkers: positions recorded per range of steps, phases added up from
    them (phase_from_rows_kernel: walker-major rows staged through LDS), one / two / three walker groups on their own streams with ragged sizes, two calls."""
    from disimpy_amd import substrates

    m = golden("cylinder_mesh_closed.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, n_sv=np.array([10, 10, 10]), perm_prob=0.2, quiet=True)
    n, T, dt = 12289, 500, 1e-3
    step_l = float(np.sqrt(6 * D * dt))
    pos0 = np.random.RandomState(6).random_sample((n, 3)) * sub.voxel_size
    ma = _mesh_arrays(sub)
    monkeypatch.setenv("DSIM_GROUPS", str(groups))
    for M in (5, 40):
        g = _random_gradient(M, T)
        ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 13, perm_prob=0.2)
        with lib.Sim(n, g, dt, step_l) as sim:
            sim.set_tuning(resort_steps=96)
            sim.rng_init_device(13)
            sim.set_positions(pos0)
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
            sim.run_mesh(1000, 1e-13, 0.2, 0, 230)
            sim.run_mesh(1000, 1e-13, 0.2, 230, T)
            assert sim.last_walk_busy_ms()[1] == groups
            c, s, _ = sim.signal()
            gpu = dict(pos=sim.positions(), phases=sim.phases(), exc=sim.iter_exc(), states=sim.rng_states(), cos=c, sin=s)
        _assert_identical(gpu, ref)


def test_c4_packed_cylinders_bit_identical(lib, oracle_exact):
    """BASELINE config C4 (bench.packed_cylinders: 4096 triangles, periodic, uniform positions,
    50^3 index), 4096 walkers x 500 steps of the C4 waveform's step length, re-sorted every 128 steps."""
    from disimpy_amd import substrates

    v, f = bench.packed_cylinders()
    sub = substrates.mesh(v, f, True, init_pos="uniform", quiet=True)
    n, T, dt = 4096, 500, 7e-6
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(2, T)
    pos0 = np.random.RandomState(123).random_sample((n, 3)) * sub.voxel_size
    ma = _mesh_arrays(sub)
    ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 123)
    kw = dict(max_iter=1000, epsilon=1e-13)
    # default tuning; frequent re-sorts; the clearance map forced on (walkers farther than a step from every
    # triangle skip the traversal: every triangle of this mesh spans the whole periodic z axis)
    for tuning in (None, dict(resort_steps=128), dict(mesh_traversal=4), dict(mesh_traversal=3)):
        gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 123, kw, mesh=ma, tuning=tuning)
        _assert_identical(gpu, ref)
    # walkers that start inside a cylinder stay inside it (mesh() moved the lattice: take the axes from
    # the vertices the substrate holds; positions are wrapped into the periodic voxel first)
    centres = sub.vertices.reshape(64, 64, 3)[:, :, :2].mean(axis=1)

    def axis_distance(p):
        q = np.mod(p, sub.voxel_size)[:, None, :2] - centres[None, :, :]
        return np.linalg.norm(q, axis=2).min(axis=1)

    was_in = axis_distance(pos0) < 1e-6 * np.cos(np.pi / 32)  # inscribed circle of the 32-gon
    assert was_in.sum() > n // 3
    assert (axis_distance(gpu["pos"][was_in]) < 1e-6).all()


def test_c5b_refined_neuron_bit_identical(lib, oracle_exact):
    """The 50 000-triangle refinement of the neuron that bench.py runs by default (C5b), small size."""
    from disimpy_amd import simulations, substrates

    m = golden("neuron_mesh.npz")
    v, f = bench.split_longest_edges(m["vertices"], m["faces"], 50000)
    assert len(f) == 50000
    sub = substrates.mesh(v, f, True, init_pos="intra", quiet=True)
    n, T, dt = 4096, 600, 7e-6
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(1, T)
    pos0 = simulations._fill_mesh(n, sub, True, 123)
    ma = _mesh_arrays(sub)
    ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 123)
    kw = dict(max_iter=1000, epsilon=1e-13)
    for tuning in (dict(resort_steps=256), dict(resort_steps=256, mesh_traversal=4)):
        gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 123, kw, mesh=ma, tuning=tuning)
        _assert_identical(gpu, ref)


@pytest.mark.parametrize("periodic,init", [(True, "uniform"), (False, "uniform"), (True, "extra")])
def test_clearance_map_is_invisible_in_open_space(lib, oracle_exact, periodic, init):
    """A small sphere in a roomy voxel, walkers everywhere / outside: most walkers are farther than a step
    from every triangle, the launches start with > 40 % of them clear and the kernel skips their traversal
    (default tuning decides that per launch; 4 forces it, 3 forbids it).  Per-walker results must not move."""
    from disimpy_amd import simulations, substrates

    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], periodic, padding=np.zeros(3) + 6e-6, n_sv=np.array([12, 10, 14]),
                          init_pos=init, quiet=True)
    n, T, dt = 6000, 300, 2e-4
    step_l = float(np.sqrt(6 * D * dt))  # 1.5 um
    g = _random_gradient(2, T)
    if init == "uniform":
        pos0 = np.random.RandomState(3).random_sample((n, 3)) * sub.voxel_size
    else:
        pos0 = simulations._fill_mesh(n, sub, False, 123)
    ma = _mesh_arrays(sub)
    ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 9)
    kw = dict(max_iter=1000, epsilon=1e-13)
    for tr in (2, 4, 3):
        gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 9, kw, mesh=ma, tuning=dict(mesh_traversal=tr, resort_steps=64))
        _assert_identical(gpu, ref)


@pytest.mark.parametrize("meas_per_pass", [1, 8])
@pytest.mark.parametrize("M", [3, 20, 40])
def test_recording_with_forced_measurements_per_pass(lib, oracle_exact, M, meas_per_pass):
    """Recording + a forced meas_per_pass < 32: the recording walk kernel carries min(32, M)
    measurements itself, the follow-up passes must start after them (round-1 advisor finding:
    measurements 8..19 were accumulated twice)."""
    n, T, dt = 1200, 40, 1e-4
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    st = oracle_exact.rng_states(n, 42)
    pos = np.zeros((n, 3))
    ph, exc, traj = oracle_exact.run(oracle_exact.geom(FREE), g, dt, step_l, pos, st, want_traj=True)
    with lib.Sim(n, g, dt, step_l) as sim:
        sim.set_tuning(meas_per_pass=meas_per_pass)
        sim.rng_init_device(42)
        sim.set_recording(True)
        sim.run_free(0, T)
        rec = sim.recorded(0, T)
        got = sim.phases()
        npt.assert_array_equal(sim.positions(), pos)
    npt.assert_array_equal(got, ph)
    npt.assert_array_equal(rec, traj)


@pytest.mark.parametrize("kind", ["free", "sphere", "mesh", "mesh re-sorted every 7 steps"])
def test_recorded_trajectories_equal_the_oracles(lib, oracle_exact, kind):
    """dsim_set_recording / dsim_get_recorded against the ORACLE's per-step positions (want_traj:
    what the reference writes to its trajectories file, simulations.py:1043-1048, :1211-1213).  A mesh run is re-sorted
    in space while it records (round 4: slot i records into the column of its walker, several launches per call each
    writing its part of the range): the recording still comes out in walker order."""
    resort = 7 if "re-sorted" in kind else 0
    kind = kind.split()[0]
    from disimpy_amd import substrates

    n, T, dt = 5000, 30, 5e-4
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(2, T)
    st = oracle_exact.rng_states(n, 77)
    if kind == "free":
        pos0 = np.zeros((n, 3))
        geom = oracle_exact.geom(FREE)
    elif kind == "sphere":
        pos0 = (np.random.RandomState(4).random_sample((n, 3)) - 0.5) * 2e-6
        geom = oracle_exact.geom(SPHERE, radius=2e-6, max_iter=1000, epsilon=1e-13)
    else:
        m = golden("cylinder_mesh_closed.npz")
        sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.array([1e-6, 1e-6, 0.0]), n_sv=np.array([6, 6, 12]),
                              quiet=True)
        pos0 = np.random.RandomState(4).random_sample((n, 3)) * sub.voxel_size
        geom = oracle_exact.geom(MESH, mesh=_mesh_arrays(sub), max_iter=1000, epsilon=1e-13)
    pos = pos0.copy()
    ph, exc, traj = oracle_exact.run(geom, g, dt, step_l, pos, st, want_traj=True, n_threads=THREADS)
    with lib.Sim(n, g, dt, step_l) as sim:
        sim.rng_init_device(77)
        sim.set_positions(pos0)
        if kind == "mesh":
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
        sim.set_recording(True)
        if resort:
            sim.set_tuning(resort_steps=resort)
        rec = []
        for t0, t1 in ((0, 17), (17, T)):
            if kind == "free":
                sim.run_free(t0, t1)
            elif kind == "sphere":
                sim.run_sphere(2e-6, 1000, 1e-13, t0, t1)
            else:
                sim.run_mesh(1000, 1e-13, 0.0, t0, t1)
            rec.append(sim.recorded(t0, t1))
        got_ph = sim.phases()
    npt.assert_array_equal(np.concatenate(rec), traj)
    npt.assert_array_equal(got_ph, ph)


def test_trajectories_file_native_writer(lib, oracle_exact, tmp_path):
    """simulation(traj=...) on 1e4 walkers x 100 steps: the file written by dsim_write_traj (blocks of
    steps double-buffered over a copy stream, formatted by host threads) is byte-identical to the
    reference's Python formulation applied to the ORACLE's per-step positions, and the native hand-over
    + formatting is >= 10x faster than the Python writer on the same positions."""
    import time

    from disimpy_amd import gradients, simulations, substrates

    n, n_t = 10000, 100
    gradient, dt = gradients.pgse(10e-3, 40e-3, n_t, np.array([1e9]), np.array([[1.0, 0, 0]]))
    path = str(tmp_path / "traj.txt")
    sub = substrates.sphere(5e-6)
    simulations.simulation(n, D, gradient, dt, sub, seed=123, traj=path, quiet=True)
    # the oracle's trajectory of the same run
    step_l = float(np.sqrt(6 * D * dt))
    pos0 = np.ascontiguousarray(simulations._fill_sphere(n, sub.radius, simulations._host_rng(123)))
    pos = pos0.copy()
    st = oracle_exact.rng_states(n, 123)
    geom = oracle_exact.geom(SPHERE, radius=sub.radius, max_iter=1000, epsilon=1e-13)
    _, _, traj = oracle_exact.run(geom, gradient, dt, step_l, pos, st, want_traj=True, n_threads=THREADS)
    want = str(tmp_path / "want.txt")
    t0 = time.perf_counter()
    simulations._write_traj_python(want, "w", pos0)
    for t in range(n_t):
        simulations._write_traj_python(want, "a", traj[t])
    t_python = time.perf_counter() - t0
    assert open(path, "rb").read() == open(want, "rb").read()
    # speed of the hand-over + writer alone
    with lib.Sim(n, gradient, dt, step_l) as sim:
        sim.rng_init_device(123)
        sim.set_positions(pos0)
        sim.set_recording(True)
        sim.run_sphere(sub.radius, 1000, 1e-13, 0, n_t)
        t0 = time.perf_counter()
        lib.write_traj_line(path, pos0, append=False)
        sim.write_traj(path, 0, n_t)
        t_native = time.perf_counter() - t0
    assert open(path, "rb").read() == open(want, "rb").read()
    print("trajectories: native %.3f s, python %.3f s" % (t_native, t_python))
    assert t_native * 10 < t_python


@pytest.mark.parametrize("grid", ["uneven", "offset"])
def test_a_grid_that_is_not_the_references_linspace_takes_the_checked_traversal(lib, oracle_exact, grid):
    """the cooperative traversal's float32 shortcut and arithmetic cell coordinates assume
    xs = linspace(0, L, n + 1).  Boundaries handed in through the C ABI that are not (one boundary moved by 3 % of a
    subvoxel; all boundaries offset from 0) must not diverge silently: upload_mesh detects them and the per-lane traversal,
    which checks every guess against the boundary values, runs instead - M = 1, 4, 40 and a recorded run, default tuning,
    against the oracle on the same irregular grid."""
    from disimpy_amd import substrates

    m = golden("cylinder_mesh_closed.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.array([1e-6, 1e-6, 0.0]), n_sv=np.array([6, 6, 12]), quiet=True)
    xs, ys, zs = sub.xs.copy(), sub.ys.copy(), sub.zs.copy()
    if grid == "uneven":
        xs[2] += 0.03 * (xs[1] - xs[0])
        zs[5] -= 0.02 * (zs[1] - zs[0])
    else:
        ys = ys + 0.25 * (ys[1] - ys[0])  # xs[0] != 0: the reference's own periodic arithmetic, whatever it does with it
    tri, svi = lib.mesh_subdivide(sub.vertices, sub.faces, xs, ys, zs, sub.n_sv)
    ma = dict(vertices=sub.vertices, faces=sub.faces, xs=xs, ys=ys, zs=zs, n_sv=sub.n_sv,
              subvoxel_indices=np.ascontiguousarray(svi, dtype=np.int64), triangle_indices=np.ascontiguousarray(tri, dtype=np.int64))
    n, T, dt = 5000, 60, 5e-4
    step_l = float(np.sqrt(6 * D * dt))
    pos0 = np.random.RandomState(8).random_sample((n, 3)) * sub.voxel_size
    for M in (1, 4, 40):
        g = _random_gradient(M, T)
        ref = _oracle_mesh(oracle_exact, ma, n, g, dt, step_l, pos0, 21, want_traj=(M == 4))
        with lib.Sim(n, g, dt, step_l) as sim:
            sim.rng_init_device(21)
            sim.set_positions(pos0)
            sim.set_mesh(ma["vertices"], ma["faces"], xs, ys, zs, ma["n_sv"], ma["subvoxel_indices"], ma["triangle_indices"])
            if M == 4:
                sim.set_recording(True)
            sim.run_mesh(1000, 1e-13, 0.0, 0, T)
            if M == 4:
                npt.assert_array_equal(sim.recorded(0, T), ref["traj"])
            c, s_, _ = sim.signal()
            gpu = dict(pos=sim.positions(), phases=sim.phases(), exc=sim.iter_exc(), states=sim.rng_states(), cos=c, sin=s_)
        _assert_identical(gpu, ref)


_FULL_SIZE_POSITIONS = {}


@pytest.mark.parametrize("workload,n,n_meas,t_mid,t_end", [
    ("c5b", 10_000_000, 1, 130, 300), ("c5b", 10_000_000, 4, 130, 300), ("c5b", 10_000_000, 8, 130, 300),
    ("c4", 10_000_000, 1, 130, 300), ("c3", 1_000_000, 1, 130, 300), ("c2", 1_000_000, 32, 400, 1000),
    ("c5b", 10_000_000, 1, 6000, 10000),  # the headline run itself: every one of its 10 000 steps
    ("c4", 10_000_000, 1, 6000, 10000), ("c3", 1_000_000, 1, 6000, 10000)])  # [r5] and the other mesh / analytic configs over their full length
def test_full_size_sample_equals_the_oracle_and_the_partition_does_not_matter(lib, oracle_exact, workload, n, n_meas, t_mid, t_end):
    """BASELINE sizes (configs C5b, C4, C3, C2): all the walkers of the bench workload for the first steps of its
    waveform (C2, and the last C5b case - the run bench.py times: ALL of it, ~11 launches per walker group with their
    re-sorts, clocks and the final run-everyone-to-the-end launch), in two calls.  (i) 514 walkers drawn from all over the shard are bit-identical -
    position, phases, RNG state, iter_exc - to the oracle run on them alone from the same initial position and RNG
    state (walkers are independent, so a sample pins the full-size run: mesh - both walker groups, the spatial
    pre-sort, the re-sorts, per-walker clocks and the restore of the order, with 1 and 4 measurements in registers and
    with 8 through recorded positions; C2 - the 32-accumulator kernel).  (ii) The same walkers cut into two ragged
    shards with `first_walker` offsets give the same bytes as the one shard."""
    import hashlib

    from disimpy_amd import simulations
    from oracle.oracle import STATE_DTYPE

    w = bench.make_workload(workload, n, n_meas) if workload not in ("c1", "c2") else bench.make_workload(workload, n)
    sub, g, dt = w["substrate"], w["gradient"], w["dt"]
    assert g.shape[0] == n_meas
    step_l = float(np.sqrt(6 * D * dt))
    if (workload, n) not in _FULL_SIZE_POSITIONS:  # the mesh fill of 1e7 points takes a few seconds: once per workload
        _FULL_SIZE_POSITIONS.clear()
        pos0, _ = simulations._initial_positions(n, sub, 123, True, 0)
        _FULL_SIZE_POSITIONS[(workload, n)] = np.zeros((n, 3)) if pos0 is None else np.ascontiguousarray(pos0)
    pos0 = _FULL_SIZE_POSITIONS[(workload, n)]

    def run(a, b):
        with lib.Sim(b - a, g, dt, step_l, first_walker=a) as sim:
            sim.rng_init_device(123)
            st0 = sim.rng_states()
            sim.set_positions(pos0[a:b])
            if sub.type == "mesh":
                sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
            for t0, t1 in ((0, t_mid), (t_mid, t_end)):
                if sub.type == "mesh":
                    sim.run_mesh(1000, 1e-13, 0.0, t0, t1)
                elif sub.type == "sphere":
                    sim.run_sphere(sub.radius, 1000, 1e-13, t0, t1)
                else:
                    sim.run_free(t0, t1)
            if sub.type == "mesh":
                assert sim.last_walk_busy_ms()[1] == 2  # two walker groups
            return st0, sim.positions(), sim.phases(), sim.rng_states(), sim.iter_exc()

    st0, pos, ph, st, exc = run(0, n)
    assert np.isfinite(pos).all()
    idx = np.unique(np.concatenate([[0, 1, n - 1], np.linspace(0, n - 1, 512).astype(np.int64)]))
    o_st = np.zeros(len(idx), dtype=STATE_DTYPE)
    o_st["s0"], o_st["s1"] = st0[idx, 0], st0[idx, 1]
    o_pos = np.ascontiguousarray(pos0[idx])
    geom = bench.oracle_geometry(oracle_exact, sub)
    o_ph, o_exc, _ = oracle_exact.run(geom, g, dt, step_l, o_pos, o_st, t_begin=0, t_end=t_end, n_threads=THREADS)
    npt.assert_array_equal(pos[idx], o_pos)
    npt.assert_array_equal(ph[:, idx], o_ph)
    npt.assert_array_equal(st[idx], np.stack([o_st["s0"], o_st["s1"]], axis=1))
    npt.assert_array_equal(np.asarray(exc[idx], dtype=bool), o_exc.astype(bool))

    def digest(parts):
        h = hashlib.sha256()
        for p in parts:
            h.update(np.ascontiguousarray(p).tobytes())
        return h.hexdigest()

    whole = [digest([x]) for x in (pos, st, np.asarray(exc, dtype=np.uint8))] + [digest([ph[m]]) for m in range(len(ph))]
    del pos, st, exc, ph
    cut = n // 3
    a, b = run(0, cut), run(cut, n)
    halves = [digest([a[k], b[k]]) for k in (1, 3)] + [digest([np.asarray(a[4], dtype=np.uint8), np.asarray(b[4], dtype=np.uint8)])]
    halves += [digest([a[2][m], b[2][m]]) for m in range(len(a[2]))]
    assert halves == whole
