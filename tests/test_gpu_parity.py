"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP kernels, called
through the C ABI (disimpy_amd._lib), against the CPU oracle on the same seeded
inputs.

Bars:
* vs the oracle's EXACT build (same log/cos spec): bit-identical positions,
  phases, RNG states and iter_exc flags on every substrate - reflecting
  geometries are chaotic, so nothing weaker than walker-level identity is
  meaningful there;
* vs the oracle's glibc build (the one pinned to the reference's golden
  trajectory): free diffusion within 1e-6 on S/N (north_star tolerance);
* vs the reference's golden trajectory itself: <= 6e-8 per-step direction error.
"""
import numpy as np
import numpy.testing as npt
import pytest

from conftest import golden
from oracle.oracle import CYLINDER, ELLIPSOID, FREE, MESH, SPHERE, STATE_DTYPE

pytestmark = pytest.mark.gpu

D = 2e-9


@pytest.fixture(scope="module")
def lib():
    from disimpy_amd import _lib

    assert _lib.device_count() > 0, "these tests need an AMD GPU"
    return _lib


def _states_to_struct(st2):
    out = np.zeros(len(st2), dtype=STATE_DTYPE)
    out["s0"], out["s1"] = st2[:, 0], st2[:, 1]
    return out


def _random_gradient(M, T, seed=5, scale=0.05):
    rng = np.random.RandomState(seed)
    g = rng.normal(size=(M, T, 3)) * scale
    g[:, T // 3 : T // 2] = 0  # a silent stretch, like the gap of a PGSE
    return g


def _gpu_run(lib, kind, n, gradient, dt, step_l, pos0, seed, geom_kw=None, chunks=None, mesh=None, tuning=None,
             first_walker=0):
    geom_kw = geom_kw or {}
    T = gradient.shape[1]
    with lib.Sim(n, gradient, dt, step_l, first_walker=first_walker) as sim:
        if tuning:
            sim.set_tuning(**tuning)
        sim.rng_init_device(seed)
        if pos0 is not None:
            sim.set_positions(pos0)
        if mesh is not None:
            sim.set_mesh(mesh["vertices"], mesh["faces"], mesh["xs"], mesh["ys"], mesh["zs"], mesh["n_sv"],
                         mesh["subvoxel_indices"], mesh["triangle_indices"])
        bounds = [0, T] if chunks is None else chunks
        for t0, t1 in zip(bounds[:-1], bounds[1:]):
            if kind == FREE:
                sim.run_free(t0, t1)
            elif kind == SPHERE:
                sim.run_sphere(geom_kw["radius"], geom_kw["max_iter"], geom_kw["epsilon"], t0, t1)
            elif kind == CYLINDER:
                sim.run_cylinder(geom_kw["radius"], geom_kw["R"], geom_kw["R_inv"], geom_kw["max_iter"],
                                 geom_kw["epsilon"], t0, t1)
            elif kind == ELLIPSOID:
                sim.run_ellipsoid(geom_kw["semiaxes"], geom_kw["R"], geom_kw["R_inv"], geom_kw["max_iter"],
                                  geom_kw["epsilon"], t0, t1)
            else:
                sim.run_mesh(geom_kw["max_iter"], geom_kw["epsilon"], geom_kw.get("perm_prob", 0.0), t0, t1)
        c, s, nexc = sim.signal()
        return dict(pos=sim.positions(), phases=sim.phases(), exc=sim.iter_exc(), states=sim.rng_states(), cos=c,
                    sin=s, nexc=nexc)


def _oracle_run(o, kind, n, gradient, dt, step_l, pos0, seed, geom_kw=None, mesh=None, first_walker=0):
    geom_kw = dict(geom_kw or {})
    st = o.rng_states(first_walker + n, seed)[first_walker:].copy()
    pos = np.zeros((n, 3)) if pos0 is None else np.array(pos0, dtype=np.float64, order="C")
    g = o.geom(kind, mesh=mesh, **geom_kw)
    ph, exc, _ = o.run(g, gradient, dt, step_l, pos, st)
    c, s = o.signal(ph, exc)
    return dict(pos=pos, phases=ph, exc=exc.astype(bool), states=np.stack([st["s0"], st["s1"]], axis=1), cos=c, sin=s)


def _assert_identical(gpu, ref):
    npt.assert_array_equal(gpu["states"], ref["states"])
    npt.assert_array_equal(gpu["exc"], ref["exc"])
    npt.assert_array_equal(gpu["pos"], ref["pos"])
    npt.assert_array_equal(gpu["phases"], ref["phases"])
    n = gpu["pos"].shape[0]
    npt.assert_allclose(gpu["cos"], ref["cos"], rtol=0, atol=1e-10 * n)
    npt.assert_allclose(gpu["sin"], ref["sin"], rtol=0, atol=1e-10 * n)


# ----------------------------------------------------------------------- RNG


def test_device_rng_seeding_matches_sequential_jumps(lib, oracle_libm):
    n, seed = 5000, 123
    g = np.zeros((1, 1, 3))
    for first in (0, 1, 777, 2**20 + 3):
        with lib.Sim(n, g, 1e-3, 1e-6, first_walker=first) as sim:
            sim.rng_init_device(seed)
            got = sim.rng_states()
        want = lib.rng_init(seed, first, n)  # host: matrices to `first`, then the reference's sequential jumps
        npt.assert_array_equal(got, want)
    st = oracle_libm.rng_states(n, seed)
    with lib.Sim(n, g, 1e-3, 1e-6) as sim:
        sim.rng_init_device(seed)
        got = sim.rng_states()
    npt.assert_array_equal(got[:, 0], st["s0"])
    npt.assert_array_equal(got[:, 1], st["s1"])


# ---------------------------------------------------------------- free path


def test_golden_trajectory_on_gpu(lib):
    """The reference's own output: 10 walkers x 999 steps, seed 123 (test_traj.txt)."""
    traj = golden("traj_free_seed123.npy").reshape(1000, 10, 3)
    step_l = 9.84731927834662e-07
    g = np.zeros((1, 999, 3))
    rows = [np.zeros((10, 3))]
    with lib.Sim(10, g, 8.0808e-05, step_l) as sim:
        sim.rng_init_device(123)
        for t in range(999):
            sim.run_free(t, t + 1)
            rows.append(sim.positions())
    full = np.array(rows)
    err = np.linalg.norm((np.diff(traj, axis=0) - np.diff(full, axis=0)) / step_l, axis=2)
    assert err.max() < 6e-8
    assert np.median(err) < 1e-12
    assert np.abs(full - traj).max() < 1e-12


@pytest.mark.parametrize("M", [1, 3, 32, 40])
def test_free_bit_identical_to_exact_oracle(lib, oracle_exact, M):
    n, T, dt = 1500, 64, 1e-4
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    ref = _oracle_run(oracle_exact, FREE, n, g, dt, step_l, None, 42)
    gpu = _gpu_run(lib, FREE, n, g, dt, step_l, None, 42)
    _assert_identical(gpu, ref)


def test_chunked_ranges_tiles_and_block_sizes_do_not_change_results(lib, oracle_exact):
    n, T, dt, M = 700, 50, 1e-4, 5
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    ref = _oracle_run(oracle_exact, FREE, n, g, dt, step_l, None, 9)
    for chunks, tuning in [
        ([0, 1, 2, 17, 50], None),
        (None, dict(tile_steps=7)),
        (None, dict(block_size=64)),
        (None, dict(meas_per_pass=2)),  # 3 passes over the measurements: replays the walk
        ([0, 25, 50], dict(meas_per_pass=1, tile_steps=3, block_size=128)),
        (None, dict(gradient_source=1)),  # LDS tiles (the scalar-load variant needs -DDSIM_SMEM_GRADIENT)
    ]:
        gpu = _gpu_run(lib, FREE, n, g, dt, step_l, None, 9, chunks=chunks, tuning=tuning)
        _assert_identical(gpu, ref)


def test_free_within_1e6_of_glibc_oracle(lib, oracle_libm):
    """north_star tolerance: outputs match on identical seeds to 1e-6 (on S/N)."""
    from disimpy_amd import gradients

    n = 20000
    gradient, dt = gradients.pgse(10e-3, 40e-3, 100, np.array([1e9, 2e9]), np.array([[1.0, 0, 0], [0, 1.0, 0]]))
    step_l = float(np.sqrt(6 * D * dt))
    ref = _oracle_run(oracle_libm, FREE, n, gradient, dt, step_l, None, 123)
    gpu = _gpu_run(lib, FREE, n, gradient, dt, step_l, None, 123)
    assert np.abs(gpu["cos"] - ref["cos"]).max() / n < 1e-6  # tolerance: 1e-6 on S/N
    assert np.abs(gpu["pos"] - ref["pos"]).max() < 1e-11  # metres


def test_walker_shards_reproduce_the_whole(lib):
    """Walker i uses RNG state i whatever the partition."""
    n, T, dt = 1000, 40, 1e-4
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(2, T)
    whole = _gpu_run(lib, FREE, n, g, dt, step_l, None, 3)
    a = _gpu_run(lib, FREE, 400, g, dt, step_l, None, 3, first_walker=0)
    b = _gpu_run(lib, FREE, 600, g, dt, step_l, None, 3, first_walker=400)
    npt.assert_array_equal(np.concatenate([a["pos"], b["pos"]]), whole["pos"])
    npt.assert_array_equal(np.concatenate([a["phases"], b["phases"]], axis=1), whole["phases"])
    npt.assert_allclose(a["cos"] + b["cos"], whole["cos"], rtol=0, atol=1e-10 * n)


# ------------------------------------------------------ reflecting geometries


def _rot(seed):
    from disimpy_amd import utils

    rng = np.random.RandomState(seed)
    R = utils.vec2vec_rotmat(rng.normal(size=3), rng.normal(size=3))
    return R, np.linalg.inv(R)


@pytest.mark.parametrize("radius", [1e-6, 5e-6])
def test_sphere_bit_identical(lib, oracle_exact, radius):
    from disimpy_amd import simulations

    n, T, dt, M = 600, 150, 2e-4, 2
    step_l = float(np.sqrt(6 * D * dt))  # 1.5 um: several reflections per step at r = 1 um
    g = _random_gradient(M, T)
    pos0 = simulations._fill_sphere(n, radius, simulations._host_rng(1))
    kw = dict(radius=radius, max_iter=1000, epsilon=1e-13)
    ref = _oracle_run(oracle_exact, SPHERE, n, g, dt, step_l, pos0, 11, kw)
    gpu = _gpu_run(lib, SPHERE, n, g, dt, step_l, pos0, 11, kw)
    _assert_identical(gpu, ref)
    assert np.linalg.norm(gpu["pos"], axis=1).max() < radius


def test_iteration_limit_flag(lib, oracle_exact):
    """max_iter = 2 with a step 20x the radius: most walkers hit the limit."""
    from disimpy_amd import simulations

    n, T, dt = 300, 5, 1e-3
    radius = 5e-7
    step_l = 1e-5
    g = _random_gradient(1, T)
    pos0 = simulations._fill_sphere(n, radius, simulations._host_rng(2))
    kw = dict(radius=radius, max_iter=2, epsilon=1e-13)
    ref = _oracle_run(oracle_exact, SPHERE, n, g, dt, step_l, pos0, 1, kw)
    gpu = _gpu_run(lib, SPHERE, n, g, dt, step_l, pos0, 1, kw)
    _assert_identical(gpu, ref)
    assert gpu["exc"].sum() > n // 2 and gpu["nexc"] == gpu["exc"].sum()
    # nansum semantics: exceeded walkers do not contribute
    keep = ~gpu["exc"]
    npt.assert_allclose(gpu["cos"][0], np.cos(gpu["phases"][0, keep]).sum(), atol=1e-9)


def test_cylinder_bit_identical(lib, oracle_exact):
    from disimpy_amd import simulations, utils

    n, T, dt, M = 500, 120, 2e-4, 3
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    orientation = np.array([1.0, 0.3, -2.0])
    orientation /= np.linalg.norm(orientation)
    R = utils.vec2vec_rotmat(orientation, np.array([1.0, 0, 0]))
    R_inv = np.linalg.inv(R)
    radius = 2e-6
    pos0 = simulations._initial_positions_cylinder(n, radius, R_inv, simulations._host_rng(3))
    kw = dict(radius=radius, R=R, R_inv=R_inv, max_iter=1000, epsilon=1e-13)
    ref = _oracle_run(oracle_exact, CYLINDER, n, g, dt, step_l, pos0, 21, kw)
    gpu = _gpu_run(lib, CYLINDER, n, g, dt, step_l, pos0, 21, kw)
    _assert_identical(gpu, ref)
    body = (R @ gpu["pos"].T).T
    assert np.linalg.norm(body[:, 1:], axis=1).max() < radius


def test_ellipsoid_bit_identical(lib, oracle_exact):
    from disimpy_amd import simulations

    n, T, dt, M = 500, 120, 2e-4, 2
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    R_inv, R = _rot(4)
    semiaxes = np.array([4e-6, 1.5e-6, 2.5e-6])
    pos0 = simulations._initial_positions_ellipsoid(n, semiaxes, R_inv, simulations._host_rng(5))
    kw = dict(semiaxes=semiaxes, R=R, R_inv=R_inv, max_iter=1000, epsilon=1e-13)
    ref = _oracle_run(oracle_exact, ELLIPSOID, n, g, dt, step_l, pos0, 31, kw)
    gpu = _gpu_run(lib, ELLIPSOID, n, g, dt, step_l, pos0, 31, kw)
    _assert_identical(gpu, ref)
    body = (R @ gpu["pos"].T).T
    assert (np.sum((body / semiaxes) ** 2, axis=1) < 1).all()


# ----------------------------------------------------------------------- mesh


def _mesh_arrays(sub):
    return dict(vertices=sub.vertices, faces=sub.faces, xs=sub.xs, ys=sub.ys, zs=sub.zs,
                subvoxel_indices=sub.subvoxel_indices, triangle_indices=sub.triangle_indices, n_sv=sub.n_sv)


@pytest.mark.parametrize("periodic", [True, False])
@pytest.mark.parametrize("n_sv", [[1, 1, 1], [1, 5, 20], [10, 10, 10]])
@pytest.mark.parametrize("perm_prob", [0.0, 0.35])
@pytest.mark.parametrize("traversal", [1, 2])  # per-lane / wavefront-cooperative traversal of the index
def test_mesh_bit_identical(lib, oracle_exact, periodic, n_sv, perm_prob, traversal):
    from disimpy_amd import substrates

    m = golden("cylinder_mesh_closed.npz")
    pad = np.array([1e-6, 0.0, 2e-6])
    sub = substrates.mesh(m["vertices"], m["faces"], periodic, padding=pad, n_sv=np.array(n_sv), quiet=True,
                          perm_prob=perm_prob)
    n, T, dt, M = 400, 80, 5e-4, 2
    step_l = float(np.sqrt(6 * D * dt))  # 2.4 um in a 5 um cylinder
    g = _random_gradient(M, T)
    rng = np.random.RandomState(8)
    pos0 = rng.random_sample((n, 3)) * sub.voxel_size  # inside and outside the cylinder
    kw = dict(max_iter=1000, epsilon=1e-13, perm_prob=perm_prob)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 77, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 77, kw, mesh=ma, tuning=dict(mesh_traversal=traversal))
    _assert_identical(gpu, ref)
    if not periodic and perm_prob == 0.0:
        assert (gpu["pos"] > 0).all() and (gpu["pos"] < sub.voxel_size).all()


@pytest.mark.parametrize("resort_steps,traversal", [(-1, 2), (7, 2), (0, 2), (7, 1)])
def test_mesh_spatial_resorting_is_invisible(lib, oracle_exact, resort_steps, traversal):
    """>= 4096 walkers: the mesh path re-sorts the walker slots by Morton key every few
    steps (tuning knob) and restores the order at the end; per-walker results must not move."""
    from disimpy_amd import substrates

    m = golden("cylinder_mesh_closed.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.array([1e-6, 1e-6, 0.0]),
                          n_sv=np.array([6, 6, 12]), quiet=True)
    n, T, dt, M = 6000, 40, 5e-4, 3
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    pos0 = np.random.RandomState(18).random_sample((n, 3)) * sub.voxel_size
    kw = dict(max_iter=1000, epsilon=1e-13)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 5, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 5, kw, mesh=ma,
                   tuning=dict(resort_steps=resort_steps, mesh_traversal=traversal), chunks=[0, 13, 40])
    _assert_identical(gpu, ref)


@pytest.mark.parametrize("M", [5, 32, 40])
def test_mesh_many_measurements_phases_from_recorded_positions(lib, oracle_exact, M):
    """More than two measurements on a mesh: the cooperative walk kernel records the positions of
    every step and a second kernel accumulates the phases from them (M = 40: two passes of that
    kernel, no replay of the walk).  Same operations in the same order: bit-identical phases."""
    from disimpy_amd import substrates

    m = golden("cylinder_mesh_closed.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.array([1e-6, 1e-6, 0.0]),
                          n_sv=np.array([6, 6, 12]), quiet=True)
    n, T, dt = 4500, 50, 5e-4
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    pos0 = np.random.RandomState(28).random_sample((n, 3)) * sub.voxel_size
    kw = dict(max_iter=1000, epsilon=1e-13)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 15, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 15, kw, mesh=ma, tuning=dict(resort_steps=9, mesh_traversal=2),
                   chunks=[0, 21, 50])
    _assert_identical(gpu, ref)


@pytest.mark.parametrize("kind,M", [("free", 1), ("free", 40), ("sphere", 3), ("mesh", 1), ("mesh", 5)])
def test_recorded_trajectories_equal_step_by_step_positions(lib, kind, M):
    """dsim_set_recording: the positions after every step of a run, fetched once per range, are
    the positions a step-by-step run reports, and recording does not change phases or state."""
    from disimpy_amd import substrates

    n, T, dt = 5000, 30, 5e-4
    step_l = float(np.sqrt(6 * D * dt))
    g = _random_gradient(M, T)
    sub = None
    if kind == "mesh":
        m = golden("cylinder_mesh_closed.npz")
        sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.array([1e-6, 1e-6, 0.0]),
                              n_sv=np.array([6, 6, 12]), quiet=True)
    pos0 = None
    if kind == "mesh":
        pos0 = np.random.RandomState(4).random_sample((n, 3)) * sub.voxel_size
    elif kind == "sphere":
        pos0 = (np.random.RandomState(4).random_sample((n, 3)) - 0.5) * 2e-6

    def run(sim, t0, t1):
        if kind == "free":
            sim.run_free(t0, t1)
        elif kind == "sphere":
            sim.run_sphere(2e-6, 1000, 1e-13, t0, t1)
        else:
            sim.run_mesh(1000, 1e-13, 0.0, t0, t1)

    def make():
        sim = lib.Sim(n, g, dt, step_l)
        sim.rng_init_device(77)
        if pos0 is not None:
            sim.set_positions(pos0)
        if sub is not None:
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                         sub.triangle_indices)
        return sim

    with make() as sim:  # step by step, no recording
        steps = []
        for t in range(T):
            run(sim, t, t + 1)
            steps.append(sim.positions())
        ref_phases, ref_states = sim.phases(), sim.rng_states()
    with make() as sim:  # two recorded ranges
        sim.set_recording(True)
        run(sim, 0, 17)
        rec = [sim.recorded(0, 5), sim.recorded(5, 17)]
        run(sim, 17, T)
        rec.append(sim.recorded(17, T))
        with pytest.raises(lib.DsimError):
            sim.recorded(0, 5)  # only the last run is held
        phases, states = sim.phases(), sim.rng_states()
        sim.set_recording(False)
    rec = np.concatenate(rec, axis=0)
    assert rec.shape == (T, n, 3)
    assert np.array_equal(rec, np.stack(steps))
    assert np.array_equal(phases, ref_phases) and np.array_equal(states, ref_states)


def test_mesh_huge_steps_fall_back_to_serial_scans(lib, oracle_exact):
    """Steps several voxels long (more than 256 subvoxels in range, periodic images): the
    cooperative traversal hands such walkers to the serial path; results stay identical."""
    from disimpy_amd import substrates

    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 2e-6, n_sv=np.array([8, 8, 8]), quiet=True)
    n, T, dt = 200, 12, 1e-1
    step_l = float(np.sqrt(6 * D * dt))  # 35 um in a 14 um voxel
    g = _random_gradient(1, T)
    pos0 = np.random.RandomState(3).random_sample((n, 3)) * sub.voxel_size
    kw = dict(max_iter=1000, epsilon=1e-13)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 9, kw, mesh=ma)
    for traversal in (1, 2):
        gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 9, kw, mesh=ma, tuning=dict(mesh_traversal=traversal))
        _assert_identical(gpu, ref)


@pytest.mark.parametrize("traversal", [1, 2])
def test_mesh_neuron_bit_identical_and_contained(lib, oracle_exact, traversal):
    """The 29 688-triangle neuron of the reference's tests at the default 50^3 index."""
    from disimpy_amd import simulations, substrates

    m = golden("neuron_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
    n, T, dt = 300, 60, 1e-3
    step_l = float(np.sqrt(6 * D * dt))
    g = np.ones((1, T, 3))
    pos0 = simulations._fill_mesh(n, sub, True, 123)
    kw = dict(max_iter=1000, epsilon=1e-13)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 123, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 123, kw, mesh=ma, tuning=dict(mesh_traversal=traversal))
    _assert_identical(gpu, ref)
    assert (gpu["pos"].max(axis=0) < sub.voxel_size).all() and (gpu["pos"].min(axis=0) > 0).all()


@pytest.mark.parametrize("n_sv,dt", [([10, 10, 10], 3e-3), ([3, 3, 3], 3e-2)])
def test_mesh_work_beyond_the_wave_queues(lib, oracle_exact, n_sv, dt):
    """Coarse index of the neuron + steps of 6 / 19 um: every walker reaches dozens of micro-cells
    with hundreds of records, far more than the cooperative traversal's LDS queues hold per pass
    (256 cell tasks, 256 items, 128 survivors).  Items and survivors are then drained in rounds,
    walkers whose cell tasks do not fit are redone serially by their own lane; results must not
    change (checked once with the -DDSIM_STATS build: rounds, mid-drain flushes and serial
    redo all occur in these two configurations)."""
    from disimpy_amd import simulations, substrates

    m = golden("neuron_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", n_sv=np.array(n_sv), quiet=True)
    n, T = 512, 12
    step_l = float(np.sqrt(6 * D * dt))
    g = np.ones((1, T, 3))
    pos0 = simulations._fill_mesh(n, sub, True, 7)
    kw = dict(max_iter=1000, epsilon=1e-13)
    ma = _mesh_arrays(sub)
    ref = _oracle_run(oracle_exact, MESH, n, g, dt, step_l, pos0, 5, kw, mesh=ma)
    gpu = _gpu_run(lib, MESH, n, g, dt, step_l, pos0, 5, kw, mesh=ma, tuning=dict(mesh_traversal=2))
    _assert_identical(gpu, ref)


@pytest.mark.parametrize("periodic", [True, False])
@pytest.mark.parametrize("intra", [True, False])
def test_fill_mesh_identical_to_oracle(lib, oracle_exact, periodic, intra):
    from disimpy_amd import simulations, substrates

    m = golden("sphere_mesh.npz")
    pad = np.zeros(3) + 1e-6
    sub = substrates.mesh(m["vertices"], m["faces"], periodic, padding=pad, n_sv=np.array([4, 5, 6]), quiet=True)
    n = 3000
    got = simulations._fill_mesh(n, sub, intra, 123)
    if periodic:
        v, f, tri, svi = sub.vertices, sub.faces, sub.triangle_indices, sub.subvoxel_indices
    else:
        v, f, tri, svi = simulations._strip_wall_triangles(sub)
    g = oracle_exact.geom(MESH, mesh=dict(vertices=v, faces=f, xs=sub.xs, ys=sub.ys, zs=sub.zs, subvoxel_indices=svi,
                                           triangle_indices=tri, n_sv=sub.n_sv))
    want = oracle_exact.fill_mesh(n, g, sub.voxel_size, intra, 123)
    npt.assert_array_equal(got, want)
    r = (sub.voxel_size - pad * 2) / 2
    d = np.linalg.norm(got - (r + pad), axis=1)
    if intra:
        assert d.max() < r.max()
    else:
        assert d.min() > 0.9 * r.min()


@pytest.mark.parametrize("intra", [True, False])
def test_fill_mesh_neuron_identical_to_oracle(lib, oracle_exact, intra):
    """The reference's rejection sampler on its hardest case: the neuron fills 0.025 % of its voxel, so a run is
    thousands of launches (the library reads the count back once per batch of them), its triangles are listed in
    up to six subvoxels of a column (the kernel replaces the reference's 1000-slot list of met triangles by a
    precomputed "met before on this ray" relation), and most columns of the index are empty."""
    from disimpy_amd import simulations, substrates

    m = golden("neuron_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
    n = 3000 if intra else 20000
    got = simulations._fill_mesh(n, sub, intra, 123)
    g = oracle_exact.geom(MESH, mesh=_mesh_arrays(sub))
    want = oracle_exact.fill_mesh(n, g, sub.voxel_size, intra, 123, max_launches=100000)
    npt.assert_array_equal(got, want)


# --------------------------------------------------------------------- signal


def test_signal_kernel_nansum_semantics(lib):
    n, T, dt, M = 5000, 30, 1e-3, 4
    g = _random_gradient(M, T, scale=5.0)
    out = _gpu_run(lib, FREE, n, g, dt, 1e-6, None, 5)
    ph = out["phases"]
    assert np.abs(ph).max() > 10  # large arguments: full-range sincos
    npt.assert_allclose(out["cos"], np.cos(ph).sum(axis=1), rtol=0, atol=1e-9 * n)
    npt.assert_allclose(out["sin"], np.sin(ph).sum(axis=1), rtol=0, atol=1e-9 * n)
    assert out["nexc"] == 0
