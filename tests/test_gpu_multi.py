"""Several GPUs behind the ONE drop-in call (``-m gpu``): ``simulations.simulation`` spreads its walkers over every
visible device through a multi-device handle of the C ABI (include/dsim.h, dsim_create_multi) - no ranks, no torch.
On the driver's one-GPU box the device list names device 0 twice (two or three shards on the one GPU, DSIM_DEVICES):
every output must be the single-shard run's bytes - per-walker arrays (walker i <- RNG state i and initial position i
whatever the split; reference outputs: disimpy/simulations.py:1406-1429) and the signal, which the device
adds up exactly as integers (dsim_get_signal_limbs)."""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 2e-9


@pytest.fixture(scope="module")
def mods():
    from disimpy_amd import _lib, gradients, simulations, substrates

    if _lib.device_count() <= 0:
        pytest.skip("needs an AMD GPU")
    return _lib, gradients, simulations, substrates


def _cases(gradients, substrates):
    g3, dt3 = gradients.pgse(10e-3, 40e-3, 150, np.array([1e9, 2e9, 5e8]), np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]))
    yield "free", 10007, g3, dt3, substrates.free(), {}
    yield "sphere, iteration limit hit", 5003, g3, dt3, substrates.sphere(2e-6), {"max_iter": 1}
    yield "cylinder", 4099, g3, dt3, substrates.cylinder(3e-6, np.array([1.0, 2.0, 0.5])), {}
    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]), quiet=True)
    yield "mesh, walkers placed by the device fill", 9001, g3, dt3, sub, {}
    g40, dt40 = gradients.pgse(10e-3, 40e-3, 120, np.linspace(1e8, 3e9, 40), np.tile(np.array([[1.0, 0, 0]]), (40, 1)))
    yield "mesh, 40 measurements", 6000, g40, dt40, sub, {}


def _run(simulations, devices, n, gradient, dt, sub, kw, **out):
    os.environ["DSIM_DEVICES"] = devices
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            res = simulations.simulation(n, D, gradient, dt, sub, seed=321, quiet=True, **kw, **out)
        return res, [str(x.message) for x in w]
    finally:
        del os.environ["DSIM_DEVICES"]


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_shards_on_one_gpu_return_the_single_shard_outputs(mods, devices):
    _lib, gradients, simulations, substrates = mods
    for name, n, gradient, dt, sub, kw in _cases(gradients, substrates):
        (sig1, pos1), w1 = _run(simulations, "0", n, gradient, dt, sub, kw, final_pos=True)
        (sig2, pos2), w2 = _run(simulations, devices, n, gradient, dt, sub, kw, final_pos=True)
        assert pos1.tobytes() == pos2.tobytes(), name
        assert sig1.tobytes() == sig2.tobytes(), name  # the device adds the signal up exactly (integer limbs): no summation order
        assert w1 == w2, name  # the warning text names the same global walker indices
        if "iteration" in name:
            assert len(w1) == 1 and "walkers [" in w1[0]
        all1, _ = _run(simulations, "0", n, gradient, dt, sub, kw, all_signals=True)
        all2, _ = _run(simulations, devices, n, gradient, dt, sub, kw, all_signals=True)
        assert all1.shape == (gradient.shape[0], n) and all1.tobytes() == all2.tobytes(), name


def test_trajectories_file_of_a_multi_shard_run_is_the_same_file(mods, tmp_path):
    _lib, gradients, simulations, substrates = mods
    gradient, dt = gradients.pgse(10e-3, 40e-3, 40, np.array([1e9]), np.array([[1.0, 0, 0]]))
    n = 1003
    files = []
    for devices in ("0", "0,0,0"):
        path = str(tmp_path / ("traj_%d.txt" % len(files)))
        _run(simulations, devices, n, gradient, dt, substrates.sphere(3e-6), {}, traj=path)
        files.append(open(path, "rb").read())
    assert len(files[0].splitlines()) == 41 and files[0] == files[1]


def test_mesh_run_cut_by_space_writes_the_same_trajectories_file(mods, tmp_path):
    """simulation() on a mesh over three shards: the walkers are cut by space (a shard owns a region), the trajectories
    file, final positions and per-walker signals are the single-shard bytes"""
    _lib, gradients, simulations, substrates = mods
    gradient, dt = gradients.pgse(10e-3, 40e-3, 50, np.array([1e9]), np.array([[1.0, 0, 0]]))
    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]), quiet=True)
    n = 2003
    out = []
    for devices in ("0", "0,0,0"):
        path = str(tmp_path / ("traj_mesh_%d.txt" % len(out)))
        (sig, pos), _ = _run(simulations, devices, n, gradient, dt, sub, {}, traj=path, final_pos=True, all_signals=True)
        out.append((open(path, "rb").read(), sig.tobytes(), pos.tobytes()))
    assert len(out[0][0].splitlines()) == 51 and out[0] == out[1]


def test_mesh_fill_split_over_devices_returns_the_same_rows(mods):
    """dsim_fill_mesh_multi: one fill session per listed device on host threads, per-launch counts merged in memory -
    the rows of the one-device fill bit for bit (disimpy/simulations.py:505-579), inside and outside the surface"""
    _lib, gradients, simulations, substrates = mods
    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]), quiet=True)
    n = golden("neuron_mesh.npz")
    neuron = substrates.mesh(n["vertices"], n["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
    for substrate, count, intra in ((sub, 20011, True), (sub, 5003, False), (neuron, 2000, True)):
        one = simulations._fill_mesh(count, substrate, intra, 123)
        for devices in ([0, 0], [0, 0, 0, 0, 0]):
            many = simulations._fill_mesh(count, substrate, intra, 123, devices=devices)
            assert one.tobytes() == many.tobytes(), (count, intra, devices)


def test_multi_device_handle_state_round_trips(mods):
    """the handle's host arrays cover all walkers in order: every shard reads / fills its slice (ragged split, more
    shards than walkers)"""
    _lib, gradients, simulations, substrates = mods
    gradient, dt = gradients.pgse(10e-3, 40e-3, 30, np.array([1e9, 3e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    step_l = float(np.sqrt(6 * D * dt))
    for n, devices in ((11, [0, 0, 0]), (2, [0, 0, 0, 0]), (4097, [0, 0])):
        rng = np.random.RandomState(n)
        pos = rng.standard_normal((n, 3)) * 1e-6
        with _lib.Sim(n, gradient, dt, step_l, first_walker=5, devices=devices) as multi, \
                _lib.Sim(n, gradient, dt, step_l, first_walker=5) as one:
            assert multi.shard_count() == min(n, len(devices)) and one.shard_count() == 1
            for sim in (multi, one):
                sim.rng_init_device(99)
                sim.set_positions(pos)
            if n > 2:  # the shards cut by space instead of by index (dsim_partition_by_position): the same bytes everywhere below
                multi.partition_by_position(pos)
            np.testing.assert_array_equal(multi.positions(), pos)
            np.testing.assert_array_equal(multi.rng_states(), one.rng_states())
            st = one.rng_states()[::-1].copy()
            multi.set_rng_states(st)
            one.set_rng_states(st)
            for sim in (multi, one):
                sim.set_recording(True)
                sim.run_sphere(4e-6, 1000, 1e-13, 0, 30)
            np.testing.assert_array_equal(multi.positions(), one.positions())
            np.testing.assert_array_equal(multi.phases(), one.phases())
            np.testing.assert_array_equal(multi.rng_states(), one.rng_states())
            np.testing.assert_array_equal(multi.iter_exc(), one.iter_exc())
            np.testing.assert_array_equal(multi.all_signals(), one.all_signals())
            np.testing.assert_array_equal(multi.recorded(3, 17), one.recorded(3, 17))
            c2, s2, e2 = multi.signal()
            c1, s1, e1 = one.signal()
            assert c1.tobytes() == c2.tobytes() and s1.tobytes() == s2.tobytes() and e1 == e2
            np.testing.assert_array_equal(multi.signal_limbs()[0], one.signal_limbs()[0])
            # the exact sum against the host's correctly rounded sum of the same terms (device cos of the device's phases:
            # all_signals), term by term rounded to 2^-60 as the device does
            import math
            sig = one.all_signals()
            for m in range(sig.shape[0]):
                terms = sig[m][~np.isnan(sig[m])]
                want = math.fsum(np.rint(terms * 2.0**60) / 2.0**60)
                assert abs(c1[m] - want) <= 1e-15 * max(1.0, abs(want)), (m, c1[m], want)
    with pytest.raises(_lib.DsimError, match="shard 1 on device 7"):
        _lib.Sim(10, gradient, dt, step_l, devices=[0, 7])


def test_every_visible_gpu_is_used_by_default(mods):
    """a node with several GPUs: simulation() without DSIM_DEVICES = one shard per visible device"""
    _lib, gradients, simulations, substrates = mods
    if _lib.device_count() < 2:
        pytest.skip("one visible GPU: the default device list is [0] (covered by every other test)")
    gradient, dt = gradients.pgse(10e-3, 40e-3, 100, np.array([1e9]), np.array([[1.0, 0, 0]]))
    n = 200001
    assert simulations._default_devices(_lib.device_count()) == list(range(_lib.device_count()))
    sig, pos = simulations.simulation(n, D, gradient, dt, substrates.sphere(5e-6), seed=5, quiet=True, final_pos=True)
    (sig1, pos1), _ = _run(simulations, "0", n, gradient, dt, substrates.sphere(5e-6), {}, final_pos=True)
    assert pos.tobytes() == pos1.tobytes() and sig.tobytes() == sig1.tobytes()


def test_two_devices_behind_one_handle_return_the_one_device_bytes(mods, tmp_path):
    """the in-process twin of test_two_rccl_ranks_on_two_gpus: DSIM_DEVICES=0,1 against 0 - a mesh with the walkers placed
    by the fill that the two devices share, cut by space, every output of simulation() (needs two GPUs: skipped, never
    silently green, on the driver's one-GPU box)"""
    _lib, gradients, simulations, substrates = mods
    if _lib.device_count() < 2:
        pytest.skip("one visible GPU")
    gradient, dt = gradients.pgse(10e-3, 40e-3, 80, np.array([1e9, 4e8]), np.array([[1.0, 0, 0], [0, 0, 1.0]]))
    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]), quiet=True)
    n = 150_001
    out = []
    for devices in ("0", "0,1"):
        path = str(tmp_path / ("traj_2gpu_%d.txt" % len(out)))
        (sig, pos), w = _run(simulations, devices, n, gradient, dt, sub, {}, final_pos=True, all_signals=True)
        (tot, _), _ = _run(simulations, devices, 2003, gradient, dt, sub, {}, final_pos=True, traj=path)
        out.append((sig.tobytes(), pos.tobytes(), tot.tobytes(), open(path, "rb").read(), w))
    assert out[0] == out[1]


def test_partition_moves_rng_states_given_before_it_and_keeps_tuning_and_recording(mods):
    """dsim_partition_by_position on a mesh re-creates the shards with the sizes of the cost-balanced cut: RNG states that
    were GIVEN (not seeded) before are moved to the new cut, the tuning and the recording switch set before it survive"""
    _lib, gradients, simulations, substrates = mods
    gradient, dt = gradients.pgse(10e-3, 40e-3, 40, np.array([1e9]), np.array([[1.0, 0, 0]]))
    step_l = float(np.sqrt(6 * D * dt))
    m = golden("sphere_mesh.npz")
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]), quiet=True)
    n = 5001
    pos = simulations._fill_mesh(n, sub, True, 5)
    states = _lib.rng_init(77, 1000, n)[::-1].copy()  # any states, not those of a seed
    out = []
    for devices in (None, [0, 0, 0]):
        with _lib.Sim(n, gradient, dt, step_l, devices=devices) as sim:
            sim.set_tuning(0, 0, 0, 16, 0, 0)
            sim.set_recording(True)
            sim.set_rng_states(states)
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
            sim.partition_by_position(pos)
            np.testing.assert_array_equal(sim.rng_states(), states)
            np.testing.assert_array_equal(sim.positions(), pos)
            sim.run_mesh(1000, 1e-13, 0.0, 0, 40)
            out.append((sim.positions().tobytes(), sim.phases().tobytes(), sim.rng_states().tobytes(), sim.recorded(0, 40).tobytes(),
                        sim.signal_limbs()[0].tobytes()))
    assert out[0] == out[1]
