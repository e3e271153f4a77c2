"""Two PRODUCT ranks through the collective on the driver's one GPU (``-m gpu``): world_size 2 over gloo, both ranks
on device 0, each running disimpy_amd.sharding.simulation_sharded on its walker shard with the HIP kernels; the
all-reduced signal must equal the one-rank result (SURVEY 8d, C4 check) - bit for bit: the sums are exact integers.
The rank processes are forked from a fork server that conftest.py starts before anything in the session touches
the GPU (a GPU-initialised process must not fork+exec on this pool)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 2e-9


def _case(name):
    """(n_walkers, gradient, dt, substrate) of a case - built identically in every process"""
    sys.path.insert(0, ROOT)
    import bench
    from disimpy_amd import gradients, substrates

    if name == "free32":
        gradient, dt = gradients.pgse(10e-3, 40e-3, 200, 1e9 * np.ones(32), bench.fib_dirs(32))
        return 20001, gradient, dt, substrates.free()
    if name == "c4mesh":
        gradient, dt = bench.misst_waveform(10000, np.array([1e9]))
        gradient, dt = np.ascontiguousarray(gradient[:, :500]), dt
        v, f = bench.packed_cylinders()
        return 8192 + 77, gradient, dt, substrates.mesh(v, f, True, init_pos="uniform", quiet=True)
    if name == "mesh_outputs":  # walkers placed by the (shared) device fill, every output of simulation()
        gradient, dt = gradients.pgse(10e-3, 40e-3, 60, np.array([1e9, 3e8]), np.array([[1.0, 0, 0], [0, 1.0, 0]]))
        m = np.load(os.path.join(ROOT, "tests", "golden", "sphere_mesh.npz"))
        sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]),
                              quiet=True)
        return 3001, gradient, dt, sub
    raise ValueError(name)


def _fill_case():
    sys.path.insert(0, ROOT)
    from disimpy_amd import substrates

    m = np.load(os.path.join(ROOT, "tests", "golden", "sphere_mesh.npz"))
    sub = substrates.mesh(m["vertices"], m["faces"], True, padding=np.zeros(3) + 1e-6, init_pos="intra", n_sv=np.array([6, 6, 6]),
                          quiet=True)
    return 30011, sub


def _rank_main(rank, world, port, name, q, backend="gloo"):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch  # noqa: F401  before the HIP library of this package (see disimpy_amd/sharding.py)
        import torch.distributed as dist

        device = 0
        if backend == "nccl":  # RCCL over xGMI: one rank per GPU
            device = rank
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from disimpy_amd import sharding

        if name == "fill":
            from disimpy_amd import simulations

            n, sub = _fill_case()
            rows = simulations._fill_mesh(n, sub, True, 123, device=device, dist=dist, walker_range=sharding.shard_range(n, rank, world))
            q.put((rank, rows, None))
        elif name.startswith("mesh_outputs"):
            n, gradient, dt, sub = _case("mesh_outputs")
            path = name.split("|", 1)[1]
            sig, pos = sharding.simulation_sharded(n, D, gradient, dt, sub, seed=123, dist=dist, device=device, final_pos=True,
                                                   all_signals=True, traj=path)
            q.put((rank, (np.asarray(sig), np.asarray(pos)), None))
        else:
            n, gradient, dt, sub = _case(name)
            sig = sharding.simulation_sharded(n, D, gradient, dt, sub, seed=123, dist=dist, device=device)
            q.put((rank, np.asarray(sig), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # the parent must not wait for the timeout to learn about it
        import traceback

        q.put((rank, None, "%s\n%s" % (e, traceback.format_exc())))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("name", ["free32", "c4mesh"])
def test_two_product_ranks_equal_one(name, rank_spawner):
    from disimpy_amd import simulations

    ctx = rank_spawner
    world = 2
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, sig, err in got:
        assert err is None, "rank %d failed: %s" % (rank, err)
    assert all(p.exitcode == 0 for p in procs)
    n, gradient, dt, sub = _case(name)
    one = simulations.simulation(n, D, gradient, dt, sub, seed=123, quiet=True)
    for rank, sig, _ in got:
        np.testing.assert_array_equal(sig, one)  # exact integer sums all-reduced: the bytes of the one-rank run
    np.testing.assert_array_equal(got[0][1], got[1][1])  # every rank holds the same total
    assert np.all(np.abs(one) <= n) and np.all(np.abs(one) > 0)


def test_two_ranks_on_a_mesh_cut_by_space_return_every_output_and_the_trajectories_file(rank_spawner, tmp_path):
    """a job of one process per GPU on a mesh: the ranks share the fill, cut the walkers by SPACE (dsim_partition_walkers:
    each rank seeds ITS walkers through dsim_set_walker_ids) and return all_signals / final_pos in walker order; rank 0
    writes the trajectories file - all of it the single-process bytes"""
    from disimpy_amd import simulations

    ctx = rank_spawner
    world = 2
    q = ctx.Queue()
    port = _free_port()
    path = str(tmp_path / "traj_two_ranks.txt")
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, "mesh_outputs|" + path, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, out, err in got:
        assert err is None, "rank %d failed: %s" % (rank, err)
    n, gradient, dt, sub = _case("mesh_outputs")
    one_path = str(tmp_path / "traj_one.txt")
    os.environ["DSIM_DEVICES"] = "0"
    try:
        sig1, pos1 = simulations.simulation(n, D, gradient, dt, sub, seed=123, quiet=True, final_pos=True, all_signals=True, traj=one_path)
    finally:
        del os.environ["DSIM_DEVICES"]
    for rank, (sig, pos), _ in got:
        assert sig.tobytes() == sig1.tobytes() and pos.tobytes() == pos1.tobytes()
    assert open(path, "rb").read() == open(one_path, "rb").read()


def test_two_ranks_share_the_mesh_fill(rank_spawner):
    """_fill_mesh with the threads of every launch split over two ranks (dsim_fill_open / dsim_fill_run sessions, counts
    exchanged over gloo): each rank's rows are the single-process rows of its walker shard, bit for bit"""
    from disimpy_amd import sharding, simulations

    ctx = rank_spawner
    world = 2
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, "fill", q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, rows, err in got:
        assert err is None, "rank %d failed: %s" % (rank, err)
    n, sub = _fill_case()
    one = simulations._fill_mesh(n, sub, True, 123)
    for rank, rows, _ in got:
        a, b = sharding.shard_range(n, rank, world)
        np.testing.assert_array_equal(rows, one[a:b])


@pytest.mark.parametrize("name", ["free32", "c4mesh", "fill"])
def test_two_rccl_ranks_on_two_gpus(name, rank_spawner):
    """The RCCL path itself - nccl backend, one rank per GPU, GPU tensors through all_reduce / all_gather - the first
    time a box shows two devices (the driver's GPU boxes have one: skipped there, never silently green)."""
    from disimpy_amd import _lib, sharding, simulations

    if _lib.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank and this box has %d" % _lib.device_count())
    ctx = rank_spawner
    world = 2
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, name, q, "nccl")) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, out, err in got:
        assert err is None, "rank %d failed: %s" % (rank, err)
    if name == "fill":
        n, sub = _fill_case()
        one = simulations._fill_mesh(n, sub, True, 123)
        for rank, rows, _ in got:
            a, b = sharding.shard_range(n, rank, world)
            np.testing.assert_array_equal(rows, one[a:b])
    else:
        n, gradient, dt, sub = _case(name)
        one = simulations.simulation(n, D, gradient, dt, sub, seed=123, quiet=True)
        for rank, sig, _ in got:
            np.testing.assert_array_equal(sig, one)
