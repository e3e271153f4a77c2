"""Walker sharding across the GPUs of one node (one process per GPU).

Walkers are independent; the only cross-walker operation of the path is the
final signal sum.  GPU g of G owns a contiguous global walker range of
floor(N/G) or floor(N/G)+1 walkers (balanced split; a rank whose range is empty
skips the GPU work and contributes zeros); walker i ALWAYS uses RNG state i and
initial position i, so results do not depend on G except for floating-point
summation order - and not even for that: the device adds the signal up exactly, as integers
(include/dsim.h, dsim_get_signal_limbs).  The single collective is one all-reduce (sum, int64,
4*M + 1 values: two limbs per cos / sin sum, exceeded-walker count) over
``torch.distributed`` - backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests.  Payload <= ~1.6 KB: latency only, link bandwidth is irrelevant.

Note: PyTorch wheels bundle their own HIP runtime.  In a process that uses both
torch.cuda and this package, import torch BEFORE the first call into
``disimpy_amd`` (bench.py and tests/conftest.py do), otherwise torch reports
"No HIP GPUs are available".
"""
import numpy as np


def shard_range(n_walkers, rank, world_size):
    """[a, b) of global walker indices owned by ``rank``."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n_walkers, world_size)  # the first `extra` ranks own one walker more
    a = rank * base + min(rank, extra)
    return a, a + base + (1 if rank < extra else 0)


def reduce_signal(cos_sum, sin_sum, n_exceeded, dist=None, device=None):
    """All-reduce (sum) the per-shard signal.  ``dist`` is ``torch.distributed``
    (initialised) or None for a single process.  Returns (cos, sin, n_exceeded)."""
    cos_sum = np.asarray(cos_sum, dtype=np.float64)
    sin_sum = np.asarray(sin_sum, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return cos_sum, sin_sum, int(n_exceeded)
    import torch

    M = len(cos_sum)
    buf = torch.from_numpy(np.concatenate([cos_sum, sin_sum, [float(n_exceeded)]]))
    if dist.get_backend() == "nccl":
        buf = buf.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    return out[:M], out[M : 2 * M], int(round(out[2 * M]))


def reduce_signal_exact(limbs, n_exceeded, dist=None, device=None):
    """All-reduce of the EXACT signal sums: ``limbs`` = the (4 M,) int64 of ``_lib.Sim.signal_limbs()`` (every term rounded
    to a multiple of 2^-60, summed as integers in two limbs per sum).  Integer sums are the same in any order, so the
    all-reduced total - turned into doubles by ``_lib.signal_from_limbs`` on every rank - is bit for bit what one GPU
    returns for all the walkers.  One collective of 4 M + 1 int64 (RCCL / gloo).  Returns (cos, sin, n_exceeded)."""
    from . import _lib

    buf = np.concatenate([np.asarray(limbs, dtype=np.int64), [int(n_exceeded)]])
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch

        t = torch.from_numpy(buf)
        if dist.get_backend() == "nccl":
            t = t.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        buf = t.cpu().numpy()
    c, s = _lib.signal_from_limbs(buf[:-1])
    return c, s, int(buf[-1])


def _all_gather(arr, dist, device=None):
    """all-gather of equally shaped numpy arrays -> list, one per rank"""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(arr))
    on_gpu = dist.get_backend() == "nccl"
    if on_gpu:
        t = t.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]


def fill_mesh_sharded(n_points, open_session, dist=None, device=None, walker_range=None):
    """``simulations._fill_mesh`` with the launches' threads split over the ranks (disimpy/simulations.py:505-579).

    The reference launches one candidate per thread again and again and appends the accepted candidates of a launch
    in thread order until ``n_points`` are there.  Thread t draws from RNG state t whatever the split, so rank r runs
    the same launches over ITS range of threads (``open_session(thread_begin, thread_end)`` -> an object with
    ``run(n_launches, capacity) -> (counts, points)`` and ``close()``: disimpy_amd._lib.FillSession) and only the
    per-launch counts are exchanged - one small all-gather per batch of launches - to place every accepted point in
    the global sequence; at the end one all-gather hands the rows out.  The result is the single-process sequence bit
    for bit; the device work per rank is 1 / world_size of it.

    Returns the rows ``walker_range`` = [a, b) of the sequence (default: this rank's walker shard)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    a, b = walker_range if walker_range is not None else shard_range(n_points, rank, world)
    t0, t1 = shard_range(n_points, rank, world)  # threads of this rank (the same balanced split as the walkers)
    sess = open_session(t0, t1) if t1 > t0 else None
    try:
        have, launches, batch = 0, 0, 1
        rows_kept, pts_kept = [], []
        while have < n_points:
            # a rank's accepted points beyond the (n_points - have)-th cannot be needed; with an estimate of the rate
            # the buffer is kept small (8x the expectation: an overflow would have to be reported, never dropped)
            cap = min(n_points - have, (t1 - t0) * batch)
            if launches and have:
                cap = min(cap, int(8.0 * have / launches * batch * (t1 - t0) / n_points) + 100_000)
            # a failure on ONE rank (buffer overflow, a library error) must end the loop on EVERY rank: the status travels
            # with the counts of the batch, nobody leaves the collective alone (the others would hang in it)
            status, problem = 0, None
            counts, pts = np.zeros(batch, dtype=np.int64), np.zeros((0, 3))
            if sess is not None:
                try:
                    counts, pts = sess.run(batch, cap)
                    if int(counts.sum()) > len(pts) and len(pts) < n_points - have:
                        status, problem = -1, "%d accepted points in one batch overflow the buffer of %d" % (counts.sum(), cap)
                except Exception as e:  # noqa: BLE001 - reported below, on every rank
                    status, problem = -2, "%s: %s" % (type(e).__name__, e)
                    counts, pts = np.zeros(batch, dtype=np.int64), np.zeros((0, 3))
            msg = np.concatenate([np.asarray(counts, dtype=np.int64), [status]])
            gathered = np.stack(_all_gather(msg, dist, device)) if world > 1 else msg[None, :]
            bad = np.flatnonzero(gathered[:, -1] < 0)
            if len(bad):
                raise RuntimeError("fill_mesh_sharded: rank %d failed (%s)" % (
                    int(bad[0]), problem if problem is not None and rank in bad else "status %d" % gathered[bad[0], -1]))
            all_counts = gathered[:, :-1]
            per_launch = all_counts.sum(axis=0)
            launch_base = have + np.concatenate([[0], np.cumsum(per_launch)[:-1]])
            mine = launch_base + all_counts[:rank].sum(axis=0)  # global row of my first point of every launch
            rows = np.concatenate([m + np.arange(c) for m, c in zip(mine, counts)]) if counts.sum() else np.zeros(0, dtype=np.int64)
            rows = rows[: len(pts)]
            keep = rows < n_points
            rows_kept.append(rows[keep].astype(np.int64))
            pts_kept.append(pts[: len(rows)][keep])
            launches += batch
            have += int(per_launch.sum())
            if launches > 1_000_000:
                raise RuntimeError("fill_mesh: no progress after 1000000 launches (is the surface closed?)")
            if 0 < have < n_points:  # the same batch rule as the library's own loop, from globally known numbers
                left = (n_points - have) / (have / launches)
                batch = int(min(64.0, max(1.0, np.floor(left * 0.9))))
            elif have == 0:
                batch = min(batch * 2, 64)
    finally:
        if sess is not None:
            sess.close()
    rows = np.concatenate(rows_kept) if rows_kept else np.zeros(0, dtype=np.int64)
    pts = np.concatenate(pts_kept) if pts_kept else np.zeros((0, 3))
    out = np.empty((b - a, 3))
    if world == 1:
        sel = (rows >= a) & (rows < b)
        out[rows[sel] - a] = pts[sel]
        return out
    sizes = [int(x[0]) for x in _all_gather(np.array([len(rows)], dtype=np.int64), dist, device)]
    width = max(sizes + [1])
    pad_rows = np.full(width, -1, dtype=np.int64)
    pad_rows[: len(rows)] = rows
    pad_pts = np.zeros((width, 3))
    pad_pts[: len(rows)] = pts
    for r_rows, r_pts in zip(_all_gather(pad_rows, dist, device), _all_gather(pad_pts, dist, device)):
        sel = (r_rows >= a) & (r_rows < b)
        out[r_rows[sel] - a] = r_pts[sel]
    return out


def _gather_rows(arr, dist, device=None):
    """Concatenation over the ranks (in rank order) of arrays that differ in their first dimension only."""
    arr = np.ascontiguousarray(arr)
    sizes = [int(x[0]) for x in _all_gather(np.array([len(arr)], dtype=np.int64), dist, device)]
    width = max(sizes + [1])
    pad = np.zeros((width,) + arr.shape[1:], dtype=arr.dtype)
    pad[: len(arr)] = arr
    return np.concatenate([part[:n] for part, n in zip(_all_gather(pad, dist, device), sizes)])


def simulation_sharded(n_walkers, diffusivity, gradient, dt, substrate, seed=123, quiet=True, max_iter=int(1e3),
                       epsilon=1e-13, dist=None, device=None, final_pos=False, all_signals=False, traj=None):
    """``simulations.simulation`` with the walkers split over the ranks of the initialised process group (one process
    per GPU); every rank returns what the single-process call returns: the total signal (sum of cos(phase) per
    measurement) or, ``all_signals``, the (M, n_walkers) per-walker signals; with ``final_pos`` also the (n_walkers, 3)
    final positions - per-walker outputs gathered in walker order - and every rank raises the reference's warning
    about walkers that exceeded ``max_iter`` with GLOBAL walker indices (disimpy/simulations.py:1406-1429).  ``traj``:
    rank 0 writes the reference's trajectories file (:1043-1048, :1194, :1211-1213), the ranks' columns gathered per range
    of steps.  On a mesh the ranks cut the walkers by SPACE (include/dsim.h, dsim_partition_walkers: the Morton order of
    the initial positions in parts of equal estimated cost; every rank computes the same cut), elsewhere by index; walker i keeps RNG
    state i and position i either way and the outputs are the single-process bytes.
    (Inside one process ``simulations.simulation`` itself spreads the walkers over all visible GPUs; this entry is
    for jobs that are launched as one process per GPU.)"""
    import warnings

    from . import _lib, simulations

    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    simulations._validate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, 128, max_iter)
    np.random.seed(seed)
    a, b = shard_range(n_walkers, rank, world)
    M = gradient.shape[0]
    ids_all = None  # mesh, several ranks: the walkers of every rank (ascending), a spatial cut
    kw = {}
    if world > 1 and substrate.type == "mesh":
        # every rank needs all initial positions for the cut; the shared fill ends in an all-gather of the rows anyway
        positions_all, _ = simulations._initial_positions(n_walkers, substrate, seed, True, device or 0, (0, n_walkers), dist=dist)
        cost = _lib.mesh_walker_cost(substrate.vertices, substrate.faces, substrate.xs, substrate.ys, substrate.zs, substrate.n_sv,
                                     substrate.subvoxel_indices, substrate.triangle_indices, float(np.sqrt(6 * diffusivity * dt)), positions_all)
        ids_all = _lib.partition_walkers(positions_all, world, cost)  # parts of equal estimated cost
        kw = {"walker_ids": ids_all[rank], "positions_all": positions_all}
        a, b = 0, len(ids_all[rank])
    def traj_sink(t0, t1, rows):
        """this rank's part of the trajectories file -> whole lines on rank 0.  ``t0 == -1``: the start line, ``rows`` =
        the (n_local, 3) initial positions ``_simulate`` hands to the device (NOT computed a second time here: a mesh
        with ``init_pos="uniform"`` draws them from the global NumPy stream, which a second call would find advanced);
        otherwise (steps, n_local, 3), the positions after the steps [t0, t1)."""
        if t0 < 0:
            cols = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, 1, 3))
        else:
            cols = np.ascontiguousarray(rows.transpose(1, 0, 2))  # (n_local, steps, 3): the gather runs over walkers
        whole = _gather_rows(cols, dist, device)
        if rank == 0:
            if ids_all is not None:
                out = np.empty_like(whole)
                out[np.concatenate(ids_all)] = whole
                whole = out
            for k in range(whole.shape[1]):
                _lib.write_traj_line(traj, np.ascontiguousarray(whole[:, k, :]), append=t0 >= 0)

    # one process: the library's own writer (dsim_write_traj) - the sink is for gathering the ranks' columns
    sink = traj_sink if traj and world > 1 else None
    if b == a:  # more ranks than walkers: nothing to simulate here, but the collectives still need this rank
        if world > 1 and ids_all is None and substrate.type == "mesh" and isinstance(substrate.init_pos, str) and substrate.init_pos in ("intra", "extra"):
            simulations._initial_positions(n_walkers, substrate, seed, True, device or 0, (a, b), dist=dist)  # the shared fill
        res = {"cos": np.zeros(M), "sin": np.zeros(M), "n_exceeded": 0, "iter_exc": np.zeros(0, dtype=bool),
               "all_signals": np.zeros((M, 0)), "positions": np.zeros((0, 3)), "limbs": np.zeros(4 * M, dtype=np.int64)}
        if traj:  # take part in the gathers of every range of steps
            n_t = gradient.shape[1]
            chunk = max(1, min(n_t, (1 << 30) // (24 * max(1, n_walkers))))
            traj_sink(-1, 0, np.zeros((0, 3)))
            for t0 in range(0, n_t, chunk):
                traj_sink(t0, min(n_t, t0 + chunk), np.zeros((min(n_t, t0 + chunk) - t0, 0, 3)))
    else:
        res = simulations._simulate(n_walkers, diffusivity, gradient, dt, substrate, seed, traj, quiet, max_iter, epsilon,
                                    device=device or 0, walker_range=None if ids_all is not None else (a, b),
                                    dist=dist if world > 1 and ids_all is None else None,
                                    need_phases=bool(all_signals), need_positions=bool(final_pos),
                                    traj_sink=sink, **kw)
    if "limbs" in res:  # the exact integer sums of the device reduction: the total does not depend on the split
        c, s, nexc = reduce_signal_exact(res["limbs"], res["n_exceeded"], dist, device)
    else:
        c, s, nexc = reduce_signal(res["cos"], res["sin"], res["n_exceeded"], dist, device)

    def in_walker_order(rows):
        """per-walker rows of every rank, gathered in rank order -> walker order"""
        whole = _gather_rows(rows, dist, device) if world > 1 else rows
        if ids_all is None:
            return whole
        out = np.empty_like(whole)
        out[np.concatenate(ids_all)] = whole
        return out

    if nexc:  # rare: the flags of all walkers, for the warning text (same on every rank)
        exc = in_walker_order(res["iter_exc"].astype(np.uint8)).astype(bool)
        if np.any(exc):
            warnings.warn(
                "Maximum number of iterations was exceeded in the intersection "
                + "check algorithm for walkers %s" % np.where(exc)[0]
            )
    signals = c
    if all_signals:
        signals = np.ascontiguousarray(in_walker_order(np.ascontiguousarray(res["all_signals"].T)).T)
    if final_pos:
        return signals, in_walker_order(res["positions"])
    return signals
