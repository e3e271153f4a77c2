"""Walker sharding across the GPUs of one node (one process per GPU).

Walkers are independent; the only cross-walker operation of the path is the
final signal sum.  GPU g of G owns a contiguous global walker range of
floor(N/G) or floor(N/G)+1 walkers (balanced split; a rank whose range is empty
skips the GPU work and contributes zeros); walker i ALWAYS uses RNG state i and
initial position i, so results do not depend on G except for floating-point
summation order.  The single collective is one all-reduce (sum, float64,
2*M + 1 values: sum cos, sum sin, exceeded-walker count) over
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


def simulation_sharded(n_walkers, diffusivity, gradient, dt, substrate, seed=123, quiet=True, max_iter=int(1e3),
                       epsilon=1e-13, dist=None, device=None):
    """``simulations.simulation`` with the walkers split over the ranks of the
    initialised process group; every rank returns the same total signal
    (sum of cos(phase) per measurement)."""
    from . import simulations

    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    simulations._validate(n_walkers, diffusivity, gradient, dt, substrate, seed, None, quiet, 128, max_iter)
    np.random.seed(seed)
    a, b = shard_range(n_walkers, rank, world)
    if b == a:  # more ranks than walkers: nothing to simulate here, but the collective still needs this rank
        M = gradient.shape[0]
        c, s, nexc = reduce_signal(np.zeros(M), np.zeros(M), 0, dist, device)
        return c
    res = simulations._simulate(n_walkers, diffusivity, gradient, dt, substrate, seed, None, quiet, max_iter, epsilon,
                                device=device or 0, walker_range=(a, b))
    c, s, nexc = reduce_signal(res["cos"], res["sin"], res["n_exceeded"], dist, device)
    return c
