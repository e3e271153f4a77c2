"""Ablation timings of the mesh kernel (development aid; run on the GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload, misst_waveform  # noqa: E402
from disimpy_amd import _lib, simulations, substrates  # noqa: E402


def run(label, sub, pos, n, T=2000, resort=0):
    gradient, dt = misst_waveform(10000, np.array([1e9]))
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    with _lib.Sim(n, gradient, dt, step_l) as sim:
        sim.set_tuning(0, 0, 0, resort, 0, 0)
        sim.rng_init_device(123)
        sim.set_positions(np.ascontiguousarray(pos))
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                     sub.triangle_indices)
        sim.run_mesh(1000, 1e-13, 0.0, 0, 200)
        sim.run_mesh(1000, 1e-13, 0.0, 200, 200 + T)
        ms, nl = sim.last_run_ms()
    print("%-46s %.3e spin-steps/s  (%d launches, %.1f ms)" % (label, n * T / (ms * 1e-3), nl, ms), flush=True)


def main():
    n = 1_000_000
    rng = np.random.RandomState(1)
    # (1) a voxel the size of the neuron's with one tiny triangle: lookups + empty-subvoxel path only
    L = np.array([2.145e-4, 1.935e-4, 2.229e-4])
    v = np.array([[0, 0, 0], [1e-7, 0, 0], [0, 1e-7, 0], L], dtype=np.float64)
    f = np.array([[0, 1, 2], [3, 3, 3]], dtype=np.int64)
    sub = substrates.mesh(v, f, True, quiet=True)
    pos = rng.random_sample((n, 3)) * sub.voxel_size
    run("empty 50^3 grid, uniform walkers", sub, pos, n, resort=-1)
    run("empty 50^3 grid, uniform walkers, sorted", sub, pos, n, resort=0)
    # (2) neuron
    w = make_workload("c5", n)
    sub = w["substrate"]
    t0 = time.time()
    pos_intra = simulations._fill_mesh(n, sub, True, 123)
    print("fill_mesh intra: %.1f s" % (time.time() - t0))
    pos_uni = rng.random_sample((n, 3)) * sub.voxel_size
    run("neuron, uniform walkers (mostly extracellular)", sub, pos_uni, n, resort=-1)
    run("neuron, uniform walkers, sorted", sub, pos_uni, n, resort=0)
    run("neuron, intra walkers", sub, pos_intra, n, resort=-1)
    for rs in (0, 32, 64, 128, 512):
        run("neuron, intra walkers, resort=%d" % rs, sub, pos_intra, n, resort=rs)
    # (3) packed cylinders
    w = make_workload("c4", n)
    sub = w["substrate"]
    pos = rng.random_sample((n, 3)) * sub.voxel_size
    for rs in (-1, 0, 16, 128):
        run("packed cylinders, resort=%d" % rs, sub, pos, n, resort=rs)


if __name__ == "__main__":
    main()
