"""Traversal statistics of the mesh kernel (development aid, GPU box):
    make -C disimpy_amd/csrc libdsim_hip_stats.so
    DSIM_LIB=disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stats.py [c5|c4]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload  # noqa: E402
from disimpy_amd import _lib, simulations  # noqa: E402

NAMES = ["reflect-loop iterations", "subvoxels in range", "runs of fine cells queued", "non-empty runs",
         "record batches (4)", "run tasks (incl. empty)", "AABB pass -> fp64 test", "- (round 2: repeated survivors dropped; no longer filtered)",
         "serial: subvoxels scanned by the owner", "serial: iterations redone (queues full / tie)", "-"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
    n, T = 200_000, 400
    w = make_workload(wl, n)
    sub = w["substrate"]
    gradient, dt = w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    if isinstance(sub.init_pos, str) and sub.init_pos == "intra":
        pos = simulations._fill_mesh(n, sub, True, 123)
    else:
        pos = np.random.RandomState(123).random_sample((n, 3)) * sub.voxel_size
    lib = _lib.load()
    buf = (C.c_ulonglong * 32)()
    with _lib.Sim(n, gradient, dt, step_l) as sim:
        sim.rng_init_device(123)
        sim.set_positions(np.ascontiguousarray(pos))
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                     sub.triangle_indices)
        sim.run_mesh(1000, 1e-13, 0.0, 0, 100)
        lib.dsim_debug_stats(buf)
        sim.run_mesh(1000, 1e-13, 0.0, 100, 100 + T)
        lib.dsim_debug_stats(buf)
    v = np.array(buf[:], dtype=np.float64)
    lane_steps = n * T
    wave_steps = (n + 63) // 64 * T
    print("%s: %d walkers x %d steps" % (wl, n, T))
    print("%-28s %14s %18s %12s" % ("event", "per lane-step", "wave trips/wave-step", "lanes/trip"))
    for i, name in enumerate(NAMES):
        lane, wave = v[i] / lane_steps, v[16 + i] / wave_steps
        print("%-28s %14.3f %18.3f %12.1f" % (name, lane, wave, (v[i] / v[16 + i]) if v[16 + i] else 0))


if __name__ == "__main__":
    main()
