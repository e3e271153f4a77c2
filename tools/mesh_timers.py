"""Shader-clock time per phase of the cooperative mesh traversal (development aid, GPU box):
    make -C disimpy_amd/csrc libdsim_hip_timers.so
    DSIM_LIB=disimpy_amd/csrc/libdsim_hip_timers.so python tools/mesh_timers.py [c5|c4] [walkers]
Cycles are wall time of a wavefront (it shares its SIMD with the other resident wavefronts).
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload  # noqa: E402
from disimpy_amd import _lib, simulations  # noqa: E402

NAMES = {0: "random step + loop head", 1: "pass set-up (LDS init)", 2: "stage 1: subvoxel limits, box, fine cells",
         12: "stage 1: runs -> queue", 13: "stage 1c: run look-up, items",
         4: "stage 2: item queue, f32 culling", 5: "stage 3A: fp64 tests + atomic min",
         6: "stage 3B: winner's triangle, ties", 7: "merge pool result / serial redo", 8: "stage 4: reflection",
         9: "position + phase update"}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    T = 400
    w = make_workload(wl, n)
    sub = w["substrate"]
    gradient, dt = w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    if isinstance(sub.init_pos, str) and sub.init_pos == "intra":
        pos = simulations._fill_mesh(n, sub, True, 123)
    else:
        pos = np.random.RandomState(123).random_sample((n, 3)) * sub.voxel_size
    lib = _lib.load()
    buf = (C.c_ulonglong * 16)()
    with _lib.Sim(n, gradient, dt, step_l) as sim:
        sim.rng_init_device(123)
        sim.set_positions(np.ascontiguousarray(pos))
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                     sub.triangle_indices)
        sim.run_mesh(1000, 1e-13, 0.0, 0, 100)
        lib.dsim_debug_timers(buf)
        sim.run_mesh(1000, 1e-13, 0.0, 100, 100 + T)
        ms, nl = sim.last_run_ms()
        lib.dsim_debug_timers(buf)
    v = np.array(buf[:], dtype=np.float64)
    wave_steps = (n + 63) // 64 * T
    total = sum(v[i] for i in NAMES)
    print("%s: %d walkers x %d steps, %.1f ms (%.3g spin-steps/s in this build)" % (wl, n, T, ms, n * T / ms * 1e3))
    print("passes per wave-step %.2f" % (v[10] / wave_steps))
    print("%-40s %16s %8s" % ("phase", "cycles/wave-step", "share"))
    for i, name in NAMES.items():
        print("%-40s %16.0f %7.1f%%" % (name, v[i] / wave_steps, 100 * v[i] / total))
    print("%-40s %16.0f" % ("total (wall cycles of one wavefront)", total / wave_steps))


if __name__ == "__main__":
    main()
