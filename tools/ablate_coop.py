"""Timing ablation of the cooperative mesh traversal (development build, GPU box):
    make -C disimpy_amd/csrc libdsim_hip_ablate.so
    DSIM_LIB=disimpy_amd/csrc/libdsim_hip_ablate.so python tools/ablate_coop.py [c5|c4]

CAVEAT: modes 1-4 switch the reflections off, so confined walkers (C5: inside the neuron) escape
into empty space within tens of steps and the later steps see no triangles at all - the numbers
are lower bounds of the stage costs, not a breakdown.  tools/mesh_timers.py (shader-clock timers
in the intact kernel) is the instrument to trust."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload  # noqa: E402
from disimpy_amd import _lib, simulations  # noqa: E402

MODES = {0: "full", 3: "survivors dropped (no fp64 tests, no reflections)",
         4: "survivors dropped, all lanes read the same 4 records", 2: "cells enumerated, nothing queued",
         1: "every subvoxel empty (step + lookups + sv_micro load)"}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
    n, T = 1_000_000, 600
    w = make_workload(wl, n)
    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    if isinstance(sub.init_pos, str) and sub.init_pos == "intra":
        pos = simulations._fill_mesh(n, sub, True, 123)
    else:
        pos = np.random.RandomState(123).random_sample((n, 3)) * sub.voxel_size
    lib = _lib.load()
    for mode, label in MODES.items():
        lib.dsim_debug_ablate(mode)
        with _lib.Sim(n, gradient, dt, step_l) as sim:
            sim.rng_init_device(123)
            sim.set_positions(np.ascontiguousarray(pos))
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                         sub.triangle_indices)
            sim.run_mesh(1000, 1e-13, 0.0, 0, 100)
            sim.run_mesh(1000, 1e-13, 0.0, 100, 100 + T)
            ms, _ = sim.last_run_ms()
        cyc = ms * 1e-3 * 2.2e9 * 1024 / ((n / 64) * T)
        print("%-58s %8.1f ms  %.3e spin-steps/s  ~%6.0f cycles per wave-step" % (label, ms, n * T / (ms * 1e-3), cyc), flush=True)
    lib.dsim_debug_ablate(0)


if __name__ == "__main__":
    main()
