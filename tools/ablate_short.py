"""Short-run timing ablation of stage 2's record gathers (development build, GPU box):
    make -C disimpy_amd/csrc libdsim_hip_ablate.so
    DSIM_LIB=disimpy_amd/csrc/libdsim_hip_ablate.so python tools/ablate_short.py [c5b] [walkers]
Modes 3 and 4 switch the reflections off (walkers then leave the structure), so only the first few steps from the
initial positions compare like with like: 12 steps, repeated from the same start."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload  # noqa: E402
from disimpy_amd import _lib, simulations  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c5b"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    T = 12
    w = make_workload(wl, n)
    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * 2e-9 * dt))
    pos = simulations._fill_mesh(n, sub, True, 123)
    lib = _lib.load()
    for mode, label in ((0, "full"), (3, "survivors dropped (no fp64 tests, no reflections)"),
                        (4, "survivors dropped, every lane reads the same 4 records"), (2, "runs looked up, nothing queued")):
        best = 1e9
        for rep in range(4):
            lib.dsim_debug_ablate(mode)
            with _lib.Sim(n, gradient, dt, step_l) as sim:
                sim.rng_init_device(123)
                sim.set_positions(np.ascontiguousarray(pos))
                sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                             sub.triangle_indices)
                sim.run_mesh(1000, 1e-13, 0.0, 0, T)
                best = min(best, sim.last_walk_ms()[0])
        print("%-58s %8.2f ms walk kernel for %d steps  (%.3e spin-steps/s)" % (label, best, T, n * T / (best * 1e-3)), flush=True)
    lib.dsim_debug_ablate(0)


if __name__ == "__main__":
    main()
