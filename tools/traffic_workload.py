"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
the C2 bench launch twice, plus two layout-conversion kernels with KNOWN byte counts and the
same access shape as the walk kernel's state traffic (8-byte lanes, coalesced), used to
calibrate FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, section HBM: the counter is only
calibrated for 16-byte lanes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload  # noqa: E402
from disimpy_amd import _lib  # noqa: E402

n = 1_000_000
w = make_workload(sys.argv[1] if len(sys.argv) > 1 else "c2", n)
g, dt = w["gradient"], w["dt"]
with _lib.Sim(n, g, dt, float(np.sqrt(6 * 2e-9 * dt))) as sim:
    sim.rng_init_device(123)
    for _ in range(2):
        sim.run_free(0, g.shape[1])
    sim.positions()   # soa_to_aos_kernel: reads 3 x 8 B per walker, writes 24 B per walker
    sim.rng_states()  # soa_to_states_kernel: reads 2 x 8 B per walker, writes 16 B per walker
    sim.signal()
