"""Development aid (GPU box): the rate of every part of the spatial cut of the headline workload (what each GPU of a
G-GPU job would run), one after the other on this GPU.  python tools/dev/parts_rate.py [G] [workload]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "c5b"
n = 10_000_000
args = bench.argparse.Namespace(block_size=0, tile_steps=0, meas_per_pass=0, resort_steps=0, gradient_source=0, mesh_traversal=0)
w = bench.make_workload(name, n, 1)
sub, T = w["substrate"], w["gradient"].shape[1]
import os
os.environ["DSIM_BENCH_POSITIONS_CACHE"] = "/tmp/pos_parts_%s.npy" % name
for k in range(G):
    sim, pos, _, ids = bench.setup_sim(args, w, n, 0, n, 0, part=(k, G))
    bench.run_steps(sim, sub, w["frames"], T, 1, None, 0, lambda: None)
    el, walk_ms, _, launches, sig = bench.run_steps(sim, sub, w["frames"], T, 1, None, 0, lambda: None)
    sim.close()
    print("part %d of %d: %d walkers, %.4g spin-steps/s (walk kernel %.4g), launches %d, extent %s um" % (
        k, G, len(ids), len(ids) * T / el, len(ids) * T / (walk_ms[0][1] * 1e-3), launches[0], np.round(np.ptp(pos, axis=0) * 1e6, 1)), flush=True)
