"""Development probe: does a GPU run a mesh workload faster when the walkers of one shard are split over several
independent handles (each with its own stream, driven by its own host thread) than in one handle?  One handle
leaves SIMDs idle at the end of every launch (the last wavefronts of a launch finish one after the other);
with several handles another handle's launch fills those slots.

    python tools/concurrency_probe.py c5b 1250000 1 2 4
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    wl, n = sys.argv[1], int(sys.argv[2])
    splits = [int(s) for s in sys.argv[3:]] or [1, 2, 4]
    import argparse

    args = argparse.Namespace(block_size=0, tile_steps=0, meas_per_pass=0, resort_steps=0, gradient_source=0, mesh_traversal=0)
    from disimpy_amd import simulations

    w = bench.make_workload(wl, n, 1)
    sub, T = w["substrate"], w["gradient"].shape[1]
    cache = "/tmp/pos_%s_%d.npy" % (wl, n)
    os.environ["DSIM_BENCH_POSITIONS_CACHE"] = cache
    for S in splits:
        edges = [n * k // S for k in range(S + 1)]
        sims = []
        for k in range(S):
            sim, _, _, _ = bench.setup_sim(args, w, n, edges[k], edges[k + 1], 0)
            sims.append(sim)

        def one_pass(sim):
            simulations._run_range(sim, sub, w["frames"], 1000, 1e-13, 0, T)

        def all_pass():
            th = [threading.Thread(target=one_pass, args=(s,)) for s in sims]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return time.perf_counter() - t0

        all_pass()
        times = [all_pass() for _ in range(2)]
        sig = sum(s.signal()[0][0] for s in sims)
        print("%s n=%d handles=%d: %.4g spin-steps/s (passes %s s)  sum cos = %.6f" % (
            wl, n, S, n * T / min(times), ["%.3f" % t for t in times], sig), flush=True)
        for s in sims:
            s.close()


if __name__ == "__main__":
    main()
