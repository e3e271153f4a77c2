"""Development aid (GPU box): the whole mesh fill of the bench workload (simulations._initial_positions, C5b neuron, intra),
as bench.py's setup_s.initial_positions times it, with its phases."""
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402
from disimpy_amd import simulations  # noqa: E402

n = int(sys.argv[1])
sub = bench.make_workload("c5b", 1000)["substrate"]
for rep in range(3):
    t = time.perf_counter()
    positions, frames = simulations._initial_positions(n, sub, 123, True, 0, (0, n))
    print("fill of %d points: %.3f s" % (n, time.perf_counter() - t))
