"""Development aid (GPU box): time a fixed number of launches of the mesh fill (C5b neuron, intra) through a fill session -
used with DSIM_LIB=<a -DDSIM_FILL_ABLATE build> to see what the draws / look-ups cost without the bucket walks."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from disimpy_amd import _lib  # noqa: E402

n, launches = int(sys.argv[1]), int(sys.argv[2])
sub = bench.make_workload("c5b", 1000)["substrate"]
t = time.perf_counter()
sess = _lib.FillSession(0, n, True, 123, sub.vertices, sub.faces, sub.voxel_size, sub.xs, sub.ys, sub.zs, sub.n_sv,
                        sub.subvoxel_indices, sub.triangle_indices)
print("session opened in %.3f s" % (time.perf_counter() - t))
sess.run(64, 10_000_000)
t = time.perf_counter()
counts, pts = sess.run(launches, 10_000_000)
dt = time.perf_counter() - t
print("%d launches of %d threads: %.3f s (%.2f ms per launch), accepted %d" % (launches, n, dt, dt / launches * 1e3, counts.sum()))
sess.close()
