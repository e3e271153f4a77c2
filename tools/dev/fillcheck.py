"""Development aid (GPU box): the neuron fill against the exact oracle at a size the test suite does not afford
(tests/test_gpu_parity.py::test_fill_mesh_neuron_identical_to_oracle uses 3000 points = 1.2e7 candidates)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_gpu_parity as tp  # noqa: E402
from disimpy_amd import simulations, substrates  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

n = int(sys.argv[1])
m = np.load("tests/golden/neuron_mesh.npz")
sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
got = simulations._fill_mesh(n, sub, True, 123)
o = Oracle(exact=True)
g = o.geom(tp.MESH, mesh=tp._mesh_arrays(sub))
t = time.perf_counter()
want = o.fill_mesh(n, g, sub.voxel_size, True, 123, max_launches=1000000)
print("oracle: %.1f s" % (time.perf_counter() - t))
print("fill of %d points inside the neuron: %s" % (n, "identical" if np.array_equal(got, want) else "DIFFERENT"))
