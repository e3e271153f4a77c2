"""CPU estimate (round 6): records a pass reads from the fine grid with disjoint cells along z (round 5) and with
overlapping cells along z as well (DS_ZOV = 1).  Walkers inside the C5b neuron (the oracle's fill), random first-pass
segments of one step length; per query the columns / cells the device would visit and the records they hold.
    DSIM_LIB=.../libdsim_hip_zov.so python tools/dev/zoverlap_records.py 1     (1 = the library was built with DS_ZOV=1)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from disimpy_amd import _lib  # noqa: E402
from oracle.oracle import MESH, Oracle  # noqa: E402

zov = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
w = bench.make_workload("c5b", 1000, 1)
sub = w["substrate"]
step_l = float(np.sqrt(6 * w["D"] * w["dt"]))
fg = _lib.mesh_fine_grid(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices, step_l)
lk = fg["lk"]
k = 1 << lk
ku = 2 * k
kz = ku if zov else k
slots = ku * ku * kz
cs = fg["cell_start"].astype(np.int64)
base = fg["sv_base"].astype(np.int64)
print("lk", lk, "records", cs[-1], "cells", len(cs) - 1, "slots/subvoxel", slots, "step_l", step_l)
cache = "/tmp/c5b_pts_%d.npy" % n_pts
if os.path.exists(cache):
    pts = np.load(cache)
else:
    o = Oracle(exact=True)
    ma = dict(vertices=sub.vertices, faces=sub.faces, xs=sub.xs, ys=sub.ys, zs=sub.zs, n_sv=sub.n_sv,
              subvoxel_indices=sub.subvoxel_indices, triangle_indices=sub.triangle_indices)
    g = o.geom(MESH, mesh=ma)
    pts = o.fill_mesh(n_pts, g, sub.voxel_size, True, 123)
    np.save(cache, pts)
rng = np.random.RandomState(5)
d = rng.standard_normal((len(pts), 3))
d /= np.linalg.norm(d, axis=1)[:, None]
n_sv = np.asarray(sub.n_sv)
h = np.asarray(sub.voxel_size) / n_sv
tot_rec = tot_runs = tot_items = 0
hist = np.zeros(200, dtype=np.int64)
for L in (step_l,):
    a, b = pts, pts + d * L
    lo, hi = np.minimum(a, b) / h, np.maximum(a, b) / h  # subvoxel units
    for i in range(len(pts)):
        sv0 = np.floor(lo[i]).astype(int)
        sv1 = np.floor(hi[i]).astype(int)
        rec = runs = items = 0
        for sx in range(sv0[0], sv1[0] + 1):
            for sy in range(sv0[1], sv1[1] + 1):
                for sz in range(sv0[2], sv1[2] + 1):
                    s = ((sx % n_sv[0]) * n_sv[1] + (sy % n_sv[1])) * n_sv[2] + (sz % n_sv[2])
                    if base[s] < 0:
                        runs += 1  # an empty run is still queued and looked up
                        continue
                    u = []
                    for ax, (sv, per) in enumerate(zip((sx, sy, sz), (ku, ku, kz))):
                        u0 = int(np.clip(np.floor((lo[i][ax] - sv) * per - 1e-3), 0, per - 1))
                        u1 = int(np.clip(np.floor((hi[i][ax] - sv) * per + 1e-3), 0, per - 1))
                        u.append((u0, u1))
                    cols = []
                    for ax in range(2):
                        u0, u1 = u[ax]
                        cols.append([min(c, ku - 2) for c in range(u0, max(u0, u1 - 1) + 1, 2)])
                    if zov:
                        c0 = min(u[2][0], kz - 2)
                        c1 = max(c0, u[2][1] - 1)
                    else:
                        c0, c1 = u[2]
                    for ca in cols[0]:
                        for cb in cols[1]:
                            c = base[s] + (ca * ku + cb) * kz
                            n = cs[c + c1 + 1] - cs[c + c0]
                            rec += n
                            runs += 1
                            items += (n + 4) // 5
        tot_rec += rec
        tot_runs += runs
        tot_items += items
        hist[min(rec, 199)] += 1
n = len(pts)
print("zov=%d: per first pass: records %.2f  runs %.3f  items(5) %.3f" % (zov, tot_rec / n, tot_runs / n, tot_items / n))
print("records histogram (0..30):", hist[:31].tolist())
