"""Event counters (-DDSIM_STATS build) for the two queue-overflow configurations of
tests/test_gpu_parity.py::test_mesh_work_beyond_the_wave_queues (development aid, GPU box):
    DSIM_LIB=disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stress_stats.py
"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disimpy_amd import _lib, simulations, substrates
m=np.load('tests/golden/neuron_mesh.npz')
NAMES = ["iterations", "subvoxels visited", "-", "cells->items", "item trips", "-", "fp64 tests", "-", "subvoxel scanned serially", "serial redo", "-"]
for n_sv,dt in (([10,10,10],3e-3),([3,3,3],3e-2)):
    sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", n_sv=np.array(n_sv), quiet=True)
    n,T=512,12
    step_l=float(np.sqrt(6*2e-9*dt))
    pos=simulations._fill_mesh(n, sub, True, 7)
    lib=_lib.load(); buf=(C.c_ulonglong*32)()
    with _lib.Sim(n, np.ones((1,T,3)), dt, step_l) as sim:
        sim.rng_init_device(5); sim.set_positions(np.ascontiguousarray(pos))
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices)
        lib.dsim_debug_stats(buf)
        sim.run_mesh(1000,1e-13,0.0,0,T)
        lib.dsim_debug_stats(buf)
    v=np.array(buf[:],dtype=float)
    print(n_sv, 'step', step_l, {NAMES[i]: (v[i], v[16+i]) for i in range(11) if NAMES[i]!='-'})
