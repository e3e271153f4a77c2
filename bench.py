#!/usr/bin/env python3
"""Headline benchmark: spin-steps/s of the fused random-walk kernels on the north-star workload.

    python bench.py --gpus N --steps K --warmup W [--workload c5b|c5|c4|c3|c2|c1]

Default workload = BASELINE config C5 as the north star states it: the neuron mesh refined to
50 000 triangles (C5b), periodic, walkers inside the neuron, 1e7 spins x 1e4 steps, M = 1.
The job is STRONG-scaled: the 1e7 walkers are split over the N ranks (walker i keeps RNG state i
and initial position i, so the result does not depend on N), one rank per GPU.

One "step" = one pass of the hot path over one batch: every walker of this rank's shard is
advanced through the whole gradient waveform (T time steps, one fused launch per re-sort
interval), the signal is reduced on the device and - for N > 1 - all-reduced over RCCL.
Inputs (gradient, walker state, mesh index) are resident in HBM before the timed region;
successive steps continue the same walk.

`--gpus N` without WORLD_SIZE in the environment makes this process a launcher: it starts N
fresh rank processes itself (before anything touches the GPU), forwards rank 0's JSON line and
exits non-zero if any rank fails.  Under `python -m torch.distributed.run` it is a rank.

Prints ONE JSON line (rank 0).  `roofline` is the contract's object for the dominant kernel (the
cooperative mesh walk kernel): bound "hbm", `achieved` = ALGORITHMIC bytes per launch (SURVEY.md 8d:
80 + 16 M bytes per spin-step x the spin-steps of a launch) over the kernel's average launch duration,
measured in THIS run with HIP events on the library's streams; `traffic` = the HBM bytes per launch from
the PMC counters in profiles/kernel_counters.json.  The kernel keeps all walker state in registers, so
HBM is not what bounds it (frac ~ 0.18): the bound that holds - vector-ALU issue - is reported next to it
in `roofline.valu` (VALU wave-instructions per wavefront-step from the rocprofv3 SQ_INSTS_VALU count of the
same kernel, times the wavefront-steps per second of this run, over 1024 SIMDs x 2.4 GHz / 4 cycles), but
only when the counters were collected from the library that is running (`counters_match_lib`: sha256 of
libdsim_hip.so) on the same workload (`counters_match_config`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD, CLOCK_HZ, CYCLES_PER_VALU = 1024, 2.4e9, 4.0  # 256 CUs x 4 SIMDs; one wave64 VALU instruction per 4 cycles
VALU_PEAK = N_SIMD * CLOCK_HZ / CYCLES_PER_VALU  # wave-instructions per second

DEFAULT_TOTAL_WALKERS = {"c1": 1000, "c2": 1_000_000, "c3": 1_000_000, "c4": 10_000_000, "c5": 10_000_000,
                         "c5b": 10_000_000}  # BASELINE.json configs


def fib_dirs(n):
    """n golden-spiral unit vectors."""
    k = np.arange(n)
    z = 1 - (2 * k + 1) / n
    phi = k * np.pi * (3 - np.sqrt(5))
    r = np.sqrt(1 - z**2)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def misst_waveform(n_t, bvals):
    """The reference tests' MISST waveform (test_simulations.py:530-538): 700-point template, 70 ms."""
    from disimpy_amd import gradients

    T = 70e-3
    g = np.zeros((1, 700, 3))
    g[0, 1:300, 0] = 1
    g[0, -300:-1, 0] = -1
    dt = T / (g.shape[1] - 1)
    g = np.concatenate([g for _ in bvals], axis=0)
    g, dt = gradients.interpolate_gradient(g, dt, n_t)
    return gradients.set_b(g, dt, bvals), dt


def packed_cylinders(nside=8, pitch=2.5e-6, radius=1e-6, height=5e-6, nseg=32):
    """BASELINE config C4: square lattice of open 32-gon prisms."""
    verts, faces = [], []
    ang = np.arange(nseg) * 2 * np.pi / nseg
    for i in range(nside):
        for j in range(nside):
            cx, cy = (i + 0.5) * pitch, (j + 0.5) * pitch
            base = len(verts)
            for a in ang:
                verts.append([cx + radius * np.cos(a), cy + radius * np.sin(a), 0.0])
                verts.append([cx + radius * np.cos(a), cy + radius * np.sin(a), height])
            for k in range(nseg):
                a0, a1 = base + 2 * k, base + 2 * ((k + 1) % nseg)
                faces.append([a0, a1, a0 + 1])
                faces.append([a1, a1 + 1, a0 + 1])
    return np.array(verts), np.array(faces, dtype=np.int64)


def split_longest_edges(vertices, faces, n_target):
    """Geometry-preserving refinement (SURVEY.md 8d, C5 option b): the longest edge is cut at its
    midpoint, together with every triangle that shares it (no T-junctions, the surface stays
    watertight and its shape unchanged), until the mesh has n_target triangles."""
    import heapq

    verts = [tuple(v) for v in np.asarray(vertices, dtype=np.float64)]
    tris = [tuple(int(i) for i in f) for f in faces]
    alive = [True] * len(tris)
    edge_faces = {}

    def key(a, b):
        return (a, b) if a < b else (b, a)

    def length2(e):
        a, b = np.array(verts[e[0]]), np.array(verts[e[1]])
        return float(np.sum((a - b) ** 2))

    heap = []

    def add_face(fi):
        a, b, c = tris[fi]
        for e in (key(a, b), key(b, c), key(c, a)):
            if e not in edge_faces:
                edge_faces[e] = set()
                heapq.heappush(heap, (-length2(e), e))
            edge_faces[e].add(fi)

    for fi in range(len(tris)):
        add_face(fi)
    n_alive = len(tris)
    while n_alive < n_target and heap:
        _, e = heapq.heappop(heap)
        fs = [fi for fi in edge_faces.pop(e, ()) if alive[fi]]
        if not fs:
            continue
        a, b = e
        m = len(verts)
        verts.append(tuple((np.array(verts[a]) + np.array(verts[b])) / 2))
        for fi in fs:
            t = tris[fi]
            alive[fi] = False
            for ed in (key(t[0], t[1]), key(t[1], t[2]), key(t[2], t[0])):
                if ed in edge_faces:
                    edge_faces[ed].discard(fi)
            i = [k for k in range(3) if {t[k], t[(k + 1) % 3]} == {a, b}][0]
            p, q, r = t[i], t[(i + 1) % 3], t[(i + 2) % 3]  # edge p -> q is cut, winding kept
            for nt in ((p, m, r), (m, q, r)):
                tris.append(nt)
                alive.append(True)
                add_face(len(tris) - 1)
            n_alive += 1
    out = np.array([t for t, ok in zip(tris, alive) if ok], dtype=np.int64)
    return np.array(verts), out


def make_workload(name, n_walkers, n_meas=1):
    """Returns dict(kind, gradient, dt, D, substrate, label, M, T).  n_meas: measurements of the
    C3-C5 waveform (M = 1 for the rate, M = 32 as the secondary figure)."""
    from disimpy_amd import gradients, substrates

    D = 2e-9
    if name == "c2":
        gradient, dt = gradients.pgse(10e-3, 40e-3, 1000, 1e9 * np.ones(32), fib_dirs(32))
        return dict(gradient=gradient, dt=dt, D=D, substrate=substrates.free(),
                    label="C2 free diffusion, %d spins x 1000 steps, 32-direction PGSE b=1e9" % n_walkers)
    if name == "c1":
        gradient, dt = gradients.pgse(10e-3, 40e-3, 100, np.array([1e9]), np.array([[1.0, 0, 0]]))
        return dict(gradient=gradient, dt=dt, D=D, substrate=substrates.free(),
                    label="C1 free diffusion, %d spins x 100 steps, single-b PGSE" % n_walkers)
    if name == "c3":
        gradient, dt = misst_waveform(10000, np.linspace(1e9, 3e9, n_meas) if n_meas > 1 else np.array([1e9]))
        return dict(gradient=gradient, dt=dt, D=D, substrate=substrates.sphere(5e-6),
                    label="C3 impermeable sphere r=5um, %d spins x 10000 steps, M=%d" % (n_walkers, n_meas))
    if name == "c4":
        gradient, dt = misst_waveform(10000, np.linspace(1e9, 3e9, n_meas) if n_meas > 1 else np.array([1e9]))
        v, f = packed_cylinders()
        sub = substrates.mesh(v, f, True, init_pos="uniform", quiet=True)
        return dict(gradient=gradient, dt=dt, D=D, substrate=sub,
                    label="C4 packed cylinders (4096 tris, periodic), %d spins x 10000 steps, M=%d" % (n_walkers, n_meas))
    if name == "c5":
        gradient, dt = misst_waveform(10000, np.linspace(1e9, 3e9, n_meas) if n_meas > 1 else np.array([1e9]))
        m = np.load(os.path.join(ROOT, "tests", "golden", "neuron_mesh.npz"))
        sub = substrates.mesh(m["vertices"], m["faces"].astype(np.int64), True, init_pos="intra", quiet=True)
        return dict(gradient=gradient, dt=dt, D=D, substrate=sub,
                    label="C5 neuron mesh (29688 tris, periodic, intra), %d spins x 10000 steps, M=%d" % (n_walkers, n_meas))
    if name == "c5b":
        gradient, dt = misst_waveform(10000, np.linspace(1e9, 3e9, n_meas) if n_meas > 1 else np.array([1e9]))
        m = np.load(os.path.join(ROOT, "tests", "golden", "neuron_mesh.npz"))
        v, f = split_longest_edges(m["vertices"], m["faces"], 50000)
        sub = substrates.mesh(v, f, True, init_pos="intra", quiet=True)
        return dict(gradient=gradient, dt=dt, D=D, substrate=sub,
                    label="C5b neuron mesh refined to %d tris (longest edges split), periodic, intra, %d spins x 10000 "
                          "steps, M=%d" % (len(f), n_walkers, n_meas))
    raise SystemExit("unknown workload %r" % name)


def oracle_geometry(o, sub):
    """Geometry descriptor of the CPU oracle for a substrate (test infrastructure, never the product path)."""
    from oracle.oracle import FREE, MESH, SPHERE

    if sub.type == "free":
        return o.geom(FREE)
    if sub.type == "sphere":
        return o.geom(SPHERE, radius=sub.radius)
    ma = dict(vertices=sub.vertices, faces=sub.faces, xs=sub.xs, ys=sub.ys, zs=sub.zs,
              subvoxel_indices=sub.subvoxel_indices, triangle_indices=sub.triangle_indices, n_sv=sub.n_sv)
    return o.geom(MESH, mesh=ma)


def sample_positions(sub, n, positions):
    """Initial positions of walkers [0, n) for the CPU legs: the rows the GPU run uses."""
    if sub.type == "free":
        return np.zeros((n, 3))
    reps = -(-n // len(positions))
    return np.ascontiguousarray(np.tile(positions, (reps, 1))[:n])


def cpu_baseline(w, positions, budget_s=15.0):
    """The CPU oracle (a port of the reference's algorithm: oracle/dsim_oracle.c) on a
    bounded sample of the SAME workload, all host cores (OpenMP) and one core."""
    from oracle.oracle import Oracle

    o = Oracle(exact=False)
    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * w["D"] * dt))
    T = gradient.shape[1]
    cores = os.cpu_count() or 1
    g = oracle_geometry(o, sub)

    def timed(n, threads):
        pos = sample_positions(sub, n, positions)
        st = o.rng_states(n, 123)
        t0 = time.perf_counter()
        o.run(g, gradient, dt, step_l, pos, st, n_threads=threads)
        return n * T / (time.perf_counter() - t0)

    probe = timed(max(8, 64 * cores // max(1, T // 100)), cores)  # short probe to size the sample
    n_all = int(max(64, min(2_000_000, probe * budget_s * 0.6 / T)))
    rate_all = timed(n_all, cores)
    n_one = int(max(16, min(2_000_000, rate_all / cores * budget_s * 0.3 / T)))
    rate_one = timed(n_one, 1)
    return {
        "value": rate_all, "unit": "spin-steps/s", "cores": cores, "kind": "port",
        "sample": "%d walkers x %d steps of the same workload on %d OpenMP threads (oracle/dsim_oracle.c, glibc "
                  "build); single thread: %.3g spin-steps/s on %d walkers" % (n_all, T, cores, rate_one, n_one),
        "single_thread_value": rate_one,
    }  # fmt: skip


def signal_rel_err(w, positions, device, n=2048):
    """Second half of BASELINE.json's metric: the signal of the first n walkers (same seed, same
    initial positions) on the GPU against the CPU oracle, outside the timed region.  `exact`
    = the oracle build that shares the library's log/cos (bit-identical walk expected: the
    difference is summation order only); `glibc` = the build pinned to the reference's golden
    trajectory (chaotic geometries decorrelate walker by walker, so this one is statistical)."""
    from disimpy_amd import _lib, simulations
    from oracle.oracle import Oracle

    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * w["D"] * dt))
    T = gradient.shape[1]
    pos0 = sample_positions(sub, n, positions)
    with _lib.Sim(n, gradient, dt, step_l, device=device, first_walker=0) as sim:
        sim.rng_init_device(123)
        if sub.type != "free":
            sim.set_positions(pos0)
        if sub.type == "mesh":
            sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                         sub.triangle_indices)
        simulations._run_range(sim, sub, w["frames"], 1000, 1e-13, 0, T)
        c_gpu, _, _ = sim.signal()
        ph_gpu = sim.phases()
    out = {"walkers": n, "steps": T}
    for name, exact in (("exact", True), ("glibc", False)):
        o = Oracle(exact=exact)
        p = pos0.copy()
        ph, exc, _ = o.run(oracle_geometry(o, sub), gradient, dt, step_l, p, o.rng_states(n, 123),
                           n_threads=os.cpu_count() or 1)
        c, _ = o.signal(ph, exc)
        out[name] = float(np.max(np.abs(c_gpu - c) / np.maximum(np.abs(c), 1e-300)))
        if exact:
            out["phases_identical"] = bool(np.array_equal(ph, ph_gpu))
    return out


def snapshot_sample(sim, idx):
    """Per-walker state of the walkers `idx` of the shard as it stands (positions, phases, RNG states, iter_exc)."""
    return dict(pos=sim.positions()[idx], ph=sim.phases()[:, idx], st=sim.rng_states()[idx], exc=np.asarray(sim.iter_exc())[idx])


def check_sample_of_timed_run(w, positions, walkers, idx, snap_first, snap_last, n_passes):
    """The run bench.py TIMES, pinned over ALL of its passes: 512 walkers spread over rank 0's shard, read back after the
    first (warm-up) pass - the one that starts from the initial state - and again after the LAST timed pass.  The exact
    build of the CPU oracle runs those walkers alone from the same initial positions and RNG states over all T steps of
    every pass (warm-up + timed: the passes continue one walk; phases keep accumulating) and is compared at both points
    (walkers are independent, so the sample pins the full-size run with everything it goes through: walker groups,
    spatial sorts, per-walker clocks, restored order, shards cut by space); bit-identity of positions, phases, RNG states
    and iter_exc is reported as "identical".  ``walkers``: the global walker index of each sampled slot; ``idx``: the
    slots (rows of ``positions``, the shard's initial positions)."""
    from disimpy_amd import _lib
    from oracle.oracle import STATE_DTYPE, Oracle

    o = Oracle(exact=True)
    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * w["D"] * dt))
    T, M = gradient.shape[1], gradient.shape[0]
    st0 = np.concatenate([_lib.rng_init(123, int(i), 1) for i in walkers])
    st = np.zeros(len(idx), dtype=STATE_DTYPE)
    st["s0"], st["s1"] = st0[:, 0], st0[:, 1]
    pos = np.zeros((len(idx), 3)) if sub.type == "free" else np.ascontiguousarray(positions[idx], dtype=np.float64)
    ph, exc = np.zeros((M, len(idx))), np.zeros(len(idx), dtype=np.uint8)
    g = oracle_geometry(o, sub)
    t0 = time.perf_counter()

    def compare(snap):
        same = {"positions": bool(np.array_equal(pos, snap["pos"])), "phases": bool(np.array_equal(ph, snap["ph"])),
                "rng_states": bool(np.array_equal(np.stack([st["s0"], st["s1"]], axis=1), snap["st"])),
                "iter_exc": bool(np.array_equal(exc.astype(bool), snap["exc"].astype(bool)))}
        return {"result": "identical" if all(same.values()) else "DIFFERENT", "fields": same,
                "max_abs_position_difference_m": float(np.max(np.abs(pos - snap["pos"]))) if len(idx) else 0.0}

    first = last = None
    for k in range(n_passes):
        o.run(g, gradient, dt, step_l, pos, st, phases=ph, iter_exc=exc, n_threads=os.cpu_count() or 1)
        if k == 0:
            first = compare(snap_first)
        if k == n_passes - 1 and snap_last is not None:
            last = compare(snap_last)
    ok = first["result"] == "identical" and (last is None or last["result"] == "identical")
    return {"walkers": int(len(idx)), "steps": int(T), "passes": int(n_passes),
            "of": "this run: after its first pass (warm-up, from the initial state) and after its last timed pass",
            "result": "identical" if ok else "DIFFERENT", "after_first_pass": first, "after_last_pass": last,
            "oracle_s": time.perf_counter() - t0}


def end_to_end(w, n_total, devices=None):
    """Wall time of ONE simulations.simulation() call on the bench workload, set-up included (initial positions, handle,
    RNG seeding, mesh index, the T steps, the signal) - the drop-in call a user of the reference makes
    (disimpy/simulations.py:1051), against `ms_per_step` which times the steps alone with everything resident."""
    from disimpy_amd import simulations

    if devices is not None:
        os.environ["DSIM_DEVICES"] = ",".join(str(d) for d in devices)
    elif "DSIM_DEVICES" not in os.environ:
        os.environ["DSIM_DEVICES"] = "0"
    t = time.perf_counter()
    sig = simulations.simulation(n_total, w["D"], w["gradient"], w["dt"], w["substrate"], seed=123, quiet=True)
    el = time.perf_counter() - t
    return {"simulation_call_s": el, "mean_S_over_N": float(np.mean(sig) / n_total), "devices": os.environ.get("DSIM_DEVICES")}


def run_steps(sim, sub, frames, T, n_steps, dist, device, fence, M=1):
    """n_steps passes over the waveform; returns (elapsed s, walk-kernel ms per step, launches, last signal)."""
    from disimpy_amd import sharding, simulations

    walk_ms, run_ms, launches = [], [], []
    sig = None
    fence()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        if sim is not None:
            simulations._run_range(sim, sub, frames, 1000, 1e-13, 0, T)
            run_ms.append(sim.last_run_ms()[0])
            ms, nl = sim.last_walk_ms()
            try:
                busy, streams = sim.last_walk_busy_ms()
            except Exception:  # a development build (DSIM_LIB) from before the entry point existed
                busy, streams = ms, 1
            walk_ms.append((ms, busy, streams))
            launches.append(nl)
            try:
                limbs, nexc = sim.signal_limbs()  # exact integer sums of the device reduction
            except (AttributeError, TypeError):  # an older build named by DSIM_LIB + DSIM_LIB_OLD_ABI (same-box A/B): one rank only
                c, s, nexc = sim.signal()
                sig = (c, s, nexc)
                continue
        else:  # a rank without walkers still takes part in the collective
            limbs, nexc = np.zeros(4 * M, dtype=np.int64), 0
        sig = sharding.reduce_signal_exact(limbs, nexc, dist, device)
    fence()
    return time.perf_counter() - t0, walk_ms, run_ms, launches, sig


def setup_sim(args, w, n_total, a, b, device, dist=None, devices=None, part=None):
    """Untimed set-up: everything resident in HBM.  Returns (sim, positions of the shard's walkers, setup_s, ids).

    The shard: walkers [a, b) - or, ``part`` = (k, G) on a mesh, the k-th of G parts of the SPATIAL cut of all n_total
    walkers (dsim_partition_walkers: what a rank of a G-GPU job owns); ``ids`` are then its walkers, ascending (else
    None).  ``devices``: a multi-device handle over these devices (dsim_create_multi; a mesh is cut by space inside)."""
    from disimpy_amd import _lib, simulations

    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * w["D"] * dt))
    setup = {}
    t = time.perf_counter()
    # the process's first HIP work (runtime start, the library's code object onto the device) under its own name: a
    # 64-walker handle and one kernel, so that the phases below time what their names say
    tiny = _lib.Sim(64, np.zeros((1, 2, 3)), dt, step_l, device=device)
    tiny.rng_init_device(1)
    tiny.close()
    setup["device_start"] = time.perf_counter() - t
    t = time.perf_counter()
    np.random.seed(123)
    spatial = part is not None and sub.type == "mesh" and part[1] > 1
    lo, hi = (0, n_total) if spatial else (a, b)
    cache = os.environ.get("DSIM_BENCH_POSITIONS_CACHE")  # development aid: A/B runs of kernel variants on one box
    if spatial and w.get("_all_positions") is not None:  # a further part of the same cut (secondary_rates)
        positions, frames = w["_all_positions"], w.get("frames")
    elif cache and os.path.exists(cache) and sub.type == "mesh":
        positions, frames = np.load(cache)[lo:hi], None
    else:
        positions, frames = simulations._initial_positions(n_total, sub, 123, True, device, (lo, hi), dist=dist, devices=devices)
        if cache and sub.type == "mesh" and (lo, hi) == (0, n_total):
            np.save(cache, positions)
    w["frames"] = frames
    setup["initial_positions"] = time.perf_counter() - t
    ids = None
    if spatial:
        t = time.perf_counter()
        if w.get("_parts") is None or len(w["_parts"]) != part[1]:
            cost = _lib.mesh_walker_cost(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices, sub.triangle_indices,
                                         step_l, positions)
            w["_parts"], w["_all_positions"] = _lib.partition_walkers(positions, part[1], cost), positions
        ids = w["_parts"][part[0]]
        positions = np.ascontiguousarray(positions[ids])
        a, b = 0, len(ids)
        setup["partition_by_space"] = time.perf_counter() - t
    if b == a:
        return None, positions, setup, ids
    t = time.perf_counter()
    sim = _lib.Sim(b - a, gradient, dt, step_l, device=device, first_walker=a, devices=devices)
    sim.set_tuning(args.block_size, args.tile_steps, args.meas_per_pass, args.resort_steps, args.gradient_source,
                   args.mesh_traversal)
    if ids is not None:
        sim.set_walker_ids(ids)
    setup["create"] = time.perf_counter() - t
    t = time.perf_counter()
    sim.rng_init_device(123)
    setup["rng_init_device"] = time.perf_counter() - t
    t = time.perf_counter()
    if sub.type == "mesh":  # before the positions: a multi-device handle cuts its walkers by the cost the index implies
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                     sub.triangle_indices)
    setup["mesh_index_upload"] = time.perf_counter() - t
    t = time.perf_counter()
    if positions is not None:
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        if sub.type == "mesh":
            sim.partition_by_position(pos)  # a multi-device handle cuts its walkers by space; else = set_positions
        else:
            sim.set_positions(pos)
    setup["upload_positions"] = time.perf_counter() - t
    return sim, positions, setup, ids


def kernel_name(sub, M, args):
    mb = 1 << max(0, (min(M, 32) - 1).bit_length())
    if sub.type != "mesh" or args.mesh_traversal == 1:
        return "walk_kernel<%s,%d>" % (sub.type, mb)
    return "walk_kernel_coop4<%d>" % mb if mb <= 4 else "walk_kernel_coop4_traj + phase_from_rows_kernel<%d>" % mb


def profile_entry(workload):
    """Counter figures of the dominant kernel for this workload (profiles/kernel_counters.json, written from
    the rocprofv3 --pmc summaries committed beside it)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "kernel_counters.json"))).get(workload)
    except Exception:
        return None


def lib_sha256():
    """sha256 of the libdsim_hip.so this process runs (the counters in profiles/ name the build they were taken on)"""
    import hashlib

    from disimpy_amd import _lib

    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None


def secondary_rates(args, device):
    """The other BASELINE configs at a glance (rank 0 of a single-GPU run, after the timed region):
    C2 and C3 at their full sizes; C4 at the per-GPU shares of its 4- and 2-GPU configurations and the headline workload
    at the per-GPU share of the 8-GPU configuration - the shares a GPU of those jobs actually gets: the FIRST of the G
    parts of the SPATIAL cut of all 1e7 walkers (dsim_partition_walkers; rounds 1-4 ran the first n / G walkers of a
    smaller fill, a random sample of the whole substrate)."""
    out = {}
    for key, name, n, meas, parts in (("c2", "c2", 1_000_000, 1, 1), ("c3", "c3", 1_000_000, 1, 1), ("c4_share_of_4_gpus", "c4", 10_000_000, 1, 4),
                                      ("c4_share_of_2_gpus", "c4", 10_000_000, 1, 2), ("c5b_share_of_2_gpus", "c5b", 10_000_000, 1, 2),
                                      ("c5b_share_of_4_gpus", "c5b", 10_000_000, 1, 4), ("c5b_share_of_8_gpus", "c5b", 10_000_000, 1, 8)):
        if name == "c5b" and "_c5b_workload" in out:
            w = out["_c5b_workload"]  # one mesh, one fill for the three cuts
        else:
            w = make_workload(name, n, meas)
            if name == "c5b":
                out["_c5b_workload"] = w
        sub, T = w["substrate"], w["gradient"].shape[1]
        # C5b: EVERY part of the cut, one after the other - the job ends with its slowest shard; C4 is uniform: part 1 stands for all
        rows = []
        for k in range(parts if name == "c5b" else 1):
            sim, _, _, ids = setup_sim(args, w, n, 0, n, device, part=(k, parts) if parts > 1 else None)
            n_here = n if ids is None else len(ids)
            run_steps(sim, sub, w["frames"], T, 1, None, device, lambda: None)
            n_timed = 2 if name != "c5b" or parts == 8 else 1
            el, walk_ms, _, _, sig = run_steps(sim, sub, w["frames"], T, n_timed, None, device, lambda: None)
            sim.close()
            walk_rate = n_here * T / (float(np.mean([m[1] for m in walk_ms])) * 1e-3)
            rows.append({"part": k, "walkers": n_here, "value": n_here * T * n_timed / el, "seconds_per_pass": el / n_timed,
                         "walk_kernel_spin_steps_per_s": walk_rate})
        slow = max(rows, key=lambda r: r["seconds_per_pass"])
        if name == "c5b":
            # every GPU runs its part at the same time: the job takes as long as its slowest part (+ one all-reduce of 5 int64);
            # `value` = what ONE GPU of that job delivers: the job's spin-steps / the longest part / the number of GPUs
            proj = n * T / slow["seconds_per_pass"]
            out[key] = {"workload": w["label"], "cut": "the %d parts of the spatial cut by estimated cost, run one after the other on this GPU" % parts,
                        "value": proj / parts, "unit": "spin-steps/s per GPU of the %d-GPU job" % parts,
                        "projected_%d_gpu_spin_steps_per_s" % parts: proj, "slowest_part": slow, "parts": rows,
                        # the walk kernels of ALL parts: sum of their spin-steps / sum of their kernel times.  (No VALU roofline
                        # here: the counters in profiles/ are those of the whole substrate, and a part - the soma's walkers,
                        # say - spends a different number of instructions per step)
                        "walk_kernel_spin_steps_per_s": sum(r["walkers"] for r in rows) * T
                        / sum(r["walkers"] * T / r["walk_kernel_spin_steps_per_s"] for r in rows),
                        "note": "projection from the parts run one after the other on ONE GPU, not a measurement on %d" % parts}
        else:
            out[key] = {"workload": w["label"], "walkers": slow["walkers"], "cut": "spatial part 1 of %d" % parts if parts > 1 else "all",
                        "value": slow["value"], "unit": "spin-steps/s", "walk_kernel_spin_steps_per_s": slow["walk_kernel_spin_steps_per_s"],
                        "roofline": valu_roofline(name, M=w["gradient"].shape[0], walk_rate=slow["walk_kernel_spin_steps_per_s"])}
    out.pop("_c5b_workload", None)
    return out


def valu_roofline(workload, M, walk_rate):
    """The bound that holds for a register-resident fused kernel: vector-ALU issue.  SURVEY 8d's byte model does not
    apply to these configurations (C2 at 6e10 spin-steps/s x 592 B would be 4.8 x the HBM peak: the fused kernel does not
    move those bytes), so the secondary entries carry the VALU figures of their own kernels instead: wave-instructions
    per wavefront-step and the measured VALU-busy share from the committed counter run (profiles/kernel_counters.json,
    tools/pmc_profile.py) - quoted only when those counters were taken on THIS build of the library - and the issue rate
    this run's throughput implies against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction."""
    prof = profile_entry(workload) or {}
    ok = bool(prof) and prof.get("lib_sha256") == lib_sha256() and prof.get("measurements", 1) in (1, M)
    use = prof if ok else {}
    valu_ws = use.get("valu_per_wave_step")
    rate = valu_ws * walk_rate / 64.0 if valu_ws else None
    if rate is not None and rate > VALU_PEAK:  # counters that do not belong to what ran here: say nothing rather than > 1
        valu_ws, rate, ok = None, None, False
        use = {}
    return {"bound": "valu", "kernel": use.get("kernel"), "valu_per_wave_step": valu_ws, "valu_busy_measured": use.get("valu_busy"),
            "salu_per_wave_step": use.get("salu_per_wave_step"), "lds_per_wave_step": use.get("lds_per_wave_step"),
            "achieved": rate, "peak": VALU_PEAK, "unit": "wave-instr/s", "frac": (rate / VALU_PEAK) if rate else None,
            "counters_match_lib": ok, "counters_source": prof.get("source")}


def launch_ranks(n_ranks, argv, script=None):
    """`--gpus N` outside a process group: start the N ranks ourselves.  Nothing in this process has
    touched HIP or torch.cuda (no exec of a GPU-initialised process either: plain children).
    Rank r gets RANK = LOCAL_RANK = r, WORLD_SIZE = N and a free rendezvous port on 127.0.0.1;
    rank 0's stdout (the JSON line) is forwarded; any failing rank ends the job with a non-zero exit."""
    script = script or os.path.abspath(__file__)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # rank 0's JSON line is collected while ALL children are watched: a rank that dies first must not leave
    failed = None   # rank 0 blocked in a collective with nobody looking
    deadline = time.time() + 3600
    pending = set(range(len(procs)))
    while pending and time.time() < deadline:
        for r in list(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0 and failed is None:
                    failed = (r, rc)
        if failed is not None:
            break
        time.sleep(0.1)
    if failed is not None or pending:
        for r in pending:  # our own children, by PID
            procs[r].kill()
        for pr in procs:
            pr.wait()
        reader.join(5)
        sys.stdout.write(b"".join(chunks).decode())
        raise SystemExit("rank %s failed (exit code %s)" % (failed or ("?", "timeout")))
    reader.join()
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c5b", choices=["c1", "c2", "c3", "c4", "c5", "c5b"])
    ap.add_argument("--walkers", type=int, default=0, help="TOTAL walkers over all GPUs (default: the config's)")
    ap.add_argument("--measurements", type=int, default=1, help="C3-C5: number of b-values (M); C1/C2 are fixed")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only for testing the "
                         "multi-rank path with several ranks on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2/C3/C4 rates of the `secondary` object")
    ap.add_argument("--no-signal-check", action="store_true")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--tile-steps", type=int, default=0)
    ap.add_argument("--meas-per-pass", type=int, default=0)
    ap.add_argument("--gradient-source", type=int, default=0, help="1 = LDS tiles, 2 = scalar loads (0 = default)")
    ap.add_argument("--mesh-traversal", type=int, default=0, help="1 = per lane, 2 = wavefront-cooperative")
    ap.add_argument("--resort-steps", type=int, default=0, help="mesh: steps between walker re-sorts (0 auto, <0 never)")
    ap.add_argument("--mode", default="ranks", choices=["ranks", "inprocess"],
                    help="--gpus N > 1: `ranks` = one process per GPU over torch.distributed (what the driver launches); "
                         "`inprocess` = ONE process, the N devices behind one multi-device handle (dsim_create_multi) - the "
                         "path simulations.simulation() takes on a node with several GPUs.  DSIM_DEVICES (e.g. 0,0,0,0) names "
                         "the devices of the in-process mode explicitly")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the timed simulations.simulation() call (end_to_end_s)")
    args = ap.parse_args(argv)

    inprocess = args.mode == "inprocess" and "WORLD_SIZE" not in os.environ
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not inprocess:
        return launch_ranks(args.gpus, argv)
    world = 1 if inprocess else int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not inprocess:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # stdout carries ONE line, the JSON: whatever the communication libraries print while they start up (RCCL's
    # version banner, gloo's connection notes) goes to stderr - file descriptor 1 points there until the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    from disimpy_amd import _lib, sharding

    if _lib.device_count() <= 0:
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback)")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # testing aid: several ranks may share one GPU (RCCL refuses that), the reduction goes over gloo
            dist.init_process_group("gloo")
    n_dev = max(1, _lib.device_count())
    device = (local_rank if args.backend == "nccl" else local_rank % n_dev) if world > 1 else 0

    n_total = args.walkers or DEFAULT_TOTAL_WALKERS[args.workload]
    t_wl = time.perf_counter()
    w = make_workload(args.workload, n_total, args.measurements)
    t_wl = time.perf_counter() - t_wl
    sub, gradient = w["substrate"], w["gradient"]
    M, T = gradient.shape[0], gradient.shape[1]
    a, b = sharding.shard_range(n_total, rank, world)
    devices = None
    if inprocess:
        env = os.environ.get("DSIM_DEVICES")
        devices = [int(x) for x in env.split(",") if x.strip() != ""] if env else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit("--gpus %d does not match DSIM_DEVICES=%s" % (args.gpus, env))

    # a mesh job of several ranks is cut by SPACE: rank r owns the r-th part of the Morton order of all initial positions
    sim, positions, setup, ids = setup_sim(args, w, n_total, a, b, device, dist, devices=devices,
                                           part=(rank, world) if world > 1 else None)
    n_local = (b - a) if ids is None else len(ids)
    setup["workload_and_mesh_index_host"] = t_wl
    setup["total"] = sum(setup.values())
    frames = w["frames"]

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first_pass, timed_sample = None, None
    if args.warmup:
        _, _, _, _, first_pass = run_steps(sim, sub, frames, T, 1, dist, device, fence, M)
        if rank == 0 and sim is not None and not args.no_signal_check:
            # 512 walkers from all over the shard as the pass from the initial state left them: compared with the CPU
            # oracle after the timed region (check_sample_of_timed_run)
            idx = np.unique(np.linspace(0, n_local - 1, 512).astype(np.int64))
            timed_sample = [idx, snapshot_sample(sim, idx), None]
        if args.warmup > 1:
            run_steps(sim, sub, frames, T, args.warmup - 1, dist, device, fence, M)
    elapsed, walk_ms, run_ms, launches, sig = run_steps(sim, sub, frames, T, args.steps, dist, device, fence, M)
    if first_pass is None:
        first_pass = sig
    if timed_sample is not None:  # the same walkers as the LAST timed pass left them
        timed_sample[2] = snapshot_sample(sim, timed_sample[0])
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    value = float(n_total) * T * args.steps / elapsed
    # ---- roofline of the dominant kernel (rank 0's shard; every shard runs the same kernel) ----------
    # walk_ms holds, per step, (sum of the launch durations, time with at least one launch running, streams): a
    # large mesh shard runs as two halves on two streams whose launches overlap; rates follow from the busy time,
    # the average launch duration (what a kernel trace reports) from the sum
    n_launch = max(1, launches[-1]) if launches else 1
    walk_sum_step = float(np.mean([m[0] for m in walk_ms])) if walk_ms else float("nan")
    walk_ms_step = float(np.mean([m[1] for m in walk_ms])) if walk_ms else float("nan")
    streams = int(walk_ms[-1][2]) if walk_ms else 1
    avg_launch_ms = walk_sum_step / n_launch
    wave_steps_per_s = -(-n_local // 64) * T / (walk_ms_step * 1e-3)
    meas_per_launch = min(M, 32 if not args.meas_per_pass else args.meas_per_pass)
    passes = -(-M // meas_per_launch)
    steps_per_launch = T * passes / n_launch
    alg_bytes_per_launch = (80.0 + 16.0 * meas_per_launch) * n_local * steps_per_launch
    alg_gbs = alg_bytes_per_launch * n_launch / (walk_ms_step * 1e-3) / 1e9
    # ---- counters of the same kernel, collected separately (tools/pmc_profile.py -> profiles/kernel_counters.json).
    # They are only quoted when they were taken on THIS build of the library (sha256) with this run's kernel
    # selection; stale or foreign counters yield nulls, never a number.
    prof = profile_entry(args.workload) or {}
    lib_hash = lib_sha256()
    default_tuning = not (args.block_size or args.tile_steps or args.meas_per_pass or args.gradient_source or args.mesh_traversal
                          or args.resort_steps or os.environ.get("DSIM_GROUPS") or os.environ.get("DSIM_MICRO")
                          or os.environ.get("DSIM_SORT_BITS"))
    counters_match_lib = bool(prof) and prof.get("lib_sha256") == lib_hash
    counters_match_config = bool(prof) and prof.get("measurements", 1) == M and default_tuning
    use = prof if (counters_match_lib and counters_match_config) else {}
    valu_ws = use.get("valu_per_wave_step")
    traffic = use.get("hbm_bytes_per_walker_launch")  # measured HBM bytes per walker and launch (state in + out + misses)
    traffic = traffic * n_local / streams if traffic else None
    hbm_gbs = (traffic * n_launch / (walk_ms_step * 1e-3) / 1e9) if traffic else None
    valu_rate = valu_ws * wave_steps_per_s if valu_ws else None
    roofline = {
        # the contract's roofline: SURVEY 8d's algorithmic bytes of the launches of this run / the time those launches
        # took (HIP events on the library's streams), against the HBM peak
        "bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_unit": "measured HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, per walker x "
                        "this shard's walkers; null when the committed counters are not of this library build)",
        "bytes_per_spin_step": 80.0 + 16.0 * meas_per_launch, "bytes_per_launch": alg_bytes_per_launch,
        "kernel": kernel_name(sub, M, args), "launches_per_step": n_launch, "avg_launch_ms": avg_launch_ms,
        "streams": streams, "walk_busy_ms_per_step": walk_ms_step, "walk_sum_ms_per_step": walk_sum_step,
        "wave_steps_per_s": wave_steps_per_s,
        "caveat": "the algorithmic bytes are what the reference's step-synchronous kernels must move per spin-step (80 + 16 M); "
                  "the time-fused kernel keeps walker state in registers, so `frac` compares throughputs, it is not an "
                  "HBM utilisation - `measured_hbm` is (and shows no wasted traffic)",
        "measured_hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None},
        "counters_match_lib": counters_match_lib, "counters_match_config": counters_match_config, "lib_sha256": lib_hash,
        "counters_source": prof.get("source"),
        # what actually bounds the kernel: vector-ALU time.  valu_busy_measured = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x
        # kernel cycles) from the counter run; issue_frac prices every instruction at 4 cycles (the fp64 rate; f32 / integer
        # instructions of several waves issue in 2), so it overstates - the measured figure is the one to quote
        "valu": {"valu_busy_measured": use.get("valu_busy"), "valu_per_wave_step": valu_ws,
                 "salu_per_wave_step": use.get("salu_per_wave_step"), "lds_per_wave_step": use.get("lds_per_wave_step"),
                 "wait_any_frac": use.get("wait_any_frac"),
                 "issue_rate": valu_rate, "issue_peak": VALU_PEAK, "issue_frac": (valu_rate / VALU_PEAK) if valu_rate else None,
                 "issue_unit": "wave-instr/s at 4 cycles per wave64 instruction"},
        "note": "register-resident fused kernel: bound by vector-ALU time and dependent index look-ups, not by "
                "HBM.  Deviations from the north star's sketch, measured and kept: the mesh kernel reads triangle "
                "records through L1/L2 (an index of tens of MB - the fine grid of overlapping cells - read by Morton-sorted walkers) instead of staging lists in LDS, and "
                "with M <= 4 it reads the gradient samples from global memory (32 B per measurement and step out of L1) instead "
                "of an LDS tile; LDS holds the log/cos tables, the subvoxel boundaries and the wave work queues.",
    }
    out = {
        "metric": "spin-steps/sec", "value": value, "unit": "spin-steps/s", "n_gpus": args.gpus if inprocess else world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": w["label"], "total_walkers": n_total, "walkers_per_gpu": n_local, "time_steps": T,
            "measurements": M, "substrate": sub.type, "triangles": int(len(sub.faces)) if sub.type == "mesh" else 0,
            "seed": 123, "mode": "inprocess (dsim_create_multi over devices %s)" % devices if inprocess else "ranks",
            "n_devices": len(devices) if inprocess else world,
            "sharding": ("walkers cut by space (Morton order of the initial positions in equal parts)" if sub.type == "mesh" else
                         "balanced walker ranges") + (", exact integer signal sums added inside the handle" if inprocess else
                                                       ", 1 all-reduce of 4M+1 int64 (exact signal sums) per step (%s)" % args.backend),
        },
        "roofline": roofline,
        "walk_kernel_spin_steps_per_s": n_local * T / (walk_ms_step * 1e-3),
        "run_spin_steps_per_s": n_local * T / (float(np.mean(run_ms)) * 1e-3) if run_ms else None,
        "setup_s": setup,
        # first pass over the waveform starts from the initial state: free diffusion expects exp(-bD)
        "signal_check": {"first_pass_mean_S_over_N": float(np.mean(first_pass[0]) / n_total),
                         "last_pass_mean_S_over_N": float(np.mean(sig[0]) / n_total)},
    }  # fmt: skip
    if sim is not None:
        sim.close()
    if rank == 0 and positions is None:
        positions = np.zeros((1, 3))
    if rank == 0 and not args.no_signal_check:
        out["signal_rel_err"] = signal_rel_err(w, positions, device)
        out["signal_rel_err"]["sample_of_timed_run"] = (
            check_sample_of_timed_run(w, positions, (a + timed_sample[0]) if ids is None else ids[timed_sample[0]], timed_sample[0],
                                      timed_sample[1], timed_sample[2], args.warmup + args.steps)
            if timed_sample is not None else "not taken (needs --warmup >= 1: the first pass starts from the initial state)")
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(w, positions)
    if rank == 0 and not args.no_end_to_end and world == 1:
        out["end_to_end_s"] = end_to_end(w, n_total, devices)
        out["end_to_end_s"]["timed_region_equivalent_s"] = elapsed / args.steps
    if rank == 0 and not args.no_secondary and world == 1 and not inprocess:
        out["secondary"] = secondary_rates(args, device)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    return 0


if __name__ == "__main__":
    sys.exit(main())
