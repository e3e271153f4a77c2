#!/usr/bin/env python3
"""Headline benchmark: spin-steps/s of the fused random-walk kernels.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5|c5b]

One "step" = one pass of the hot path over one batch: every walker of this
rank's shard is advanced through the whole gradient waveform (T time steps, one
fused kernel launch per block of <= 32 measurements), the signal is reduced on
the device and - for N > 1 - all-reduced over RCCL.  Default workload = BASELINE
config C2: free diffusion, 1e6 spins x 1e3 steps, 32-direction PGSE.  Walker
shards are weak-scaled: each rank owns 1e6 walkers with global indices
[rank*1e6, (rank+1)*1e6), so RNG states and results are partition-independent.
Inputs (gradient tiles, walker state, mesh index) are resident in HBM before
the timed region; successive steps continue the same walk.

Prints ONE JSON line (rank 0).  `roofline.achieved` is the ALGORITHMIC byte
rate (80 + 16 M bytes per spin-step, the state the reference's
step-synchronous kernels round-trip through HBM) over the kernel time measured
with HIP events on the library's stream; `roofline.traffic` is the HBM traffic
measured with rocprofv3 PMC counters (profiles/), far below it because the
fused kernel keeps walker state in registers.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def cpu_baseline(w, budget_s=15.0):
    """The CPU oracle (a port of the reference's algorithm: oracle/dsim_oracle.c) on a
    bounded sample of the SAME workload, all host cores (OpenMP) and one core."""
    from disimpy_amd import simulations
    from oracle.oracle import CYLINDER, ELLIPSOID, FREE, MESH, SPHERE, Oracle  # noqa: F401

    o = Oracle(exact=False)
    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    step_l = float(np.sqrt(6 * w["D"] * dt))
    T = gradient.shape[1]
    cores = os.cpu_count() or 1

    def geom_and_pos(n):
        if sub.type == "free":
            return o.geom(FREE), np.zeros((n, 3))
        if sub.type == "sphere":
            return o.geom(SPHERE, radius=sub.radius), simulations._fill_sphere(n, sub.radius, simulations._host_rng(123))
        ma = dict(vertices=sub.vertices, faces=sub.faces, xs=sub.xs, ys=sub.ys, zs=sub.zs,
                  subvoxel_indices=sub.subvoxel_indices, triangle_indices=sub.triangle_indices, n_sv=sub.n_sv)
        g = o.geom(MESH, mesh=ma)
        if isinstance(sub.init_pos, str) and sub.init_pos == "intra":
            pos = o.fill_mesh(n, g, sub.voxel_size, True, 123)
        else:
            pos = np.random.RandomState(123).random_sample((n, 3)) * sub.voxel_size
        return g, pos

    def timed(n, threads):
        g, pos = geom_and_pos(n)
        st = o.rng_states(n, 123)
        pos = np.ascontiguousarray(pos)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "c5b"])
    ap.add_argument("--walkers", type=int, default=0, help="walkers per GPU (default: the config's)")
    ap.add_argument("--measurements", type=int, default=1, help="C3-C5: number of b-values (M); C1/C2 are fixed")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only for testing the "
                         "multi-rank path with several ranks on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--tile-steps", type=int, default=0)
    ap.add_argument("--meas-per-pass", type=int, default=0)
    ap.add_argument("--gradient-source", type=int, default=0, help="1 = LDS tiles, 2 = scalar loads (0 = default)")
    ap.add_argument("--mesh-traversal", type=int, default=0, help="1 = per lane, 2 = wavefront-cooperative")
    ap.add_argument("--resort-steps", type=int, default=0, help="mesh: steps between walker re-sorts (0 auto, <0 never)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import torch

    from disimpy_amd import _lib, sharding, simulations

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

    default_walkers = {"c1": 1000, "c2": 1_000_000, "c3": 1_000_000, "c4": 1_000_000, "c5": 1_000_000, "c5b": 1_000_000}
    n_local = args.walkers or default_walkers[args.workload]
    n_total = n_local * world
    w = make_workload(args.workload, n_local, args.measurements)
    sub, gradient, dt = w["substrate"], w["gradient"], w["dt"]
    M, T = gradient.shape[0], gradient.shape[1]
    step_l = float(np.sqrt(6 * w["D"] * dt))
    a, b = sharding.shard_range(n_total, rank, world)
    assert b - a == n_local

    # ---- untimed set-up: everything resident in HBM ------------------------------
    t_setup = time.perf_counter()
    np.random.seed(123)
    positions, frames = simulations._initial_positions(n_total, sub, 123, True, device)
    sim = _lib.Sim(n_local, gradient, dt, step_l, device=device, first_walker=a)
    sim.set_tuning(args.block_size, args.tile_steps, args.meas_per_pass, args.resort_steps, args.gradient_source,
                   args.mesh_traversal)
    t_rng = time.perf_counter()
    sim.rng_init_device(123)
    t_rng = time.perf_counter() - t_rng
    if positions is not None:
        sim.set_positions(np.ascontiguousarray(positions[a:b], dtype=np.float64))
    if sub.type == "mesh":
        sim.set_mesh(sub.vertices, sub.faces, sub.xs, sub.ys, sub.zs, sub.n_sv, sub.subvoxel_indices,
                     sub.triangle_indices)
    t_setup = time.perf_counter() - t_setup

    kernel_ms = []
    launches = []

    def one_step():
        simulations._run_range(sim, sub, frames, 1000, 1e-13, 0, T)
        ms, nl = sim.last_run_ms()
        kernel_ms.append(ms)
        launches.append(nl)
        c, s, nexc = sim.signal()
        return sharding.reduce_signal(c, s, nexc, dist, device)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first_pass = None
    for _ in range(args.warmup):
        r_ = one_step()
        if first_pass is None:
            first_pass = r_
    kernel_ms.clear()
    launches.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sig = one_step()
        if first_pass is None:
            first_pass = sig
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    spin_steps = float(n_total) * T * args.steps
    value = spin_steps / elapsed
    # roofline of the dominant kernel (walk_kernel): algorithmic bytes per launch / launch duration
    n_launch = max(1, launches[-1])
    meas_per_launch = min(M, 32 if not args.meas_per_pass else args.meas_per_pass)
    # a launch covers all walkers, meas_per_launch measurements and T * passes / n_launch time steps
    # (mesh runs are cut into one launch per re-sort interval; M > 32 needs several passes)
    mb_pow2 = 1 << max(0, (meas_per_launch - 1).bit_length())
    passes = -(-M // meas_per_launch)
    steps_per_launch = T * passes / n_launch
    bytes_per_launch = (80.0 + 16.0 * meas_per_launch) * n_local * steps_per_launch
    avg_launch_ms = float(np.mean(kernel_ms)) / n_launch
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    traffic = None
    valu = None  # the bound that actually holds (rocprofv3 SQ counters of the same workload, profiles/)
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            if n_local == default_walkers[args.workload] and not args.meas_per_pass:
                prof = json.load(open(tfile)).get(args.workload, {})
                traffic = prof.get("bytes_per_launch")
                valu = prof.get("valu")
        except Exception:
            traffic = None

    out = {
        "metric": "spin-steps/sec", "value": value, "unit": "spin-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": w["label"], "walkers_per_gpu": n_local, "time_steps": T, "measurements": M,
            "substrate": sub.type, "seed": 123, "sharding": "walker ranges, 1 RCCL all-reduce of 2M+1 f64 per step",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 "
            "FETCH_SIZE x2 + WRITE_SIZE, separate passes, calibrated: profiles/hbm_traffic.json)",
            "traffic_GBps": (traffic / (avg_launch_ms * 1e-3) / 1e9) if traffic else None,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "kernel": ("walk_kernel<%s,%d>" % (sub.type, mb_pow2) if sub.type != "mesh" or args.mesh_traversal == 1
                       else "walk_kernel_coop4<%d>" % mb_pow2 if mb_pow2 <= 2
                       else "walk_kernel_coop4_traj + phase_from_traj_kernel<%d>" % mb_pow2),
            "algorithmic_bytes_per_spin_step": 80.0 + 16.0 * meas_per_launch,
            "launches_per_step": n_launch, "avg_launch_ms": avg_launch_ms, "valu": valu,
            "note": "time-fused kernel: walker state stays in registers, so real HBM traffic (traffic) is far "
                    "below the algorithmic bytes of the reference's step-synchronous formulation; the kernel "
                    "is fp64-VALU bound",
        },
        "kernel_spin_steps_per_s": n_local * T / (float(np.mean(kernel_ms)) * 1e-3),
        "setup_s": {"total": t_setup, "rng_init_device": t_rng},
        # first pass over the waveform starts from the initial state: C2 expects exp(-bD) = exp(-2) = 0.1353
        "signal_check": {"first_pass_mean_S_over_N": float(np.mean(first_pass[0]) / n_total),
                         "last_pass_mean_S_over_N": float(np.mean(sig[0]) / n_total)},
    }  # fmt: skip
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(w)
    sim.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
