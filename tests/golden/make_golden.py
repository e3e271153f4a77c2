"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run ONCE in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

What it does
------------
* converts the data files held by the reference's own test-suite
  (/root/reference/disimpy/tests/*.txt|*.pkl|*.npy) into small .npy/.npz
  fixtures (data only - no reference source text is copied);
* imports the reference's pure-numpy host modules (gradients.py, utils.py and
  substrates.py - the latter with a 3-line identity stub standing in for
  numba.jit, which is absent from this image) and records their OUTPUTS for
  fixed inputs.  These outputs pin this repo's own re-implementations.

disimpy/simulations.py itself cannot be imported here (needs numba + CUDA), so
the step kernels are pinned by the reference's golden trajectory file instead
(test_traj.txt -> traj_free_seed123.npy).

Nothing under tests/, bench.py or the package reads /root/reference at run time.
"""
import os
import pickle
import sys
import types

import numpy as np

REF = "/root/reference"
REF_TESTS = os.path.join(REF, "disimpy", "tests")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference_host_modules():
    """Import gradients/utils/substrates of the reference without running
    disimpy/__init__.py (which imports numba.cuda)."""
    if not hasattr(np, "trapz"):
        np.trapz = np.trapezoid
    numba_stub = types.ModuleType("numba")
    numba_stub.jit = lambda *a, **k: (lambda f: f)
    numba_stub.boolean = np.bool_
    sys.modules.setdefault("numba", numba_stub)
    # utils.py imports matplotlib at module level; headless backend
    import matplotlib

    matplotlib.use("Agg")
    pkg = types.ModuleType("disimpy")
    pkg.__path__ = [os.path.join(REF, "disimpy")]
    sys.modules["disimpy"] = pkg
    import importlib

    utils = importlib.import_module("disimpy.utils")
    gradients = importlib.import_module("disimpy.gradients")
    substrates = importlib.import_module("disimpy.substrates")
    return gradients, utils, substrates


def _load_pkl(name):
    with open(os.path.join(REF_TESTS, name), "rb") as f:
        return pickle.load(f)


def fib_dirs(n):
    """n golden-spiral unit vectors."""
    k = np.arange(n)
    z = 1 - (2 * k + 1) / n
    phi = k * np.pi * (3 - np.sqrt(5))
    r = np.sqrt(1 - z**2)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def main():
    gradients, utils, substrates = _import_reference_host_modules()

    # ---- data files of the reference test-suite --------------------------------
    traj = np.loadtxt(os.path.join(REF_TESTS, "test_traj.txt"))
    assert traj.shape == (1000, 30)
    np.save(os.path.join(OUT, "traj_free_seed123.npy"), traj)

    misst = {}
    for geom in ("cylinder", "sphere"):
        for sd in ("1ms", "30ms"):
            fn = f"misst_{geom}_signal_smalldelta_{sd}_bigdelta_40ms_radius_5um.txt"
            misst[f"{geom}_{sd}"] = np.loadtxt(os.path.join(REF_TESTS, fn))
    np.savez(os.path.join(OUT, "misst_signals.npz"), **misst)

    for name in ("sphere_mesh", "cylinder_mesh_closed", "cylinder_mesh_open"):
        m = _load_pkl(name + ".pkl")
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            vertices=np.asarray(m["vertices"], dtype=np.float64),
            faces=np.asarray(m["faces"], dtype=np.int64),
        )
    m = _load_pkl("neuron-model.pkl")
    assert m["vertices"].dtype == np.float32
    np.savez_compressed(
        os.path.join(OUT, "neuron_mesh.npz"),
        vertices=m["vertices"],  # float32, as shipped
        faces=m["faces"].astype(np.int32),
    )
    for name in ("desired_triangle_indices.npy", "desired_subvoxel_indices.npy"):
        np.save(os.path.join(OUT, name), np.load(os.path.join(REF_TESTS, name)))

    # ---- outputs of the reference's gradients.py / utils.py --------------------
    g = {}
    rng = np.random.RandomState(2024)
    # example waveform of the reference tests (test_gradients.py / test_simulations.py)
    T = 80e-3
    ex = np.zeros((1, 100, 3))
    ex[0, 1:11, 0] = 1
    ex[0, -11:-1, 0] = -1
    ex_dt = T / (ex.shape[1] - 1)
    g["ex_gradient"] = ex
    g["ex_dt"] = ex_dt
    g["ex_calc_b"] = gradients.calc_b(ex, ex_dt)
    g["ex_calc_q"] = gradients.calc_q(ex, ex_dt)
    ig, idt = gradients.interpolate_gradient(ex, ex_dt, 777)
    g["ex_interp777"] = ig
    g["ex_interp777_dt"] = idt
    bs = np.linspace(1, 2e9, 7)
    ex7 = np.concatenate([ex for _ in bs], axis=0)
    g["ex_set_b_in_bs"] = bs
    g["ex_set_b"] = gradients.set_b(ex7, ex_dt, bs)
    Rs = np.zeros((7, 3, 3))
    vk = rng.normal(size=(7, 2, 3))
    for i in range(7):
        Rs[i] = utils.vec2vec_rotmat(vk[i, 0], vk[i, 1])
    g["v2v_in"] = vk
    g["v2v_out"] = Rs
    g["v2v_parallel"] = utils.vec2vec_rotmat(np.array([1.0, 0, 0]), np.array([2.0, 0, 0]))
    g["v2v_antiparallel"] = utils.vec2vec_rotmat(np.array([1.0, 0, 0]), np.array([-1.0, 0, 0]))
    g["ex_rotate"] = gradients.rotate_gradient(g["ex_set_b"], Rs)
    # BASELINE config C1: pgse(10 ms, 40 ms, 100 pts, b=1e9 along x)
    c1, c1dt = gradients.pgse(10e-3, 40e-3, 100, np.array([1e9]), np.array([[1.0, 0, 0]]))
    g["c1_pgse"] = c1
    g["c1_dt"] = c1dt
    # BASELINE config C2: 32-direction PGSE, 1000 points; keep 4 of the 32 directions
    dirs = fib_dirs(32)
    c2, c2dt = gradients.pgse(10e-3, 40e-3, 1000, 1e9 * np.ones(32), dirs)
    g["c2_dirs"] = dirs
    g["c2_pgse_rows_0_7_19_31"] = c2[[0, 7, 19, 31]]
    g["c2_dt"] = c2dt
    g["c2_calc_b"] = gradients.calc_b(c2, c2dt)
    # MISST waveform of the reference tests, interpolated to 1000 points, 5 b-values
    T = 70e-3
    mw = np.zeros((1, 700, 3))
    mw[0, 1:300, 0] = 1
    mw[0, -300:-1, 0] = -1
    mdt = T / (mw.shape[1] - 1)
    mb = np.linspace(1, 3e9, 5)
    mw5 = np.concatenate([mw for _ in mb], axis=0)
    mi, midt = gradients.interpolate_gradient(mw5, mdt, 1000)
    g["misst_interp_dt"] = midt
    g["misst_set_b"] = gradients.set_b(mi, midt, mb)
    np.savez_compressed(os.path.join(OUT, "gradients_golden.npz"), **g)

    # ---- outputs of the reference's substrates.py ------------------------------
    s = {}
    cases = []
    sm = _load_pkl("sphere_mesh.pkl")
    cm = _load_pkl("cylinder_mesh_closed.pkl")
    om = _load_pkl("cylinder_mesh_open.pkl")
    for tag, mesh in (("sphere", sm), ("cylc", cm), ("cylo", om)):
        for periodic in (True, False):
            for pad in (0.0, 1e-6):
                for n_sv in ([1, 1, 1], [1, 5, 20], [10, 10, 10], [2, 5, 10]):
                    cases.append((tag, mesh, periodic, pad, n_sv))
    for tag, mesh, periodic, pad, n_sv in cases:
        sub = substrates.mesh(
            mesh["vertices"],
            mesh["faces"],
            periodic,
            padding=np.zeros(3) + pad,
            n_sv=np.array(n_sv),
            quiet=True,
        )
        key = f"{tag}_p{int(periodic)}_pad{int(pad * 1e6)}_sv{'x'.join(map(str, n_sv))}"
        s[key + "_tri"] = sub.triangle_indices.astype(np.int32)
        s[key + "_sv"] = sub.subvoxel_indices.astype(np.int32)
        s[key + "_voxel"] = sub.voxel_size
        s[key + "_nverts"] = np.array(sub.vertices.shape[0])
        s[key + "_nfaces"] = np.array(sub.faces.shape[0])
        s[key + "_vsum"] = sub.vertices.sum(axis=0)
    # unit-level known answers
    tri = np.array([[0.5, 0.5, 0.5], [1.5, 0.5, 0.5], [0.5, 1.5, 0.5]])
    boxes = rng.uniform(-1, 2, size=(200, 2, 3))
    boxes.sort(axis=1)
    s["sat_tri"] = tri
    s["sat_boxes"] = boxes
    s["sat_out"] = np.array([substrates._triangle_box_overlap(tri, b) for b in boxes])
    xs = np.linspace(0, 1, 11)
    iv = rng.uniform(-0.3, 1.3, size=(300, 2))
    iv[:20] = np.round(iv[:20], 1)  # exercise the boundary-equality rules
    s["interval_xs"] = xs
    s["interval_in"] = iv
    s["interval_out"] = np.array([substrates._interval_sv_overlap(xs, a, b) for a, b in iv])
    av, af = substrates._aabb_to_mesh(np.array([0.0, 0.1, 0.2]), np.array([1.0, 2.0, 3.0]))
    s["aabb_vertices"] = av
    s["aabb_faces"] = af
    np.savez_compressed(os.path.join(OUT, "substrates_golden.npz"), **s)

    # neuron at the default 50^3 (about 15-20 s in pure Python)
    nm = _load_pkl("neuron-model.pkl")
    sub = substrates.mesh(nm["vertices"], nm["faces"], True, quiet=True)
    np.savez_compressed(
        os.path.join(OUT, "neuron_index_50.npz"),
        triangle_indices=sub.triangle_indices.astype(np.int32),
        subvoxel_counts=(sub.subvoxel_indices[:, 1] - sub.subvoxel_indices[:, 0]).astype(np.int16),
        voxel_size=sub.voxel_size,
    )
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
