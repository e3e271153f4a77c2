"""bench.py's synthetic workloads (CPU): the C5b refinement keeps the neuron's surface intact."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _edge_counts(faces):
    c = collections.Counter()
    for a, b, cc in faces:
        for e in ((a, b), (b, cc), (cc, a)):
            c[(min(e), max(e))] += 1
    return collections.Counter(c.values())


def _area(v, f):
    A, B, C = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    return 0.5 * np.linalg.norm(np.cross(B - A, C - A), axis=1).sum()


def _volume(v, f):
    A, B, C = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    return np.einsum("ij,ij->i", A, np.cross(B, C)).sum() / 6


def test_longest_edge_refinement_preserves_the_surface():
    m = np.load(os.path.join(ROOT, "tests", "golden", "neuron_mesh.npz"))
    v0, f0 = m["vertices"].astype(np.float64), m["faces"].astype(np.int64)
    v, f = bench.split_longest_edges(v0, f0, 33000)
    assert len(f) == 33000 and f.max() < len(v)
    # same manifold structure: every edge still has the same number of adjacent triangles
    assert set(_edge_counts(f)) == set(_edge_counts(f0))
    np.testing.assert_allclose(_area(v, f), _area(v0, f0), rtol=1e-9)
    np.testing.assert_allclose(_volume(v, f), _volume(v0, f0), rtol=1e-9)  # winding kept
    longest = lambda vv, ff: max(np.linalg.norm(vv[ff[:, i]] - vv[ff[:, (i + 1) % 3]], axis=1).max() for i in range(3))  # noqa: E731
    assert longest(v, f) < longest(v0, f0)


def test_fib_dirs_are_unit_vectors_and_packed_cylinders_have_4096_triangles():
    d = bench.fib_dirs(32)
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-12)
    v, f = bench.packed_cylinders()
    assert f.shape == (4096, 3) and f.max() < len(v)


def test_bench_starts_its_own_ranks(tmp_path, capfd):
    """`python bench.py --gpus N` outside a process group is a launcher (SURVEY 8e / VERDICT r1 #2): N child
    ranks with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*, rank 0's JSON line forwarded, failures propagated."""
    import json
    import subprocess

    import pytest

    child = tmp_path / "rank.py"
    child.write_text(
        "import json, os, sys\n"
        "r = int(os.environ['RANK'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "int(os.environ['MASTER_PORT'])\n"
        "if '--fail' in sys.argv and r == 1: sys.exit(3)\n"
        "print(json.dumps({'rank': r, 'world': int(os.environ['WORLD_SIZE']), 'argv': sys.argv[1:]}))\n"
    )
    assert bench.launch_ranks(3, ["--gpus", "3"], script=str(child)) == 0
    line = json.loads(capfd.readouterr().out.strip().splitlines()[-1])
    assert line == {"rank": 0, "world": 3, "argv": ["--gpus", "3"]}  # only rank 0's line comes through
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(2, ["--fail"], script=str(child))
    assert "rank" in str(e.value)
    # the real entry point: without a GPU every rank refuses to run and the launcher reports it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, env={k: v for k, v in os.environ.items()
                                                                               if k not in ("WORLD_SIZE", "RANK")})
    from disimpy_amd import _lib

    if _lib.device_count() <= 0:
        assert r.returncode != 0 and "failed" in r.stderr


def test_strong_scaling_split_is_balanced():
    from disimpy_amd.sharding import shard_range

    for n, world in ((10_000_000, 8), (5, 4), (1, 8), (7, 3)):
        sizes = [b - a for a, b in (shard_range(n, r, world) for r in range(world))]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
