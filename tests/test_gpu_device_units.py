"""The HIP DEVICE functions, one by one, on the GPU (run with ``-m gpu``).

The reference tests each of its device functions through a wrapper kernel
(disimpy/tests/test_simulations.py:23-360).  ``libdsim_unit.so`` (a test-only build of
disimpy_amd/csrc/dsim_unit.hip, not part of the product ABI) wraps the functions of
dsim_device.h / dsim_math.h the same way; here they are compared

(i)  bit for bit with the CPU oracle's restatement of the reference (``dso_ua_eval``, exact
     build) on >= 1e6 random inputs plus edge inputs (zero / infinite / NaN lengths, arguments
     below 2^-767 where ``ds_sqrt_dev`` leaves its fast path, subnormal quotients,
     boundary-equal coordinates) - this is what shows that dsim_math.h gives the same bits on
     gfx950 as on x86-64, which every walker-level identity test silently relies on;
(ii) with the known answers the reference's own tests assert.
"""
import ctypes as C
import os

import numpy as np
import numpy.testing as npt
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

UNIT_LIB = os.path.join(ROOT, "disimpy_amd", "csrc", "libdsim_unit.so")
_f64p = C.POINTER(C.c_double)
N = 1 << 20


@pytest.fixture(scope="module")
def dev():
    assert os.path.exists(UNIT_LIB), "build it with `make -C disimpy_amd/csrc libdsim_unit.so` (__graft_entry__.build() does)"
    lib = C.CDLL(UNIT_LIB)
    lib.dsu_last_error.restype = C.c_char_p
    lib.dsu_eval.argtypes = [C.c_int, C.c_longlong, C.c_int, _f64p, C.c_int, _f64p, _f64p, C.c_int]

    def run(op, cols, n_out, xs=None):
        a = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64) for c in cols]))
        out = np.zeros((n_out, a.shape[1]))
        xs_ = None if xs is None else np.ascontiguousarray(xs, dtype=np.float64)
        rc = lib.dsu_eval(op, a.shape[1], a.shape[0], a.ctypes.data_as(_f64p), n_out, out.ctypes.data_as(_f64p),
                          None if xs_ is None else xs_.ctypes.data_as(_f64p), 0 if xs_ is None else len(xs_))
        assert rc == 0, lib.dsu_last_error().decode()
        return out

    return run


@pytest.fixture(scope="module")
def host(oracle_exact):
    lib = oracle_exact.lib
    lib.dso_ua_eval.argtypes = [C.c_int, C.c_int64, C.POINTER(_f64p), C.POINTER(_f64p), _f64p, C.c_int64]

    def run(op, cols, n_out, xs=None):
        a = [np.ascontiguousarray(c, dtype=np.float64) for c in cols]
        out = [np.zeros(len(a[0])) for _ in range(n_out)]
        pin = (_f64p * len(a))(*[c.ctypes.data_as(_f64p) for c in a])
        pout = (_f64p * n_out)(*[c.ctypes.data_as(_f64p) for c in out])
        xs_ = None if xs is None else np.ascontiguousarray(xs, dtype=np.float64)
        rc = lib.dso_ua_eval(op, len(a[0]), pin, pout, None if xs_ is None else xs_.ctypes.data_as(_f64p),
                             0 if xs_ is None else len(xs_))
        assert rc == 0
        return np.stack(out)

    return run


def _same_bits(a, b):
    """bit-for-bit equality, any NaN == any NaN (payloads may differ between fp units)"""
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | nan))


def _bits(u64):
    return np.asarray(u64, dtype=np.uint64).view(np.float64)


def _vecs(rng, n, k=3, scale=1.0):
    return [(rng.random_sample(n) - 0.5) * scale for _ in range(k)]


# ------------------------------------------------------------------ dsim_math.h on the device


def test_logf_on_device_equals_host_and_is_correctly_rounded(dev, host):
    rng = np.random.RandomState(1)
    u = (rng.randint(0, 2**53, N).astype(np.float64) * 2.0**-53).astype(np.float32)  # the uniforms Box-Muller sees
    edge = np.array([0.0, 2.0**-149, 2.0**-126, 2.0**-53, 2.0**-24, 0.5, np.nextafter(np.float32(1), np.float32(0)), 1.0,
                     1.5, 3.0e38, np.inf], dtype=np.float32)
    x = np.concatenate([u, edge, np.float32(2.0) ** rng.randint(-149, 128, 4096).astype(np.float32) *
                        (1 + rng.random_sample(4096).astype(np.float32))]).astype(np.float32).astype(np.float64)
    got, want = dev(0, [x], 1)[0], host(0, [x], 1)[0]
    assert _same_bits(got, want)
    # normal float32 arguments (the uniforms are 0 or >= 2^-53): the float64 logarithm rounded once to float32,
    # i.e. the correctly rounded logf - what CUDA's libdevice returns on all but ~1e-7 of the inputs
    nrm = np.isfinite(x) & (x >= 2.0**-126)
    npt.assert_array_equal(got[nrm], np.log(x[nrm]).astype(np.float32).astype(np.float64))
    assert got[x == 0][0] == -np.inf  # u1 == 0: the reference does not guard it either


def test_cos_on_device_equals_host(dev, host):
    rng = np.random.RandomState(2)
    u2 = (rng.randint(0, 2**53, N).astype(np.float64) * 2.0**-53).astype(np.float32).astype(np.float64)
    x = np.concatenate([6.283185307179586 * u2, [0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 6.283185307179586, 1e-300, 5e-324],
                        np.arange(257) * (np.pi / 128)])
    got, want = dev(1, [x], 1)[0], host(1, [x], 1)[0]
    assert _same_bits(got, want)
    assert np.abs(got - np.cos(x)).max() < 3e-16


def test_div3_on_device_equals_host_and_ieee_division(dev, host):
    rng = np.random.RandomState(3)
    x, y, z = _vecs(rng, N, 3, 4.0)
    ln = np.sqrt(x * x + y * y + z * z)
    # edge rows: zero length (0/0 and x/0), infinite length, subnormal quotients, huge/tiny ratios
    ex = np.array([0.0, 1.0, -1.0, 1e-310, 3e-308, 1e300, 1.0, 5e-324, 1.0, np.inf, np.nan])
    el = np.array([0.0, 0.0, np.inf, 3.0, 1e10, 1e-300, 1e308, 7.0, 5e-324, 2.0, 1.0])
    x, y, z, ln = np.concatenate([x, ex]), np.concatenate([y, ex[::-1]]), np.concatenate([z, -ex]), np.concatenate([ln, el])
    got, want = dev(2, [x, y, z, ln], 3), host(2, [x, y, z, ln], 3)
    assert _same_bits(got, want)
    with np.errstate(all="ignore"):
        ieee = np.stack([x / ln, y / ln, z / ln])
    assert _same_bits(got[:, :N], ieee[:, :N])  # the reference divides three times (simulations.py:70-73)


def test_sqrt_dev_is_correctly_rounded_everywhere(dev, host):
    rng = np.random.RandomState(4)
    x = np.concatenate([
        -2.0 * np.log(rng.random_sample(N)),                    # Box-Muller radii
        rng.random_sample(4096) * 2.0 ** rng.randint(-1074, 1024, 4096),  # the whole exponent range
        [0.0, -0.0, 5e-324, 2.0**-1022, 2.0**-767, np.nextafter(2.0**-767, 0), 2.0**-766, 1.0, 4.0, 1.7976931348623157e308,
         np.inf, -1.0, np.nan]])
    got = dev(3, [x], 1)[0]
    with np.errstate(all="ignore"):
        assert _same_bits(got, np.sqrt(x))
    assert _same_bits(got, host(3, [x], 1)[0])


def test_u64_to_unit_f64_is_exact(dev, host):
    rng = np.random.RandomState(5)
    u = np.concatenate([rng.randint(0, 2**63, N, dtype=np.uint64) * np.uint64(2) + rng.randint(0, 2, N).astype(np.uint64),
                        np.array([0, 1, 2047, 2048, 2**11 - 1, 2**43, 2**43 - 1, 2**64 - 1, 2**63, 2**32, 2**32 - 1], dtype=np.uint64)])
    got = dev(4, [_bits(u)], 1)[0]
    want = (u >> np.uint64(11)).astype(np.float64) * 2.0**-53  # numba/cuda/random.py:128-139
    assert _same_bits(got, want) and _same_bits(got, host(4, [_bits(u)], 1)[0])
    assert got.max() < 1.0 and got.min() >= 0.0


# --------------------------------------------------------------------- RNG and the random step


def test_random_step_and_uniforms_on_device(dev, host, oracle_exact):
    st = oracle_exact.rng_states(N // 4, 12)
    cols = [_bits(st["s0"]), _bits(st["s1"])]
    got, want = dev(15, cols, 5), host(15, cols, 5)
    assert _same_bits(got, want)  # three normals, normalisation, and the advanced state
    npt.assert_almost_equal(np.linalg.norm(got[:3], axis=0), 1.0)  # test_simulations.py:139
    assert abs(got[:3].mean()) < 3e-3
    g2, w2 = dev(16, cols, 4), host(16, cols, 4)
    assert _same_bits(g2, w2)
    assert np.all(g2[0] == g2[0].astype(np.float32))  # float32 uniforms (numba/cuda/random.py:214-217)


# ------------------------------------------------------------- geometry: oracle identity + known answers


def test_vector_helpers(dev, host):
    rng = np.random.RandomState(123)
    a, b = _vecs(rng, 100000), _vecs(rng, 100000)
    got = dev(18, a + b, 4)
    assert _same_bits(got, host(18, a + b, 4))
    A, B = np.stack(a, 1), np.stack(b, 1)
    npt.assert_almost_equal(got[0], np.einsum("ij,ij->i", A, B))  # test_simulations.py:23-41
    npt.assert_almost_equal(got[1:].T, np.cross(A, B))            # :44-64
    nv = dev(14, a, 3)
    assert _same_bits(nv, host(14, a, 3))
    npt.assert_almost_equal(nv.T, A / np.linalg.norm(A, axis=1)[:, None])  # :67-84
    R = [rng.random_sample(100000) - 0.5 for _ in range(9)]
    mm = dev(17, R + a, 3)
    assert _same_bits(mm, host(17, R + a, 3))
    npt.assert_almost_equal(mm.T, np.einsum("kij,kj->ki", np.stack(R, 1).reshape(-1, 3, 3), A))  # :142-159


def test_line_quadric_intersections(dev, host):
    """d = 1.1414213562373097 (test_simulations.py:181, :204, :227) + oracle identity on random rays."""
    s = 1 / np.sqrt(2)
    one = lambda *v: [np.array([x]) for x in v]  # noqa: E731
    npt.assert_almost_equal(dev(11, one(-0.1, -0.1, s, s, 1.0), 1)[0, 0], 1.1414213562373097)
    npt.assert_almost_equal(dev(12, one(-0.1, -0.1, 0.0, s, s, 0.0, 1.0), 1)[0, 0], 1.1414213562373097)
    npt.assert_almost_equal(dev(13, one(-0.1, -0.1, 0.0, s, s, 0.0, 1.0, 1.0, 1.0), 1)[0, 0], 1.1414213562373097)
    rng = np.random.RandomState(7)
    n = 200000
    r0, st = _vecs(rng, n, 3, 1.0), _vecs(rng, n, 3, 2.0)
    rad = [0.2 + rng.random_sample(n)]
    for op, cols in ((11, r0[:2] + st[:2] + rad), (12, r0 + st + rad), (13, r0 + st + [rad[0], rad[0] * 0.7, rad[0] * 1.3])):
        assert _same_bits(dev(op, cols, 1), host(op, cols, 1))


def test_ray_triangle_known_answers_and_oracle_identity(dev, host):
    """[1, -1, 10, nan, nan] (test_simulations.py:242-259); Moller-Trumbore with reciprocal-then-multiply,
    inclusive bounds, det != 0 (simulations.py:234-275)."""
    tri = np.array([2.0, 0, 0, 0, 2.0, 0, 0.0, 0, 0])
    r0s = np.array([[0.1, 0.1, 1.0]] * 4 + [[10.0, 10, 0]])
    steps = np.array([[0, 0, -1.0], [0, 0, 1], [0, 0, -0.1], [1.0, 1.0, 0], [0, 0, 1.0]])
    cols = [np.full(5, t) for t in tri] + [r0s[:, k] for k in range(3)] + [steps[:, k] for k in range(3)]
    npt.assert_almost_equal(dev(5, cols, 1)[0], [1, -1, 10, np.nan, np.nan])
    rng = np.random.RandomState(8)
    n = N
    tri = _vecs(rng, n, 9, 2.0)
    r0, st = _vecs(rng, n, 3, 2.0), _vecs(rng, n, 3, 2.0)
    # edge cases: rays through a vertex / along an edge (u or v exactly 0 or 1), parallel rays (det == 0),
    # degenerate triangles
    k = 4096
    for c in range(9):
        tri[c][:k] = np.round(tri[c][:k] * 4) / 4
    for c in range(3):
        r0[c][:k] = np.round(r0[c][:k] * 4) / 4
        st[c][:k] = np.round(st[c][:k] * 2) / 2
    tri[3][:64], tri[4][:64], tri[5][:64] = tri[0][:64], tri[1][:64], tri[2][:64]  # B == A
    got, want = dev(5, tri + r0 + st, 1), host(5, tri + r0 + st, 1)
    assert _same_bits(got, want)
    assert np.isfinite(got).mean() > 0.01  # some rays do hit


def test_reflection_and_crossing_known_answers_and_oracle_identity(dev, host):
    """test_simulations.py:278-295, :312-325, :346-359: the step turns, the normal is flipped IN PLACE,
    the walker is moved epsilon off the surface."""
    n0 = np.array([0.0, 1.0, 1.0]) / np.sqrt(2)
    one = lambda *v: [np.array([float(x)]) for x in v]  # noqa: E731
    for eps in (0.0, 0.5):
        o = dev(6, one(0, 0, 0, 0, 0, 1, 0.5, *n0, eps), 9)[:, 0]
        npt.assert_almost_equal(o[3:6], [0.0, -1.0, 0.0])
        npt.assert_almost_equal(o[6:9], -n0)
        npt.assert_almost_equal(o[0:3], np.array([0, 0, 0.5]) + eps * -n0)
    o = dev(7, one(0, 0, 0, 0, 0, 1, 1.0, 0, 0, -1.0, 1e-10), 9)[:, 0]  # crossing the plane z = 1 upwards
    npt.assert_almost_equal(o[0:3], [0.0, 0, 1 + 1e-10])
    rng = np.random.RandomState(9)
    n = 500000
    r0, st, nn = _vecs(rng, n, 3, 2.0), _vecs(rng, n), _vecs(rng, n)
    ln = np.sqrt(sum(c * c for c in st))
    st = [c / ln for c in st]
    ln = np.sqrt(sum(c * c for c in nn))
    nn = [c / ln for c in nn]
    d, eps = [rng.random_sample(n)], [np.full(n, 1e-13)]
    for op in (6, 7):
        cols = r0 + st + d + nn + eps
        assert _same_bits(dev(op, cols, 9), host(op, cols, 9))
    # grazing hit: step perpendicular to the normal (dp == 0) and a zero reflected step (0/0 -> NaN, like the reference)
    cols = one(0, 0, 0, 1, 0, 0, 0.5, 0, 1, 0, 1e-13)
    assert _same_bits(dev(6, cols, 9), host(6, cols, 9))


def test_subvoxel_lookups_reproduce_the_reference_scans(dev, host):
    """_ll/_ul_subvoxel_overlap(_periodic) (simulations.py:616-679) are linear scans with strict / non-strict
    comparisons; the device uses a multiply + correction against the real boundaries, and the cooperative
    kernel a loop-free variant that must either agree or report that it cannot decide."""
    for n_sv, L in ((50, 2.145e-4), (10, 1.0), (1, 3e-6), (7, 1.93e-4)):
        xs = np.linspace(0, L, n_sv + 1)
        rng = np.random.RandomState(n_sv)
        n = 400000
        x1 = (rng.random_sample(n) * 5 - 2) * L
        x2 = x1 + (rng.random_sample(n) - 0.5) * L * rng.choice([1e-3, 0.05, 0.5, 3.0], n)
        # boundary-equal coordinates, both orders, in several periodic images
        b = np.concatenate([xs + k * L for k in (-2, -1, 0, 1, 3)])
        x1 = np.concatenate([x1, b, b, np.nextafter(b, np.inf), np.nextafter(b, -np.inf), b[:-1]])
        x2 = np.concatenate([x2, b, np.nextafter(b, np.inf), b, b, b[1:]])
        # coordinates at 1e-7 ... 1e-3 subvoxel widths from a boundary, on both sides (the float32 shortcut's margin)
        h = L / n_sv
        near = np.concatenate([b + s * d * h for d in (1e-7, 1e-6, 3e-5, 9e-5, 1.1e-4, 2e-4, 1e-3) for s in (-1, 1)])
        x1 = np.concatenate([x1, near, near])
        x2 = np.concatenate([x2, near + 0.3 * h, near - 0.45 * h])
        for op in (8, 9):
            got, want = dev(op, [x1, x2], 2, xs), host(op, [x1, x2], 2, xs)
            npt.assert_array_equal(got, want)
        fast, want = dev(10, [x1, x2], 3, xs), host(9, [x1, x2], 2, xs)
        ok = fast[2] == 1.0
        assert ok.mean() > 0.99  # the loop-free path decides almost always ...
        npt.assert_array_equal(fast[:2, ok], want[:, ok])  # ... and is right whenever it says so
        f32 = dev(19, [x1, x2], 3, xs)  # the float32 shortcut tried first: decides unless a coordinate is
        ok = f32[2] == 1.0              # within 1e-4 subvoxel widths of a boundary, and is right whenever it does
        n_eq = 5 * len(b) - 1  # the boundary-equal inputs appended above never qualify
        assert ok[:n].mean() > 0.998 and not ok[n : n + n_eq].any()
        npt.assert_array_equal(f32[:2, ok], want[:, ok])
    xs = np.linspace(0, 1.0, 11)
    g = dev(8, [np.array([0.25, 0.35, -1.0, 0.95, xs[3], xs[3]]), np.array([0.35, 0.25, 0.05, 7.0, xs[3], xs[4]])], 2, xs)
    npt.assert_array_equal(g.T, [[2, 4], [2, 4], [0, 1], [9, 10], [3, 3], [3, 4]])
    g = dev(9, [np.array([-0.05, 0.95, 2.31, -1.25]), np.array([0.05, 1.05, 2.32, -1.15])], 2, xs)
    npt.assert_array_equal(g.T, [[-1, 1], [9, 11], [23, 24], [-13, -11]])


def test_kernel_parameters_read_from_the_kernarg_segment_are_the_parameters(dev):
    """fresh_params<1> (csrc/dsim_kernels.h) re-reads WalkParams from the kernarg segment instead of keeping ~100 dwords in
    SGPRs; that is only right while the struct is the kernel's first and only argument.  The unit
    library launches a kernel of that signature with every byte of the struct patterned and compares on the device."""
    import ctypes as C
    import os

    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "disimpy_amd", "csrc", "libdsim_unit.so"))
    assert lib.dsu_kernarg_check() == 0
