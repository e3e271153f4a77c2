"""disimpy_amd/csrc/dsim_math.h (the product's bit-reproducible logf / cos) against
glibc, through the oracle's two builds (exact build = dsim_math.h, default = glibc)."""
import numpy as np


def _ulp_f32(x):
    return np.spacing(np.abs(x).astype(np.float32)).astype(np.float64)


def test_logf_is_correctly_rounded_on_samples(oracle_exact, oracle_libm):
    rng = np.random.RandomState(0)
    bits = rng.randint(0x00800000, 0x3F800001, size=200000).astype(np.uint32)
    xs = np.concatenate([bits.view(np.float32), np.float32([1.0, 0.5, 2.0**-53, 0.99999994, 1.2e-38])])
    mine = np.array([oracle_exact.lib.dso_u_logf(float(x)) for x in xs], dtype=np.float32)
    cr = np.log(xs.astype(np.float64)).astype(np.float32)  # float64 log rounded once: correctly rounded except ~1e-8 of inputs
    assert np.array_equal(mine, cr)
    gl = np.array([oracle_libm.lib.dso_u_logf(float(x)) for x in xs[:20000]], dtype=np.float32)
    assert np.abs(mine[:20000].astype(np.float64) - gl) .max() <= _ulp_f32(gl).max()  # glibc logf is within 1 ulp
    assert oracle_exact.lib.dso_u_logf(0.0) == -np.inf
    assert oracle_exact.lib.dso_u_logf(1.0) == 0.0


def test_cos_0_2pi_absolute_error(oracle_exact):
    """Table + Taylor cos: absolute error < 3e-16 on every float32 uniform sampled (<= 2 ulp
    where |cos| >= 1/2, the class of CUDA's double cos), exact at the table nodes."""
    rng = np.random.RandomState(1)
    u = rng.randint(0, 2**24, size=200000).astype(np.float64) / 2**24
    u = np.concatenate([u, [0.0, 0.25, 0.5, 0.75, 1.0 - 2.0**-24, 1.0, 2.0**-30], np.arange(257) / 256.0])
    x = 6.283185307179586 * u.astype(np.float32).astype(np.float64)
    mine = np.array([oracle_exact.lib.dso_u_cos(float(v)) for v in x])
    ref = np.cos(x.astype(np.longdouble))
    err = np.abs(mine - ref).astype(np.float64)
    assert err.max() < 3e-16
    big = np.abs(ref) >= 0.5
    assert (err[big] / np.spacing(np.abs(ref[big]).astype(np.float64))).max() <= 2.0
    assert oracle_exact.lib.dso_u_cos(0.0) == 1.0


def test_div3_equals_ieee_division(oracle_exact):
    """ds_div3 (one division + residual corrections) returns the IEEE quotients of the
    reference's three divisions (simulations.py:72-73) on 3e6 random vectors, including
    lengths with long runs of ones in the significand."""
    import ctypes as C

    lib = oracle_exact.lib
    lib.dso_u_div3.argtypes = [C.POINTER(C.c_double), C.c_double]
    rng = np.random.RandomState(4)
    n = 1000000
    v = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-8, 3, size=(n, 1))
    length = np.sqrt((v * v).sum(axis=1))
    length[::7] = np.nextafter(2.0 ** rng.randint(-20, 20, size=len(length[::7])), 0)  # 0x1.fff...p
    want = v / length[:, None]
    got = v.copy()
    for i in range(n):
        lib.dso_u_div3(got[i].ctypes.data_as(C.POINTER(C.c_double)), length[i])
    assert np.array_equal(got, want)
    for bad in (0.0, np.inf, np.nan):
        g = np.array([1.0, 0.0, -2.0])
        lib.dso_u_div3(g.ctypes.data_as(C.POINTER(C.c_double)), bad)
        with np.errstate(all="ignore"):
            np.testing.assert_array_equal(g, np.array([1.0, 0.0, -2.0]) / bad)


def _f32_bits(x):
    return int(np.float32(x).view(np.uint32))


def test_logf_is_correctly_rounded_on_every_possible_uniform(oracle_exact):
    """EXHAUSTIVE: a float32 uniform of the random step (numba/cuda/random.py:199-221 - the 53-bit fraction rounded to
    float32) is 0 or one of the 444 596 225 floats of [2^-53, 1]; on every one of them ds_logf returns the correctly rounded
    logarithm (the x87 long double logl rounded once to float32).  The bit-identity tests compare the kernels with an
    oracle that shares dsim_math.h - this test is what says the shared function is the right one."""
    import ctypes as C

    first_bad = C.c_uint32(0)
    lo, hi = _f32_bits(2.0**-53), _f32_bits(1.0) + 1
    assert hi - lo == 444596225
    bad = oracle_exact.lib.dso_u_logf_sweep(lo, hi, C.byref(first_bad))
    assert bad == 0, "first wrong input: bits %#x" % first_bad.value


def test_cos_error_on_every_possible_uniform(oracle_exact):
    """EXHAUSTIVE: cos(6.283185307179586 * u2) for every float32 u2 of [2^-53, 1]: within 1.2e-16 of the long double
    cosine everywhere, within 1.01 ulp of it where |cos| >= 1/2 (measured: 1.116e-16 and 1.0054 ulp)."""
    import ctypes as C

    a, b = np.zeros(1), np.zeros(1)
    p = C.POINTER(C.c_double)
    oracle_exact.lib.dso_u_cos_sweep(_f32_bits(2.0**-53), _f32_bits(1.0) + 1, a.ctypes.data_as(p), b.ctypes.data_as(p))
    assert 0 < a[0] < 1.2e-16
    assert b[0] < 1.01


def test_perm_threshold_is_the_reference_comparison_on_integer_bits(oracle_exact):
    """ds_perm_threshold (dsim_math.h): `perm_prob < u` of disimpy/simulations.py:988 with u = k * 2^-53, k the upper 53 bits
    of the draw (numba/cuda/random.py:128-139), decided as k > threshold - for probabilities all over [0, 1], at both ends,
    outside the range, and for the k around every threshold"""
    import ctypes as C

    f = oracle_exact.lib.dso_u_perm_threshold
    f.argtypes, f.restype = [C.c_double], C.c_int64
    rng = np.random.RandomState(8)
    probs = np.concatenate([rng.uniform(0, 1, 2000), 10.0 ** rng.uniform(-18, 0, 2000), rng.randint(0, 2**53, 2000) * 2.0**-53,
                            [0.0, 1.0, 0.5, 2.0**-53, 2.0**-54, 1 - 2.0**-53, 1 + 2.0**-52, 7.0, -0.0, -1e-300, -3.0, np.inf, -np.inf, np.nan, 5e-324]])
    top = 2**53 - 1
    for p in probs:
        t = int(f(float(p)))
        ks = {0, 1, 2, top, top - 1, 2**52, int(rng.randint(0, 2**53))}
        if 0 <= t <= top:
            ks |= {max(0, t - 1), t, min(top, t + 1)}
        for k in ks:
            u = float(k) * 2.0**-53  # exact: k < 2^53
            assert (p < u) == (k > t), (p, k, t)
