"""The C-ABI library loads and exports every symbol include/dsim.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsim_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from disimpy_amd import _lib

    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from disimpy_amd import _lib

    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.dsim_abi_version() == 1


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    """On a box without a GPU every compute entry point must raise."""
    from disimpy_amd import _lib, gradients, simulations, substrates

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    g = np.zeros((1, 4, 3))
    with pytest.raises(_lib.DsimError):
        _lib.Sim(10, g, 1e-3, 1e-6)
    h = ctypes.c_void_p()
    gx = np.zeros((1, 4))
    rc = _lib.load().dsim_create(ctypes.byref(h), 0, 10, 0, 1, 4, _lib.p_f64(gx), _lib.p_f64(gx), _lib.p_f64(gx), 1e-3, 1e-6)
    assert rc == -2 and b"no HIP device" in _lib.load().dsim_last_error()
    gradient, dt = gradients.pgse(10e-3, 40e-3, 10, np.array([1e9]), np.array([[1.0, 0, 0]]))
    with pytest.raises(Exception, match="unable to detect an AMD GPU"):
        simulations.simulation(10, 2e-9, gradient, dt, substrates.free(), quiet=True)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "disimpy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the CPU oracle", "").replace("oracle's", "").replace(
                    "oracle/dsim_oracle.c", ""
                ).replace("CPU oracle", "").lower().replace("the oracle", ""), os.path.join(dirpath, f)


def test_host_rng_init_matches_oracle(oracle_libm):
    """dsim_rng_init (host, no GPU needed) == the oracle's restatement of
    create_xoroshiro128p_states, including a non-zero first index reached through the jump matrices."""
    from disimpy_amd import _lib

    st = oracle_libm.rng_states(3000, 123)
    got = _lib.rng_init(123, 0, 3000)
    np.testing.assert_array_equal(got[:, 0], st["s0"])
    np.testing.assert_array_equal(got[:, 1], st["s1"])
    for first in (1, 2, 255, 256, 1023, 2999):
        np.testing.assert_array_equal(_lib.rng_init(123, first, 1)[0], got[first])
    st7 = oracle_libm.rng_states(5, 7)
    got7 = _lib.rng_init(7, 3, 2)
    assert (int(got7[0, 0]), int(got7[0, 1])) == (int(st7["s0"][3]), int(st7["s1"][3]))
