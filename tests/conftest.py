import os
import sys

import numpy as np
import pytest

try:  # PyTorch bundles its own HIP runtime: when both are used in one process (the sharding
    import torch  # noqa: F401  tests), torch has to be imported BEFORE libdsim_hip.so is loaded
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


_SPAWNER = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Tests that need extra processes on the GPU box (two product ranks over gloo) fork them from a fork server, and the
    # server is started HERE - before any test has initialised the GPU: on this pool a process that has touched HIP must
    # not fork+exec.  The server itself never touches the GPU; its children start clean.
    global _SPAWNER
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        import multiprocessing as mp
        from multiprocessing import forkserver

        _SPAWNER = mp.get_context("forkserver")
        _SPAWNER.set_forkserver_preload(["numpy", "torch"])
        forkserver.ensure_running()


@pytest.fixture(scope="session")
def rank_spawner():
    if _SPAWNER is None:
        pytest.skip("needs the fork server that conftest starts for `-m gpu` sessions")
    return _SPAWNER


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle.oracle import Oracle

    return Oracle(exact=False)


@pytest.fixture(scope="session")
def oracle_exact():
    from oracle.oracle import Oracle

    return Oracle(exact=True)
