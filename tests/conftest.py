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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
