import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from hydrodem_amd import backend
        return backend.device_count() > 0
    except Exception:  # pylint: disable=broad-except
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently;
    # without -m, GPU tests are skipped where there is no device.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU here")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def built():
    """libhydrodem_hip.so + liboracle_c.so exist (compile if the tree is fresh)."""
    import __graft_entry__ as g
    from hydrodem_amd import backend
    if not os.path.exists(backend.LIB_PATH):
        g.build()
    from oracle import c_oracle
    c_oracle.build()
    return True
