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


def _stale(lib_path, source_dir, suffixes):
    """The library is missing or older than one of its sources."""
    if not os.path.exists(lib_path):
        return True
    built_at = os.path.getmtime(lib_path)
    return any(os.path.getmtime(os.path.join(source_dir, f)) > built_at
               for f in os.listdir(source_dir) if f.endswith(suffixes))


@pytest.fixture(scope="session")
def built():
    """libhydrodem_hip.so + liboracle_c.so exist and are not older than their sources
    (a stale library would test yesterday's kernels)."""
    import __graft_entry__ as g
    from hydrodem_amd import backend
    csrc = os.path.dirname(backend.LIB_PATH)
    if _stale(backend.LIB_PATH, csrc, (".hip", ".h", "Makefile")) or \
            _stale(backend.LIB_PATH, os.path.join(ROOT, "include"), (".h",)):
        g.build()
    from oracle import c_oracle
    c_oracle.build()
    return True
