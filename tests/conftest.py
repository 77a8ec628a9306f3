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


# BASELINE configs 4 and 5 on one GPU (tests/test_gpu_zz_configs.py): the rasters and
# their C priority-flood oracles take minutes of host time, so they are made in background
# threads from the moment the collection is known (the C calls and NumPy's loops release
# the GIL) while the other GPU tests run.
BIG_CASES = {"config4": (32768, 32768, 4),        # 4 row blocks of 8192 x 32768
             "config5": (16384, 65536, 2)}        # 2 of config 5's 8192 x 65536 blocks


def _big_oracle(h, w):
    import oracle
    from oracle import c_oracle
    z = oracle.synth_dem(h, w)
    want = c_oracle.sinkfill_pflood(z)
    return z, want, c_oracle.d8(want)


def pytest_collection_finish(session):
    if not any("big" in getattr(item, "fixturenames", ()) for item in session.items):
        return
    if session.config.getoption("collectonly", False) or not _has_gpu():
        return
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    c_oracle.build()
    pool = ThreadPoolExecutor(len(BIG_CASES))
    session.config._hdem_big = (pool, {k: pool.submit(_big_oracle, h, w)
                                       for k, (h, w, _) in BIG_CASES.items()})


@pytest.fixture(scope="session")
def big(request, built):
    """{case: future of (raster, filled oracle, D8 oracle)}; see BIG_CASES."""
    pool, futures = request.config._hdem_big
    yield futures
    pool.shutdown(wait=True, cancel_futures=True)


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
