"""
CPU suite, part 2: the host side of the boundary -- operator protocol, error
classes and messages (the reference's tests assert the texts,
`cguerrero/tests/test_sliding_window.py:92-93,110-111,130-133`), and that the
C-ABI library loads and exports every symbol `include/hydrodem_hip.h`
declares.  No compute calls: there is no GPU here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import hydrodem_amd as hd
from hydrodem_amd import backend
from hydrodem_amd.filters import Filter, ComposedFilter, ComposedFilterResults

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    yield


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hydrodem_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(hdem_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(backend.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding declares a prototype for each of them
    bound = set(backend.SIGNATURES) | set(backend.OTHER_SYMBOLS)
    assert declared == bound, declared ^ bound


def test_library_loads_without_gpu_and_fails_loudly():
    lib = backend.load_library()
    assert lib.hdem_version() >= 100
    if backend.device_count() == 0:
        with pytest.raises(hd.BackendError):
            hd.SinkFill().apply(np.zeros((4, 4), np.float32))
        with pytest.raises(hd.BackendError):
            hd.PostProcessingFinal().apply(np.zeros((4, 4), np.float32))


def test_missing_library_is_an_error_not_a_fallback(tmp_path):
    with pytest.raises(hd.BackendError) as e:
        backend.load_library(str(tmp_path / "nope.so"))
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hydrodem_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text, f


def test_filter_base_type_check_and_messages():
    class Identity(Filter):
        def apply(self, image_to_filter):
            super().apply(image_to_filter)
            return image_to_filter

    a = np.arange(6.0).reshape(2, 3)
    assert Identity().apply(a) is a
    with pytest.raises(hd.NumpyArrayExpectedError) as e:
        Identity().apply([1, 2])
    assert str(e.value) == "Expected numpy ndarray type. Provided: <class 'list'>"
    assert isinstance(e.value, hd.HydroDEMException)
    with pytest.raises(TypeError):
        Filter()                                     # abstract
    assert str(hd.WindowSizeEvenError(4)) == "Window size: 4 cannot be an even number"
    assert str(hd.WindowSizeHighError(7, (5, 5))) == \
        "Window size: 7 cannot be higher than grid dimensions: (5, 5)"
    assert str(hd.CenterCloseBorderError((0, 0), 3)) == \
        "Center of window: (0, 0) too close of border. Window size: 3"


def test_simple_filters_semantics():
    a = np.array([[1.0, 2.0], [3.0, 4.0]])
    assert np.array_equal(hd.LowerThan(value=2.5).apply(a), a < 2.5)
    assert np.array_equal(hd.GreaterThan(value=2.5).apply(a), a > 2.5)
    b = hd.BooleanToInteger().apply(a > 2)
    assert b.dtype.kind == "i" and np.array_equal(b, [[0, 0], [1, 1]])
    assert np.array_equal(hd.ProductFilter(3).apply(a), 3 * a)
    assert np.array_equal(hd.ProductFilter().apply(a), a)
    assert np.array_equal(hd.AdditionFilter(a).apply(a), 2 * a)
    assert np.array_equal(hd.SubtractionFilter(minuend=10).apply(a), 10 - a)
    assert hd.SubtractionFilter().apply(3) == -3.0          # no type check, like the reference
    for cls, kw in ((hd.LowerThan, {"value": 1}), (hd.ProductFilter, {}),
                    (hd.AdditionFilter, {}), (hd.BooleanToInteger, {})):
        with pytest.raises(hd.NumpyArrayExpectedError):
            cls(**kw).apply(5)
    with pytest.raises(TypeError):
        hd.LowerThan(3)                                    # keyword-only, like the reference


def test_composed_filters_fold_left_and_keep_results():
    c = ComposedFilter()
    c.filters = [hd.AdditionFilter(1), hd.ProductFilter(2)]
    a = np.ones((2, 2))
    assert np.array_equal(c.apply(a), (a + 1) * 2)
    r = ComposedFilterResults()
    r.filters = [hd.AdditionFilter(1), hd.ProductFilter(2)]
    out = r.apply(a)
    assert np.array_equal(out, (a + 1) * 2)
    assert set(r.results) == {"AdditionFilter", "ProductFilter"}
    assert np.array_equal(r.results["AdditionFilter"], a + 1)
    with pytest.raises(hd.NumpyArrayExpectedError):
        c.apply("x")
    m = hd.MaskTallGroves()
    assert np.array_equal(m.apply(np.array([[1.4, 1.5, 1.6]])), [[0, 0, 1]])


def test_gpu_filter_shapes_match_the_reference_api():
    g = hd.GrovesCorrection(np.zeros((3, 3)))
    names = [type(f).__name__ for f in g.filters]
    assert names == ["QuadraticFilter", "SubtractionFilter", "MaskTallGroves",
                     "ProductFilter", "SubtractionFilter"]
    assert g.filters[0].window_size == 15 and g.filters[4].minuend == 1
    assert g.partial_results == []
    it = hd.GrovesCorrectionsIter(np.zeros((3, 3)))
    assert len(it.filters) == 3 and all(isinstance(f, hd.GrovesCorrection) for f in it.filters)
    assert len(hd.GrovesCorrectionsIter(np.zeros((3, 3)), iterations=5).filters) == 5
    p = hd.PostProcessingFinal()
    assert [type(f).__name__ for f in p.filters] == ["Convolve", "Around"]
    assert np.array_equal(p.filters[0].weights, np.ones((3, 3)))
    with pytest.raises(TypeError):
        hd.QuadraticFilter(15)                             # keyword-only
    for f in (hd.SinkFill(), hd.D8FlowDirection(), hd.QuadraticFilter(window_size=3), p, g, it):
        with pytest.raises(hd.NumpyArrayExpectedError):
            f.apply([[1.0]])
    assert hd.SinkFill(epsilon=0.01).epsilon == 0.01


def test_dropin_directory_resolves_flat_imports():
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from filters import Filter, ComposedFilter, ComposedFilterResults\n"
        "from filters.custom_filters import GrovesCorrectionsIter, PostProcessingFinal, QuadraticFilter\n"
        "from filters.simple_filters import ProductFilter, SubtractionFilter, AdditionFilter\n"
        "from filters.extension_filters import Convolve, Around\n"
        "from exceptions import WindowSizeEvenError, NumpyArrayExpectedError\n"
        "from sliding_window import SlidingWindow\n"
        "import hydrodem_amd\n"
        "assert GrovesCorrectionsIter is hydrodem_amd.GrovesCorrectionsIter\n"
        "assert Filter is hydrodem_amd.Filter\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "hydrodem_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_python_constants_match_the_header():
    header = open(os.path.join(ROOT, "include", "hydrodem_hip.h")).read()
    defines = {m.group(1): int(m.group(2), 0)
               for m in re.finditer(r"#define\s+(HDEM_FILL_[A-Z_]+)\s+(0x[0-9a-fA-F]+|\d+)", header)}
    assert defines["HDEM_FILL_WARM"] == backend.FILL_WARM
    assert defines["HDEM_FILL_ACT_TOP"] == backend.FILL_ACT_TOP
    assert defines["HDEM_FILL_ACT_BOTTOM"] == backend.FILL_ACT_BOTTOM
    assert defines["HDEM_FILL_GHOST_TOP"] == backend.FILL_GHOST_TOP
    assert defines["HDEM_FILL_GHOST_BOTTOM"] == backend.FILL_GHOST_BOTTOM
    assert defines["HDEM_FILL_SYNC_ONLY"] == backend.FILL_SYNC_ONLY
    assert defines["HDEM_FILL_NO_VERIFY"] == backend.FILL_NO_VERIFY
    assert defines["HDEM_FILL_RESUME"] == backend.FILL_RESUME
    assert defines["HDEM_FILL_GHOST_GIVEN"] == backend.FILL_GHOST_GIVEN
    assert defines["HDEM_FILL_NO_COARSE"] == backend.FILL_NO_COARSE
    enums = dict(re.findall(r"\b(HDEM_(?:K|ERR)_[A-Z0-9_]+|HDEM_OK)\s*=\s*(\d+)", header))
    assert int(enums["HDEM_K_FILL_TILE"]) == backend.K_FILL_TILE
    assert int(enums["HDEM_K_FILL_ROUND"]) == backend.K_FILL_ROUND
    assert int(enums["HDEM_K_BLOCKMAX"]) == backend.K_BLOCKMAX
    assert int(enums["HDEM_K_GROVES"]) == backend.K_GROVES
    assert int(enums["HDEM_K_COPY"]) == backend.K_COPY
    assert int(enums["HDEM_K_FILL_COARSE"]) == backend.K_FILL_COARSE
    assert int(enums["HDEM_K_FILL_FLAT"]) == backend.K_FILL_FLAT
    assert int(enums["HDEM_K_ELEMENTWISE"]) == backend.K_ELEMENTWISE
    ops = dict(re.findall(r"\b(HDEM_EW_[A-Z]+)\s*=\s*(\d+)", header))
    assert [int(ops["HDEM_EW_" + n]) for n in ("MUL", "ADD", "RSUB", "GT", "LT", "NONZERO")] == \
        [backend.EW_MUL, backend.EW_ADD, backend.EW_RSUB, backend.EW_GT, backend.EW_LT,
         backend.EW_NONZERO]
    types = dict(re.findall(r"\b(HDEM_T_[A-Z0-9]+)\s*=\s*(\d+)", header))
    assert {np.dtype(np.float32): int(types["HDEM_T_F32"]), np.dtype(np.float64): int(types["HDEM_T_F64"]),
            np.dtype(np.uint8): int(types["HDEM_T_U8"]),
            np.dtype(np.int64): int(types["HDEM_T_I64"])} == backend._EW_TYPES
    assert int(enums["HDEM_ERR_WINDOW_EVEN"]) == backend.WINDOW_EVEN
    assert int(enums["HDEM_ERR_WINDOW_HIGH"]) == backend.WINDOW_HIGH
    assert int(enums["HDEM_ERR_NOT_CONVERGED"]) == backend.NOT_CONVERGED
    # struct layouts the binding mirrors
    # struct layouts the binding mirrors: 6 int32 + 6 int64 + 2 int32 / 2 int64 + 1 double
    assert ctypes.sizeof(backend.FillStats) == 104 and backend.FillStats.pending.offset == 72
    assert backend.FillStats.partial_residency.offset == 80
    assert ctypes.sizeof(backend.KernelStat) == 24


def test_band_ranges_cover_the_raster_with_halos():
    from hydrodem_amd import streaming as S
    assert S.band_ranges(10, 4, 1) == [(0, 4, 0, 5), (4, 8, 3, 9), (8, 10, 7, 10)]
    assert S.band_ranges(5, 8, 21) == [(0, 5, 0, 5)]
    with pytest.raises(ValueError):
        S.band_ranges(5, 0, 1)


def test_bench_reports_pmc_traffic_only_for_the_kernel_it_was_taken_from(tmp_path, monkeypatch):
    """`roofline.traffic` is a committed rocprofv3 --pmc record stamped with the hash of the
    kernel's source; a record of another build, or of another size, is not reported."""
    import json
    import bench
    rec = tmp_path / "traffic.json"
    monkeypatch.setattr(bench, "TRAFFIC_RECORD", str(rec))
    assert bench.profiled_traffic(16384)[0] is None                    # no record
    good = {"kernel_hash": bench.kernel_source_hash(), "size": 16384, "bytes_per_launch": 1.5e10,
            "head": "abc1234", "source": "x"}
    rec.write_text(json.dumps(good))
    value, meta = bench.profiled_traffic(16384)
    assert value == 1.5e10 and meta["traffic_profile_head"] == "abc1234"
    assert bench.profiled_traffic(4096)[0] is None
    rec.write_text(json.dumps(dict(good, kernel_hash="0" * 16)))
    value, meta = bench.profiled_traffic(16384)
    assert value is None and "another build" in meta["traffic_note"]
    # the committed record belongs to the committed kernel
    monkeypatch.undo()
    if os.path.exists(bench.TRAFFIC_RECORD):
        committed = json.load(open(bench.TRAFFIC_RECORD))
        assert committed["kernel_hash"] == bench.kernel_source_hash(), \
            "hdem_sinkfill.hip changed after the PMC passes: re-run tools/profile_round.sh"
