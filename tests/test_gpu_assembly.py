"""
Element-wise operators on device rasters and the device-resident final assembly
(SURVEY 8f-2), against outputs of the imported reference (tests/golden/assembly.npz,
made by tests/golden/make_golden_assembly.py from `hydro_dem_process.py:80-88,147-149`
evaluated with the reference's own filter classes).
"""
import numpy as np
import pytest

import hydrodem_amd as hd
from hydrodem_amd import backend, assembly
from hydrodem_amd.filters import ComposedFilter, ComposedFilterResults

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden, built):
    return golden("assembly.npz")


def test_final_assembly_is_bit_for_bit_the_references(gold):
    out, (first, second, third) = assembly.final_dem(
        gold["srtm"], gold["mask_lagoons"].astype(np.int64), gold["hsheds"],
        gold["lagoons_values"], gold["rivers"].astype(np.int64), keep_terms=True)
    assert out.dtype == np.float64
    assert np.array_equal(first, gold["first_term"])
    assert np.array_equal(second, gold["lagoons_values"])
    assert np.array_equal(third, gold["third_term"])
    assert np.array_equal(out, gold["final_dem"])


def test_the_same_calls_through_the_host_api_give_the_references_types(gold):
    """`hydro_dem_process.py:80-88` unchanged: host arrays in, NumPy result types out."""
    rivers, mask = gold["rivers"].astype(np.int64), gold["mask_lagoons"].astype(np.int64)
    both = hd.AdditionFilter(addend=mask).apply(rivers)
    neither = hd.SubtractionFilter(minuend=1).apply(both)
    first = hd.ProductFilter(factor=gold["srtm"]).apply(neither)
    third = hd.ProductFilter(factor=gold["hsheds"]).apply(rivers)
    assert both.dtype == np.int64 and first.dtype == np.float64 and third.dtype == np.float64
    final = hd.PostProcessingFinal().apply(first + gold["lagoons_values"] + third)
    assert final.dtype == np.float64 and np.array_equal(final, gold["final_dem"])


def test_mask_chains_on_device_equal_the_reference(gold):
    probe = backend.DeviceRaster.from_host(gold["probe"])
    for cls, key in ((hd.MaskPositives, "positives"), (hd.MaskNegatives, "negatives"),
                     (hd.MaskTallGroves, "tall")):
        chain = cls()
        got = chain.apply_device(probe)
        assert got.dtype == np.uint8 and np.array_equal(got.to_host(), gold[key]), key
        # the host call keeps the reference's type (int64 from ``bool * 1``): no silent
        # device detour for the element-wise members
        host = chain.apply(gold["probe"])
        assert host.dtype == np.int64 and np.array_equal(host, gold[key])


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8])
def test_elementwise_operators_on_every_raster_type(dtype):
    rng = np.random.default_rng(3)
    a = (rng.random((37, 53)) * 9).astype(dtype)            # odd size: the scalar tail
    b = (rng.random((37, 53)) * 5).astype(np.float32)
    da, db = backend.DeviceRaster.from_host(a), backend.DeviceRaster.from_host(b)
    wide = np.float64 if dtype == np.float64 else np.float32
    cases = [(hd.ProductFilter(factor=db), (b.astype(np.float64) * a).astype(wide)),
             (hd.AdditionFilter(addend=2.5), (2.5 + a.astype(np.float64)).astype(wide)),
             (hd.SubtractionFilter(minuend=db), (b.astype(np.float64) - a).astype(wide)),
             (hd.GreaterThan(value=4), (a > 4).astype(np.uint8)),
             (hd.LowerThan(value=4.0), (a < 4.0).astype(np.uint8)),
             (hd.BooleanToInteger(), a)]
    for f, want in cases:
        got = f.apply_device(da)
        assert got.dtype == want.dtype and np.array_equal(got.to_host(), want), type(f).__name__
    # a host array as operand is uploaded for the call
    got = hd.ProductFilter(factor=b).apply_device(da)
    assert np.array_equal(got.to_host(), (b.astype(np.float64) * a).astype(wide))
    with pytest.raises(ValueError, match="operand shape"):
        hd.ProductFilter(factor=np.ones((3, 3), np.float32)).apply_device(da)


def test_composed_filter_results_keeps_stages_on_the_device(gold):
    class Chain(ComposedFilterResults):  # pylint: disable=too-few-public-methods
        def __init__(self):
            super().__init__()
            self.filters = [hd.GreaterThan(value=0.0), hd.BooleanToInteger(),
                            hd.ProductFilter(factor=3.0)]

    chain = Chain()
    last = chain.apply_device(backend.DeviceRaster.from_host(gold["probe"]))
    assert set(chain.results.keys()) == {"GreaterThan", "BooleanToInteger", "ProductFilter"}
    assert chain.results.downloaded() == []                 # nothing downloaded yet
    assert len(chain.results) == 3 and chain.results.get("Around") is None
    assert np.array_equal(chain.results["GreaterThan"], gold["positives"])
    assert chain.results.downloaded() == ["GreaterThan"]
    assert np.array_equal(chain.results.get("BooleanToInteger"), gold["positives"])
    assert [k for k, _ in chain.results.items()] == list(chain.results)
    assert np.array_equal(last.to_host(), gold["positives"] * 3.0)
    with pytest.raises(KeyError):
        chain.results["Around"]                             # pylint: disable=pointless-statement
    chain.results.release(keep=last)                        # the caller's raster survives
    assert np.array_equal(last.to_host(), gold["positives"] * 3.0)
    last.free()
    # the host form is the reference's: every stage a host array, stored eagerly
    host = Chain()
    host.apply(gold["probe"])
    assert np.array_equal(host.results["BooleanToInteger"], gold["positives"])


def test_device_chain_of_stencils_and_algebra_against_the_oracle():
    """groves -> (x 1) -> box mean as ONE device chain, checked against the oracle's
    step-by-step result on the host (not against the GPU's own)."""
    import oracle
    from oracle import c_oracle
    img = oracle.synth_dem(120, 150, pits=False)
    groves = oracle.synth_groves(120, 150)
    chain = ComposedFilter()
    chain.filters = [hd.GrovesCorrection(groves), hd.ProductFilter(factor=1.0),
                     hd.PostProcessingFinal()]
    with backend.DeviceRaster.from_host(img) as raster:
        got = chain.apply_device(raster).to_host()
    smooth = c_oracle.groves_ref(img, groves, 1).astype(np.float32)
    want = c_oracle.boxmean3(smooth, True)
    assert (got != want).sum() <= 2                         # (groves: borderline threshold cells)
    # the same list through apply(): ProductFilter has no auto_device, so the members run
    # one by one with the reference's host semantics
    assert (chain.apply(img) != want).sum() <= 2


def test_groves_class_with_values_other_than_0_and_1_follows_the_reference_algebra():
    """`custom_filters.py:724-732`: ``m = class * tall``, ``out = highlight * (1 - m) +
    smooth`` -- for a class value k that is a blend, not a mask.  The fused kernel is for
    0 / 1 rasters; other values take the literal algebra (float64, like the reference)."""
    import oracle
    from oracle import c_oracle
    img = oracle.synth_dem(90, 110, pits=False)
    img[40:50, 30:60] += 4.0                                   # a tall grove
    klass = (oracle.synth_groves(90, 110).astype(np.int64) * 2)
    klass[42:48, 35:55] = 2
    got = hd.GrovesCorrection(klass).apply(img)
    smooth = c_oracle.quadratic_ref(img, 15)
    hl = img - smooth
    want = hl * (1 - klass * ((hl > 1.5) * 1)) + smooth
    assert got.dtype == np.float64
    sure = np.abs(hl - 1.5) > 1e-3
    assert (klass * (hl > 1.5)).sum() > 10 and np.abs(got - want)[sure].max() <= 1e-4
    # iterated: member by member too
    got3 = hd.GrovesCorrectionsIter(klass, iterations=2).apply(img)
    assert got3.dtype == np.float64 and np.isfinite(got3).all()
    with pytest.raises(NotImplementedError):
        hd.GrovesCorrection(klass).apply_device(backend.DeviceRaster.from_host(img))
