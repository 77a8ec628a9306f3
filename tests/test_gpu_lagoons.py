"""
Parity of the HydroSHEDS / lagoon branch on the HIP path, through
the C ABI: bit-exact against the reference's own rasters, against outputs of the
imported reference on seeded inputs (tests/golden/lagoons.npz) and, at a larger
size, against the CPU oracle.
"""
import numpy as np
import pytest

import hydrodem_amd as hd
from hydrodem_amd import backend
from oracle import hdem_oracle_lagoons as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lz(golden, built):
    assert backend.device_count() >= 1
    return golden("lagoons.npz")


def test_reference_rasters_chain(lz):
    maj = hd.MajorityFilter(window_size=11).apply(lz["ref_nan_values"])
    assert maj.dtype == np.float64 and np.array_equal(maj, lz["ref_majority_11"])
    tidy = hd.TidyingLagoons().apply(lz["ref_majority_11"])
    assert np.array_equal(tidy, lz["ref_lagoons"])
    # the generic (step by step) route of the same chain
    t = hd.TidyingLagoons()
    t.filters[3] = hd.GreyDilation(size=(7, 7))
    t._stock = lambda: False
    assert np.array_equal(t.apply(lz["ref_majority_11"].astype(np.float64)), lz["ref_lagoons"])


def test_synthetic_chain_matches_the_reference(lz):
    hs = lz["hs"].copy()
    fixed = hd.CorrectNANValues().apply(hs)
    assert fixed is hs and np.array_equal(fixed, lz["hs_fixed"], equal_nan=True)
    assert np.array_equal(hd.MajorityFilter(window_size=11).apply(fixed), lz["hs_majority"])
    assert np.array_equal(hd.MajorityFilter(window_size=5).apply(fixed), lz["hs_majority5"])
    det = hd.LagoonsDetection()
    raw = lz["hs"].copy()
    mask = det.apply(raw)
    assert mask.dtype == np.int64 and np.array_equal(mask, lz["hs_mask"])
    assert det.hsheds_nan_fixed is raw and np.array_equal(raw, lz["hs_fixed"], equal_nan=True)
    assert np.array_equal(det.lagoons_values, lz["hs_tidy"])
    assert np.array_equal(det.results["MaskPositives"], det.mask_lagoons)


def test_morphology_wrappers_match_scipy_through_the_reference(lz):
    m = lz["morph_in"].astype(bool)
    assert np.array_equal(hd.BinaryErosion(iterations=1).apply(m), lz["erosion1"].astype(bool))
    assert np.array_equal(hd.BinaryErosion(iterations=2).apply(m), lz["erosion2"].astype(bool))
    assert np.array_equal(hd.BinaryClosing().apply(m), lz["closing_default"].astype(bool))
    assert np.array_equal(hd.BinaryClosing(structure=np.ones((3, 3))).apply(m),
                          lz["closing_ones3"].astype(bool))
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
    assert np.array_equal(hd.BinaryClosing(structure=cross).apply(m), lz["closing_cross"].astype(bool))
    g = hd.GreyDilation(size=(7, 7)).apply(lz["grey_in"])
    assert g.dtype == np.float32 and np.array_equal(g, lz["grey77"])
    assert np.array_equal(hd.GreyDilation(size=(3, 5)).apply(lz["grey_in"]), lz["grey35"])
    assert np.array_equal(hd.ExpandFilter(window_size=7).apply(m.astype(np.float64)), lz["expand7"])
    assert np.array_equal(hd.BitwiseXOR(operand=m.astype(np.int64)).apply((~m).astype(np.int64) * 3),
                          lz["xor"])
    assert np.array_equal(hd.MaskPositives().apply(lz["grey_in"] - 25), lz["positives"])
    assert np.array_equal(hd.MaskNegatives().apply(lz["grey_in"] - 25), lz["negatives"])


@pytest.mark.parametrize("shape", [(700, 900), (131, 157), (64, 33)])
def test_larger_raster_against_the_oracle(shape):
    hs = L.synth_hsheds(*shape, seed=4)
    want_mask, stages = L.lagoons_detection(hs)
    det = hd.LagoonsDetection()
    mask = det.apply(hs.copy())
    assert np.array_equal(mask, want_mask) and mask.sum() > shape[0] * shape[1] // 700
    assert np.array_equal(det.hsheds_nan_fixed, stages["CorrectNANValues"], equal_nan=True)
    assert np.array_equal(det.lagoons_values, stages["TidyingLagoons"])
    assert np.array_equal(hd.MajorityFilter(window_size=11).apply(stages["CorrectNANValues"]),
                          stages["MajorityFilter"])


@pytest.mark.parametrize("window", [3, 5, 7, 9, 11, 13, 15])
def test_majority_near_the_share_threshold(window):
    """Few distinct values whose share hovers around 70 %, voids in between: the
    separable vote must name the same winners as a full count (oracle), window by
    window, for every window size the kernel is built for."""
    rng = np.random.default_rng([20240611, window])
    h, w_ = 203, 331                                  # not multiples of the 64 x 32 tile
    img = np.where(rng.random((h, w_)) < 0.74, 5.0, rng.integers(1, 4, (h, w_))).astype(np.float32)
    img[rng.random((h, w_)) < 0.01] = np.nan
    img[60:140, 100:260] = np.where(rng.random((80, 160)) < 0.9, -3.0, 5.0)
    want = L.majority_filter(img, window=window)
    got = hd.MajorityFilter(window_size=window).apply(img)
    assert np.array_equal(got, want)
    if window > 3:
        assert 0 < np.count_nonzero(want) < want.size


@pytest.mark.parametrize("shape", [(7, 7), (8, 23), (11, 11), (12, 70), (65, 9)])
def test_smallest_rasters(shape):
    """Rasters barely larger than the windows (7 for the tidying, 11 for the whole
    detection): every cell is within reach of a border, the reflected halo of the
    dilation folds back on itself."""
    rng = np.random.default_rng([5, *shape])
    img = np.where(rng.random(shape) < 0.8, 3.0, rng.integers(0, 3, shape)).astype(np.float32)
    assert np.array_equal(hd.TidyingLagoons().apply(img.copy()), L.tidying_lagoons(img))
    for ws in (3, 5, 7):
        assert np.array_equal(hd.MajorityFilter(window_size=ws).apply(img),
                              L.majority_filter(img, window=ws))
    if min(shape) >= 11:
        hs = img.copy()
        hs[rng.random(shape) < 0.05] = -32768.0
        want_mask, stages = L.lagoons_detection(hs.copy())
        det = hd.LagoonsDetection()
        assert np.array_equal(det.apply(hs.copy()), want_mask)
        assert np.array_equal(det.lagoons_values, stages["TidyingLagoons"])


def test_error_behaviour():
    with pytest.raises(hd.WindowSizeHighError):
        hd.MajorityFilter(window_size=11).apply(np.zeros((8, 30), dtype=np.float32))
    with pytest.raises(hd.WindowSizeEvenError):
        hd.MajorityFilter(window_size=4).apply(np.zeros((30, 30), dtype=np.float32))
    with pytest.raises(ValueError, match="odd"):
        hd.GreyDilation(size=(4, 4)).apply(np.zeros((9, 9), dtype=np.float32))
    with pytest.raises(hd.NumpyArrayExpectedError):
        hd.BinaryErosion(iterations=1).apply([[1, 0]])
