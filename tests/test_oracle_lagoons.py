"""
Oracle of the HydroSHEDS / lagoon branch against tests/golden/
lagoons.npz: three rasters of the reference's own test suite that chain exactly,
and outputs of the imported reference on seeded inputs.  CPU only.
"""
import numpy as np
import pytest

from oracle import hdem_oracle_lagoons as L


@pytest.fixture(scope="module")
def lz(golden):
    return golden("lagoons.npz")


def test_reference_rasters_chain(lz):
    assert lz["ref_nan_values"].shape == (519, 508)
    assert np.array_equal(L.majority_filter(lz["ref_nan_values"], 11), lz["ref_majority_11"])
    assert np.array_equal(L.tidying_lagoons(lz["ref_majority_11"]), lz["ref_lagoons"])


def test_synthetic_chain_matches_the_reference(lz):
    fixed = L.correct_nan_values(lz["hs"])
    assert fixed.dtype == np.float32 and np.isnan(fixed).sum() == 2
    assert np.array_equal(fixed, lz["hs_fixed"], equal_nan=True)
    mask, stages = L.lagoons_detection(lz["hs"])
    assert np.array_equal(stages["MajorityFilter"], lz["hs_majority"])
    assert np.array_equal(stages["TidyingLagoons"], lz["hs_tidy"])
    assert np.array_equal(mask, lz["hs_mask"]) and mask.sum() > 500
    assert np.array_equal(L.majority_filter(lz["hs_fixed"], 5), lz["hs_majority5"])
