"""
River branch of the drop-in namespace (reference `custom_filters.py:128-199,
770-831`): host-side operators, pinned by outputs of the imported reference
(tests/golden/rivers.npz, tests/golden/make_golden_rivers.py).  ``RouteRivers``
needs no GPU; the composed chains run ``ExpandFilter`` / ``BinaryClosing`` on the
device and are in the ``gpu`` half.
"""
import numpy as np
import pytest

import hydrodem_amd as hd
from hydrodem_amd.filters.custom_filters import RouteRivers, ProcessRivers, ClipLagoonsRivers
from oracle import hdem_oracle_rivers as orv


@pytest.fixture(scope="module")
def gold(golden):
    return golden("rivers.npz")


@pytest.mark.parametrize("window,key", [(3, "routed3"), (5, "routed5")])
def test_route_rivers_equals_the_imported_reference(gold, window, key):
    got = RouteRivers(window_size=window, dem=gold["dem"]).apply(gold["mask"])
    assert got.dtype == np.float64 and np.array_equal(got, gold[key])
    assert np.array_equal(orv.route_rivers(gold["mask"], gold["dem"], window), gold[key])


def test_route_rivers_on_the_reference_suites_own_rasters(gold):
    got = RouteRivers(window_size=3, dem=gold["ref_hsheds"]).apply(gold["ref_expand"].astype(float))
    assert np.array_equal(got, gold["ref_routed"])
    assert np.array_equal(orv.route_rivers(gold["ref_expand"], gold["ref_hsheds"]), gold["ref_routed"])


def test_route_rivers_keeps_its_own_copy_of_the_dem_and_checks_the_window(gold):
    dem = gold["dem"].copy()
    f = RouteRivers(window_size=3, dem=dem)
    dem[:] = 0                                             # custom_filters.py:163: deepcopy
    before = f.dem.copy()
    assert np.array_equal(f.apply(gold["mask"]), gold["routed3"])
    assert np.array_equal(f.dem, before, equal_nan=True)   # the 10000 marks go to a working copy
    with pytest.raises(hd.WindowSizeEvenError):
        RouteRivers(window_size=4, dem=gold["dem"]).apply(gold["mask"])
    with pytest.raises(hd.NumpyArrayExpectedError):
        f.apply([[1.0]])
    with pytest.raises(TypeError):
        RouteRivers(3, gold["dem"])                        # keyword-only, like the reference


def test_river_chains_have_the_reference_members(gold):
    p = ProcessRivers(gold["ref_hsheds"])
    assert [type(f).__name__ for f in p.filters] == [
        "MaskPositives", "ExpandFilter", "RouteRivers", "BinaryClosing"]
    assert p.filters[1].window_size == 3 and p.filters[2].window_size == 3
    c = ClipLagoonsRivers(gold["seed_lagoons"], gold["seed_closing"])
    assert [type(f).__name__ for f in c.filters] == ["ProductFilter", "BitwiseXOR"]
    # the element-wise half needs no device
    got = c.apply(gold["seed_closing"])
    assert got.dtype == np.int64 and np.array_equal(got, gold["seed_clipped"])


def test_oracle_chain_equals_the_imported_reference(gold):
    pos, exp, routed, closing = orv.process_rivers(gold["seed_rivers"], np.nan_to_num(gold["dem"], nan=100.0))
    assert np.array_equal(pos, gold["seed_positives"]) and np.array_equal(exp, gold["seed_expand"])
    assert np.array_equal(routed, gold["seed_routed"]) and np.array_equal(closing, gold["seed_closing"])
    assert np.array_equal(orv.clip_lagoons_rivers(gold["seed_lagoons"], closing), gold["seed_clipped"])


@pytest.mark.gpu
def test_process_rivers_and_clip_equal_the_imported_reference(built, gold):
    rivers_routed = ProcessRivers(np.nan_to_num(gold["dem"], nan=100.0)).apply(gold["seed_rivers"])
    assert rivers_routed.dtype == bool and np.array_equal(rivers_routed, gold["seed_closing"])
    got = ClipLagoonsRivers(gold["seed_lagoons"], rivers_routed).apply(rivers_routed)
    assert got.dtype == np.int64 and np.array_equal(got, gold["seed_clipped"])
    # image_hsheds.py:203-205 on rasters of the reference's own suite
    rivers_routed = ProcessRivers(gold["ref_hsheds"]).apply(gold["ref_rivers"].astype(np.float32))
    assert np.array_equal(rivers_routed, gold["ref_closing"].astype(bool))
    got = ClipLagoonsRivers(gold["ref_lagoons_mask"].astype(np.int64), rivers_routed).apply(rivers_routed)
    assert np.array_equal(got, gold["ref_clipped"])
