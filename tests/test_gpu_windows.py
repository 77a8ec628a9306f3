"""
Constructor arguments the reference's own pipeline never uses: CorrectNANValues with a
window other than 3 and BlanksFourier with a window other than 55, against outputs of the
imported reference (tests/golden/windows.npz, make_golden_windows.py).
"""
import numpy as np
import pytest

import hydrodem_amd as hd
from hydrodem_amd import backend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wz(golden, built):
    return golden("windows.npz")


@pytest.mark.parametrize("ws", [3, 5, 7])
def test_correct_nan_values_any_window(wz, ws):
    hs = wz["hs"].copy()
    got = hd.CorrectNANValues(window_size=ws).apply(hs)
    assert got is hs                                      # in place, like the reference
    assert np.array_equal(got, wz[f"fixed{ws}"], equal_nan=True)
    # the device form: the same cells, out of place
    dev = hd.CorrectNANValues(window_size=ws).apply_device(
        backend.DeviceRaster.from_host(wz["hs"])).to_host()
    assert np.array_equal(dev, wz[f"fixed{ws}"], equal_nan=True)


@pytest.mark.parametrize("ws", [15, 21, 35])
def test_blanks_fourier_any_window(wz, ws):
    found, modified = hd.BlanksFourier(window_size=ws).apply(wz["q"].copy())
    assert np.array_equal(found, wz[f"found{ws}"])
    assert found.sum() > 100
    # the reference multiplies in float64; here the float32 cell is kept or zeroed
    assert np.array_equal(modified, wz[f"modified{ws}"].astype(np.float32))


def test_window_limits(built):
    q = np.ones((300, 300), dtype=np.float32)
    with pytest.raises(hd.WindowSizeEvenError):
        hd.BlanksFourier(window_size=20).apply(q)
    with pytest.raises(hd.WindowSizeHighError):
        hd.BlanksFourier(window_size=301).apply(q)
    with pytest.raises(hd.WindowSizeHighError):
        hd.CorrectNANValues(window_size=5).apply(np.zeros((4, 9), dtype=np.float32))
    with pytest.raises(hd.WindowSizeEvenError):
        hd.CorrectNANValues(window_size=4).apply(np.zeros((9, 9), dtype=np.float32))
    with pytest.raises(ValueError):
        hd.CorrectNANValues(window_size=13).apply(np.zeros((40, 40), dtype=np.float32))
    with pytest.raises(ValueError):
        hd.BlanksFourier(window_size=5).apply(q)
