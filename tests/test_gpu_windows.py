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


@pytest.mark.parametrize("shape,ws", [((300, 1501), 55), ((97, 2050), 21), ((64, 470), 55),
                                      ((33, 513), 7)])
def test_blanks_fourier_across_block_seams(built, shape, ws):
    """Several 458-column blocks side by side and several 32-row segments on top of each
    other, widths that are no multiple of four: every cell against the NumPy oracle (cells
    within rounding of the threshold may fall either way; there are none in these cases
    unless the assert says so)."""
    from oracle import hdem_oracle_fourier as F
    rng = np.random.default_rng(shape[1] + ws)
    q = np.exp(rng.normal(0.0, 1.5, shape)).astype(np.float32)
    q[rng.random(shape) < 0.002] *= 200.0
    want, want_q, margin = F.blanks_fourier(q, ws)
    found, modified = hd.BlanksFourier(window_size=ws).apply(q.copy())
    diff = found != want
    borderline = np.abs(margin) <= 1e-6 * np.abs(q)
    assert not (diff & ~borderline).any(), f"{int((diff & ~borderline).sum())} cells differ"
    assert found.sum() > 20
    same = ~diff
    assert np.array_equal(modified[same], want_q.astype(np.float32)[same])
