"""
float64 / complex128 forms of three mirrored operators against outputs of the imported
reference (tests/golden/f64.npz, made by tests/golden/make_golden_f64.py): the reference
works in double for such input (extension_filters.py:183,345,379,414) and so does the GPU
path since round 3 -- the inputs hold values float32 cannot represent, so a float32 detour
would show.
"""
import numpy as np
import pytest

import hydrodem_amd as hd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden):
    return golden("f64.npz")


def test_fourier_transform_in_double(gold, built):
    x = gold["x"]
    got = hd.FourierTransform().apply(x)
    assert got.dtype == np.complex128
    scale = np.abs(gold["fft"]).max()
    assert np.abs(got - gold["fft"]).max() <= 1e-13 * scale         # (float32: ~1e-7)
    back = hd.FourierITransform().apply(got)
    assert back.dtype == np.complex128
    assert np.abs(back - gold["ifft"]).max() <= 1e-12 * np.abs(x).max()
    assert np.abs(back.real - x).max() <= 1e-10                      # the digits float32 drops
    c = gold["c"]
    assert np.abs(hd.FourierTransform().apply(c) - gold["c_fft"]).max() <= 1e-13 * np.abs(gold["c_fft"]).max()
    assert np.abs(hd.FourierITransform().apply(c) - gold["c_ifft"]).max() <= 1e-13
    # float32 input keeps the reference's complex64
    assert hd.FourierTransform().apply(x.astype(np.float32)).dtype == np.complex64


def test_grey_dilation_keeps_float64_values(gold, built):
    x = gold["x"]
    for size, key in (((7, 7), "dil77"), ((3, 5), "dil35")):
        got = hd.GreyDilation(size=size).apply(x)
        assert got.dtype == np.float64 and np.array_equal(got, gold[key])
    got = hd.GreyDilation(size=(7, 7)).apply(gold["xi"])
    assert got.dtype == np.int64 and np.array_equal(got, gold["dil_int"])


def test_convolve_general_weights_in_double(gold, built):
    x, w = gold["x"], gold["w"]
    got = hd.Convolve(weights=w).apply(x)
    assert got.dtype == np.float64
    # same products, same order of summation (the flipped weights in C order, zeros
    # skipped): the last bit may differ where SciPy's compiler contracts a multiply-add
    assert np.abs(got - gold["conv"]).max() <= 4 * np.finfo(np.float64).eps * np.abs(gold["conv"]).max()
    got32 = hd.Convolve(weights=w).apply(x.astype(np.float32))
    assert got32.dtype == np.float32
    assert np.abs(got32 - gold["conv_f32"]).max() <= 2 * np.finfo(np.float32).eps * np.abs(gold["conv_f32"]).max()
