"""
Regenerates tests/golden/f64.npz.  RUNS ONLY IN THE BUILD CONTAINER (see make_golden.py):
what the imported reference returns for float64 / complex128 input whose values float32
cannot hold -- FourierTransform / FourierITransform (scipy.fftpack works in double for
them, extension_filters.py:379,414), GreyDilation (scipy.ndimage.grey_dilation keeps the
input's type, :345) and Convolve with general weights (scipy.ndimage.convolve accumulates
and returns double, :183).

    python tests/golden/make_golden_f64.py
"""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))

from filters.extension_filters import (FourierTransform, FourierITransform, GreyDilation,  # noqa: E402
                                       Convolve)


def main():
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(64)
    out = {}
    # elevations with 12 significant digits: float32 keeps 7
    x = 1234.5 + rng.normal(0.0, 3.0, (45, 64)) + 1e-9 * rng.integers(0, 1000, (45, 64))
    assert not np.array_equal(x, x.astype(np.float32).astype(np.float64))
    out["x"] = x
    spec = FourierTransform().apply(x)
    out["fft"] = spec
    out["ifft"] = FourierITransform().apply(spec)
    c = (rng.normal(size=(33, 40)) + 1j * rng.normal(size=(33, 40))).astype(np.complex128)
    out["c"] = c
    out["c_fft"] = FourierTransform().apply(c)
    out["c_ifft"] = FourierITransform().apply(c)
    print("fft dtypes", spec.dtype, out["ifft"].dtype, out["c_fft"].dtype)
    for size in ((7, 7), (3, 5)):
        out[f"dil{size[0]}{size[1]}"] = GreyDilation(size=size).apply(x)
    xi = rng.integers(-5, 40, (30, 37)).astype(np.int64)
    out["xi"] = xi
    out["dil_int"] = GreyDilation(size=(7, 7)).apply(xi)
    w = rng.normal(size=(5, 3))
    w[1, 1] = 0.0
    out["w"] = w
    out["conv"] = Convolve(weights=w).apply(x)
    out["conv_f32"] = Convolve(weights=w).apply(x.astype(np.float32))
    print("conv dtypes", out["conv"].dtype, out["conv_f32"].dtype, "dil", out["dil77"].dtype,
          out["dil_int"].dtype)
    np.savez_compressed(os.path.join(HERE, "f64.npz"), **out)


if __name__ == "__main__":
    main()
