"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's HydroSHEDS /
lagoon branch, NumPy + SciPy.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import it; the product path never does.

Reference lines are in cguerrero/hydrodem/filters/custom_filters.py unless noted.
Pinned by tests/golden/lagoons.npz: rasters the reference's own test suite holds
(hsheds_nan_values_expected -> MajorityFilter(11) -> hsheds_majority_11_expected
-> TidyingLagoons -> lagoons_expected) and outputs of the imported reference
operators on seeded inputs (tests/golden/make_golden_lagoons.py).

Third-party arithmetic on the path: scipy.ndimage.binary_erosion /
binary_closing / grey_dilation (extension_filters.py:187-345), unpinned by the
reference; here SciPy 1.15.3 -- this file calls the same functions.
"""

import numpy as np
from scipy import ndimage

from .hdem_oracle_fourier import _box_sum, expand  # noqa: F401  (ExpandFilter :76-125)


def correct_nan_values(dem, window=3):
    """CorrectNANValues.apply (:287-317): interior cells < 0 become the mean of
    the cells of their window (centre excluded) that are >= 0; the windows read
    a float32 snapshot (sliding_window.py:128-132), the mean is NumPy's float32
    ``mean`` of at most 8 values in window order.  Returns a new array of the
    input's dtype (the reference writes into its input)."""
    out = np.array(dem, copy=True)
    g = np.asarray(dem).astype(np.float32)
    r = window // 2
    h, w_ = g.shape
    for j in range(r, h - r):
        for i in range(r, w_ - r):
            if int(g[j, i] < 0) == 1:
                win = g[j - r:j + r + 1, i - r:i + r + 1].copy()
                win[r, r] = np.nan
                nb = win[~np.isnan(win)]
                nb = nb[nb >= 0]
                with np.errstate(invalid="ignore"), np.testing.suppress_warnings() as sup:
                    sup.filter(RuntimeWarning)
                    out[j, i] = nb.mean()
    return out


def majority_filter(img, window=11, fraction=0.7):
    """MajorityFilter.apply (:44-73): the value that fills more than 70 % of
    (window^2 - 1) cells of the window minus its four corners (the centre counts),
    else 0; centres whose window fits only.  Returns float64 like the reference."""
    g = np.asarray(img).astype(np.float32)
    h, w_ = g.shape
    r = window // 2
    out = np.zeros((h, w_))
    need = (window ** 2 - 1) * fraction
    for v in np.unique(g[~np.isnan(g)]):
        m = (g == v).astype(np.float64)
        cnt = _box_sum(m, window)
        pad = np.pad(m, r)
        cnt -= (pad[0:h, 0:w_] + pad[0:h, 2 * r:2 * r + w_] +
                pad[2 * r:2 * r + h, 0:w_] + pad[2 * r:2 * r + h, 2 * r:2 * r + w_])
        hit = cnt > need
        hit[:r] = hit[h - r:] = False
        hit[:, :r] = hit[:, w_ - r:] = False
        out[hit] = v
    return out


def binary_erosion(mask, iterations=1):
    """BinaryErosion.apply (extension_filters.py:187-235)."""
    return ndimage.binary_erosion(mask, iterations=iterations)


def binary_closing(mask, structure=None):
    """BinaryClosing.apply (extension_filters.py:238-293)."""
    return ndimage.binary_closing(mask, structure=structure)


def grey_dilation(img, size):
    """GreyDilation.apply (extension_filters.py:296-345)."""
    return ndimage.grey_dilation(img, size=size)


def tidying_lagoons(img):
    """TidyingLagoons.apply (:564-610): erode the non-zero cells twice, expand by
    the 7 x 7 circular window, multiply with the input, 7 x 7 grey dilation."""
    content = binary_erosion(img, 2)
    content = expand(content, 7)
    content = img * content
    return grey_dilation(content, (7, 7))


def lagoons_detection(hsheds):
    """LagoonsDetection.apply (:613-661).  Returns (mask, dict of stages)."""
    fixed = correct_nan_values(hsheds)
    major = majority_filter(fixed, 11)
    tidy = tidying_lagoons(major)
    mask = (tidy > 0.0) * 1
    return mask, {"CorrectNANValues": fixed, "MajorityFilter": major, "TidyingLagoons": tidy}


from hdem_synth import synth_hsheds  # noqa: E402,F401  (re-exported for the tests)
