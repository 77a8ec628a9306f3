"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's river branch
(`cguerrero/hydrodem/filters/custom_filters.py:128-199,770-831`), plain Python
loops + SciPy.  Only tests/, smoke() and bench.py's cpu_baseline leg may import
it; the product path never does.

Pinned by tests/golden/rivers.npz: outputs of the imported reference on seeded
inputs and on two rasters of the reference's own test suite used as inputs
(tests/golden/make_golden_rivers.py).  The reference's own pair for this
operator does not close here: `tests/test_filter.py:84-93` reads
``hsheds_rivers_routing_input.tif`` / ``mask_rivers_routing_input.tif`` from
``tests_inputs.zip``, a missing blob.
"""

import numpy as np
from scipy import ndimage

from .hdem_oracle_fourier import expand


def route_rivers(mask_rivers, dem, window=3):
    """RouteRivers.apply (:165-199), cell by cell as the reference does it:
    centres where the float32 mask truncates to 1 (sliding_window.py:193), in
    raster order over the positions where the window fits (:187-192); the
    minimum of the *working* DEM window (:189-191), every position equal to it
    (np.nonzero, :192) marked 1 and raised to 10000 in the working DEM
    (:196-197)."""
    m = np.asarray(mask_rivers).astype(np.float32)
    work = np.asarray(dem).astype(np.float32)          # deep copy + float32 read
    r = window // 2
    h, w_ = m.shape
    routed = np.zeros(work.shape)
    for j in range(r, h - r):
        for i in range(r, w_ - r):
            if int(m[j, i]) != 1:
                continue
            win = work[j - r:j + r + 1, i - r:i + r + 1].copy()
            low = np.amin(win)
            for dj in range(window):
                for di in range(window):
                    if win[dj, di] == low:
                        routed[j - r + dj, i - r + di] = 1
                        work[j - r + dj, i - r + di] = 10000
    return routed


def process_rivers(rivers, hsheds):
    """ProcessRivers (:770-798): MaskPositives -> ExpandFilter(3) ->
    RouteRivers(3, hsheds) -> scipy binary_closing (cross structure).  Returns
    every stage."""
    positives = (np.asarray(rivers) > 0.0) * 1
    expanded = expand(positives, 3).astype(np.float64)
    routed = route_rivers(expanded, hsheds, 3)
    closing = ndimage.binary_closing(routed)
    return positives, expanded, routed, closing


def clip_lagoons_rivers(mask_lagoons, rivers_routed_closing):
    """ClipLagoonsRivers (:801-831) applied to the closing itself, as
    `image_hsheds.py:204-205` does: ``closing XOR (mask_lagoons * closing)``."""
    inter = mask_lagoons * rivers_routed_closing
    return np.bitwise_xor(rivers_routed_closing, inter)
