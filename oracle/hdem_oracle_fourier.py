"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Fourier destripe
chain, vectorised NumPy.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; the product path never does.

Every function names the reference lines it restates
(/root/reference/cguerrero/hydrodem/filters/custom_filters.py unless noted).
Pinned by tests/golden/fourier.npz: outputs of the *imported reference
operators* on seeded inputs (tests/golden/make_golden_fourier.py).

Third-party arithmetic on the path: scipy.fftpack.fft2/ifft2/fftshift/ifftshift
(extension_filters.py:379,414,447,480).  SciPy is not pinned by the reference
(requirements.txt is empty); here SciPy 1.15.3, whose fftpack keeps float32
input in single precision (complex64).
"""

import numpy as np
from scipy import fftpack

MARGIN = 10          # FourierProcessQuarters._margin (:925)
DETECT_WINDOW = 55   # DetectBlanksFourier (:455)
DETECT_INNER = 5     # BlanksFourier (:420)
DETECT_FACTOR = 4    # BlanksFourier (:426)
EXPAND_WINDOW = 13   # MaskFourier (:560)


def fourier_initial(dem):
    """FourierInitial (:834-877): fft2 -> fftshift -> abs.  Returns
    (shifted spectrum, its magnitude)."""
    spec = fftpack.fftshift(fftpack.fft2(dem))
    return spec, np.abs(spec)


def _box_sum(a, size):
    """Sum of ``a`` over the size x size window centred on every cell, zeros
    outside (float64, summed-area table)."""
    h, w_ = a.shape
    r = size // 2
    sat = np.zeros((h + 1, w_ + 1), dtype=np.float64)
    np.cumsum(np.cumsum(a, axis=0, dtype=np.float64), axis=1, out=sat[1:, 1:])
    y0 = np.clip(np.arange(h) - r, 0, h)[:, None]
    y1 = np.clip(np.arange(h) + r + 1, 0, h)[:, None]
    x0 = np.clip(np.arange(w_) - r, 0, w_)[None, :]
    x1 = np.clip(np.arange(w_) + r + 1, 0, w_)[None, :]
    return sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]


def hollow_mean(q, window=DETECT_WINDOW, inner=DETECT_INNER):
    """Mean of the cells of ``q`` inside the window x window square around every
    cell but outside the inner x inner one, cells beyond the array edge left out
    (IgnoreBorderInnerSliding + np.nanmean, :417-423; sliding_window.py:739-793).
    float64; the reference sums float32 windows pairwise, ~1e-6 relative away."""
    v = np.asarray(q).astype(np.float32).astype(np.float64)
    ones = np.ones_like(v)
    total = _box_sum(v, window) - _box_sum(v, inner)
    count = _box_sum(ones, window) - _box_sum(ones, inner)
    with np.errstate(invalid="ignore", divide="ignore"):
        return total / count


def blanks_fourier(q, window=DETECT_WINDOW):
    """BlanksFourier.apply (:400-429): cells brighter than 4x the hollow mean
    -> 1; also returns the image with those cells zeroed.  ``margin`` (float64):
    q - 4*mean, to tell threshold-borderline cells."""
    mean = hollow_mean(q, window)
    margin = q.astype(np.float64) - DETECT_FACTOR * mean
    found = (margin > 0).astype(np.float64)
    return found, q * (1 - found), margin


def detect_blanks_fourier(q):
    """DetectBlanksFourier.apply (:447-462): two detection passes, the second on
    the image with the first pass's cells zeroed; masks added.  Returns
    (mask, [margin of pass 1, margin of pass 2])."""
    total = np.zeros(q.shape)
    margins = []
    for _ in (0, 1):
        found, q, margin = blanks_fourier(q)
        total += found
        margins.append(margin)
    return total, margins


def isolated_points(mask, window=3):
    """IsolatedPoints.apply (:344-366): interior cells equal to 1 keep the 1 only
    if another cell of their window (centre excluded) is > 0.  Jacobi: the
    sliding window reads a snapshot (sliding_window.py:128-132).  The reference
    writes into its input; this returns a copy."""
    m = np.asarray(mask).astype(np.float32)
    out = np.array(mask, dtype=np.float64, copy=True)
    r = window // 2
    h, w_ = m.shape
    if h < window or w_ < window:
        return out
    nb = _box_sum((m > 0).astype(np.float64), window) - (m > 0)
    inner = np.zeros_like(m, dtype=bool)
    inner[r:h - r, r:w_ - r] = True
    ones = inner & (m.astype(np.int64) == 1)
    out[ones] = np.where(nb[ones] > 0, 1.0, 0.0)
    return out


def expand(mask, window=EXPAND_WINDOW):
    """ExpandFilter.apply (:103-125): 1 where the window x window square minus
    its four corner cells (CircularWindow, sliding_window.py:475-499) holds a
    cell > 0; only centres whose window fits, the rest 0."""
    m = (np.asarray(mask).astype(np.float32) > 0).astype(np.float64)
    h, w_ = m.shape
    r = window // 2
    out = np.zeros((h, w_))
    if h < window or w_ < window:
        return out
    cnt = _box_sum(m, window)
    pad = np.pad(m, r)
    corners = (pad[0:h, 0:w_] + pad[0:h, 2 * r:2 * r + w_] +
               pad[2 * r:2 * r + h, 0:w_] + pad[2 * r:2 * r + h, 2 * r:2 * r + w_])
    hit = (cnt - corners) > 0.5
    out[r:h - r, r:w_ - r] = hit[r:h - r, r:w_ - r]
    return out


def mask_fourier(q):
    """MaskFourier (:537-561): detect -> isolated points -> expand."""
    detected, margins = detect_blanks_fourier(q)
    return expand(isolated_points(detected)), detected, margins


def quarter_slices(ny, nx):
    """The two upper quarters FourierProcessQuarters looks at (:953-966)."""
    mid_y, mid_x, x_odd = ny // 2, nx // 2, nx % 2
    first = (slice(0, mid_y - MARGIN), slice(0, mid_x - MARGIN))
    second = (slice(0, mid_y - MARGIN), slice(mid_x + MARGIN + x_odd, nx))
    return first, second


def assemble_mask(ny, nx, m1, m2):
    """_fill_complete_quarters, _getting_reversed_masks, _fill_complete_mask
    (:985-1050): quarter masks back at their place, point-mirrored copies in the
    lower half (second -> lower left, first -> lower right)."""
    mid_y, y_odd = divmod(ny, 2)
    mid_x, x_odd = divmod(nx, 2)
    q1 = np.zeros((mid_y, mid_x))
    q2 = np.zeros((mid_y, mid_x))
    q1[:mid_y - MARGIN, :mid_x - MARGIN] = m1
    q2[:mid_y - MARGIN, MARGIN:mid_x] = m2
    full = np.zeros((ny, nx))
    full[:mid_y, :mid_x] = q1
    full[:mid_y, mid_x + x_odd:] = q2
    full[mid_y + y_odd:, :mid_x] = q2[::-1, ::-1]
    full[mid_y + y_odd:, mid_x + x_odd:] = q1[::-1, ::-1]
    return full


def fourier_mask(mag):
    """FourierProcessQuarters.apply (:927-951) on the shifted magnitude.
    Returns (full mask, dict of per-quarter stages)."""
    ny, nx = mag.shape
    s1, s2 = quarter_slices(ny, nx)
    m1, d1, g1 = mask_fourier(mag[s1])
    m2, d2, g2 = mask_fourier(mag[s2])
    return assemble_mask(ny, nx, m1, m2), {"detected": (d1, d2), "margins": (g1, g2),
                                           "masks": (m1, m2)}


def detect_apply_fourier(dem):
    """DetectApplyFourier.apply (:1083-1101): spectrum, mask, (1 - mask) *
    shifted spectrum, ifftshift, ifft2, abs."""
    spec, mag = fourier_initial(dem)
    mask, stages = fourier_mask(mag)
    out = np.abs(fftpack.ifft2(fftpack.ifftshift((1 - mask) * spec)))
    return out, mask, stages


from hdem_synth import synth_striped_dem  # noqa: E402,F401  (re-exported for the tests)
