"""
CPU oracle (NumPy) for the raster hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  Nothing under ``hydrodem_amd/`` does:
the product path is the HIP library and fails loudly without it.

Each function restates one reference operator (file:line cited, relative to
the reference checkout) in vectorised NumPy, or -- for the two operators the
reference does not implement -- is the normative definition the HIP kernels
are held to.

Pinning status
--------------
* ``quadratic_*``, ``groves_*``, ``boxmean3_round`` : PINNED.  Checked
  cell-for-cell against outputs of the imported reference operators
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the
  reference's own expected rasters (``tests_expected.zip``, the derived
  groves known-answer of SURVEY 8c-iii).
* ``sinkfill_*``, ``d8_flow_direction`` : PARITY UNPINNED.  The reference has
  no sink-fill and no D8 code, test or fixture.  Their credibility
  rests on property tests, hand-checked micro grids and two independent
  algorithms (Jacobi relaxation here, priority-flood in ``hdem_oracle.c``)
  agreeing bit for bit.
"""

import numpy as np

# ---------------------------------------------------------------------------
# A1  sink fill  (no reference code; spec: SURVEY 8a row A1)
# ---------------------------------------------------------------------------

_NEIGH = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))


def sinkfill_init(z):
    """Start state of the relaxation: W = Z on every pinned cell, +inf on the
    rest.  Pinned = the one-cell raster border (outlets; mirrors the
    "interior only, border untouched" convention of
    `sliding_window.py:187-192`), nodata (NaN) cells themselves (kept NaN) and
    every cell 8-adjacent to a nodata cell (nodata is "outside": an outlet)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    h, w_ = z.shape
    w = np.full_like(z, np.inf)
    w[0, :] = z[0, :]
    w[-1, :] = z[-1, :]
    w[:, 0] = z[:, 0]
    w[:, -1] = z[:, -1]
    nan = np.isnan(z)
    if nan.any():
        near = nan.copy()
        pad = np.pad(nan, 1, constant_values=False)
        for dy, dx in _NEIGH:
            near |= pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w_]
        w[near] = z[near]
    return w


def _min8(w):
    """Minimum over the 8 neighbours of every interior cell, NaN ignored."""
    h, w_ = w.shape
    m = None
    for dy, dx in _NEIGH:
        v = w[1 + dy:h - 1 + dy, 1 + dx:w_ - 1 + dx]
        m = v.copy() if m is None else np.fmin(m, v)
    return m


def sinkfill_sweep(z, w, eps=0.0):
    """One Jacobi sweep  W[c] <- max(Z[c], min(W[c], min8(W[n] + eps)))  on
    the interior; returns (new W, number of changed cells).  float32
    throughout, one rounding in ``W[n] + eps``."""
    eps = np.float32(eps)
    cand = _min8(w)
    if eps != 0:
        cand = (cand + eps).astype(np.float32)
    zi = z[1:-1, 1:-1]
    wi = w[1:-1, 1:-1]
    new = np.fmax(zi, np.fmin(wi, cand))
    nanz = np.isnan(zi)
    if nanz.any():
        new = np.where(nanz, wi, new)
    changed = int(np.count_nonzero(new != wi) - np.count_nonzero(nanz))
    out = w.copy()
    out[1:-1, 1:-1] = new
    return out, changed


def sinkfill_jacobi(z, eps=0.0, max_sweeps=None):
    """Normative sink fill: iterate :func:`sinkfill_sweep` from
    :func:`sinkfill_init` until no cell changes.  The limit is the greatest
    fixed point of the sweep operator = spill elevation
    ``min over 8-connected paths to a pinned cell of max(Z along the path)``
    (for eps = 0), and does not depend on the update order.  Returns
    (W float32, sweeps)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    w = sinkfill_init(z)
    sweeps = 0
    if min(z.shape) < 3:
        return w, 0
    while True:
        w, changed = sinkfill_sweep(z, w, eps)
        sweeps += 1
        if changed == 0:
            return w, sweeps
        if max_sweeps is not None and sweeps >= max_sweeps:
            return w, sweeps


def sinkfill_is_fixed_point(z, w, eps=0.0):
    """True when one more sweep changes nothing (size-independent check)."""
    _, changed = sinkfill_sweep(np.ascontiguousarray(z, np.float32),
                                np.ascontiguousarray(w, np.float32), eps)
    return changed == 0


# ---------------------------------------------------------------------------
# A2  D8 flow direction  (no reference code; spec: SURVEY 8a row A2.  The tie
#     rule -- first minimum in row-major window order -- is the one
#     `custom_filters.py:193-195` gets from ``np.nonzero``.)
# ---------------------------------------------------------------------------

# window order NW, N, NE, W, E, SW, S, SE and the ESRI code of each
D8_OFFSETS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0),
              (1, 1))
D8_CODES = (32, 64, 128, 16, 1, 8, 4, 2)
D8_DIAG = np.float32(0.70710678)


def d8_flow_direction(z):
    """ESRI D8 code of the steepest positive drop ``(z_c - z_k) * w_k`` with
    w = 1 (cardinal) or float32(0.70710678) (diagonal), all in float32, one
    rounding per operation; first neighbour in window order wins ties; no
    positive drop -> 0; border cells -> 0; NaN never compares greater, so a
    nodata centre gives 0 and a nodata neighbour is never chosen."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    h, w_ = z.shape
    out = np.zeros((h, w_), dtype=np.uint8)
    if h < 3 or w_ < 3:
        return out
    zc = z[1:-1, 1:-1]
    best = np.zeros_like(zc)
    code = np.zeros(zc.shape, dtype=np.uint8)
    with np.errstate(invalid="ignore"):
        for (dy, dx), c in zip(D8_OFFSETS, D8_CODES):
            zk = z[1 + dy:h - 1 + dy, 1 + dx:w_ - 1 + dx]
            drop = zc - zk
            if dy != 0 and dx != 0:
                drop = drop * D8_DIAG
            better = drop > best
            best = np.where(better, drop, best)
            code = np.where(better, np.uint8(c), code)
    out[1:-1, 1:-1] = code
    return out


# ---------------------------------------------------------------------------
# A3  QuadraticFilter  (custom_filters.py:226-257)
# ---------------------------------------------------------------------------

def quadratic_constants(ws=15):
    """v, r0..r3 and the equivalent correlation kernel K = a(xx^2+yy^2)+b
    (custom_filters.py:240-246,255-256)."""
    v = np.linspace(-ws / 2 + 1, ws / 2, ws)
    xx, yy = np.meshgrid(v, v)
    r0 = float(ws ** 2)
    r1 = float((xx * xx).sum())
    r2 = float((xx ** 4).sum())
    r3 = float((xx * xx * yy * yy).sum())
    den = 2 * r1 ** 2 - r0 * (r2 + r3)
    a = r1 / den
    b = -(r2 + r3) / den
    return v, (r0, r1, r2, r3, den), (a, b)


def _window_sums(g, ws, wx):
    """Separable correlation sum_{dy,dx} g[y+dy, x+dx] * wy? -- helper:
    returns the valid-region sliding sums of ``g`` weighted along x by
    ``wx`` (1-D) and unweighted along y is NOT applied here."""
    h, w_ = g.shape
    n = w_ - ws + 1
    acc = np.zeros((h, n), dtype=np.float64)
    for k in range(ws):
        acc += g[:, k:k + n] * wx[k]
    return acc


def quadratic_exact64(dem, ws=15):
    """QuadraticFilter in exact-as-possible float64 arithmetic on the
    float32-rounded grid (the reference reads every window through
    ``grid.astype('float32')``, sliding_window.py:132).  Border ring of
    ``ws // 2`` cells returned unchanged (custom_filters.py:249).  Output
    float64; compare with a tolerance -- the reference itself accumulates s1
    in float32 and sits ~5e-5 m from this."""
    dem = np.asarray(dem)
    g = dem.astype(np.float32).astype(np.float64)
    h, w_ = g.shape
    v, (r0, r1, r2, r3, den), _ = quadratic_constants(ws)
    ones = np.ones(ws)
    v2 = v * v
    ny = h - ws + 1

    def colsum(rows, wy):
        acc = np.zeros((ny, rows.shape[1]), dtype=np.float64)
        for k in range(ws):
            acc += rows[k:k + ny, :] * wy[k]
        return acc

    rs0 = _window_sums(g, ws, ones)      # sum_x w
    rs2 = _window_sums(g, ws, v2)        # sum_x w x^2
    s1 = colsum(rs0, ones)
    s2 = colsum(rs2, ones)               # sum w xx^2
    s3 = colsum(rs0, v2)                 # sum w yy^2
    val = ((s2 + s3) * r1 - s1 * (r2 + r3)) / den
    out = dem.astype(np.float64).copy()
    p = ws // 2
    out[p:h - p, p:w_ - p] = val
    return out


# ---------------------------------------------------------------------------
# A4  GrovesCorrection / GrovesCorrectionsIter  (custom_filters.py:696-767,
#     MaskTallGroves :533-534)
# ---------------------------------------------------------------------------

def groves_pass(img, groves, smooth, thr=1.5):
    """The mask algebra of one GrovesCorrection given the smoothed image:
    hl = img - smooth; m = groves * (hl > thr); out = hl*(1-m) + smooth
    (custom_filters.py:725-732).  Evaluated in the dtype of the operands."""
    hl = img - smooth
    tall = (hl > thr) * 1
    m = groves * tall
    return hl * (1 - m) + smooth, m


def groves_exact64(img, groves, iterations=3, ws=15, thr=1.5):
    """GrovesCorrectionsIter with the float64 quadratic; returns
    (out float64, list of per-iteration (highlight, mask))."""
    cur = np.asarray(img, dtype=np.float32).astype(np.float64)
    groves = (np.asarray(groves) != 0).astype(np.int64)
    stages = []
    for _ in range(iterations):
        smooth = quadratic_exact64(cur, ws)
        out, m = groves_pass(cur, groves, smooth, thr)
        stages.append((cur - smooth, m))
        cur = out
    return cur, stages


# ---------------------------------------------------------------------------
# A5  PostProcessingFinal = Convolve + Around
#     (custom_filters.py:1124-1125; extension_filters.py:166-184,113-130)
# ---------------------------------------------------------------------------

def boxmean3_round(x):
    """3x3 sum with edge-inclusive mirror border (SciPy ``mode='reflect'``),
    accumulated in double in raster order of the window, cast to the input
    dtype, divided by 9 in that dtype, rounded half-to-even.  Pure NumPy
    restatement of ``scipy.ndimage.convolve(x, ones((3,3))) / 9`` followed
    by ``np.around``."""
    x = np.asarray(x)
    dt = x.dtype if x.dtype in (np.float32, np.float64) else np.float64
    p = np.pad(x.astype(np.float64), 1, mode="symmetric")
    h, w_ = x.shape
    acc = np.zeros((h, w_), dtype=np.float64)
    for dy in range(3):
        for dx in range(3):
            acc += p[dy:dy + h, dx:dx + w_]
    s = acc.astype(dt)
    return np.around(s / dt.type(9) if dt == np.float32 else s / 9)


def boxmean3(x):
    """The Convolve() half alone (mean, no rounding)."""
    x = np.asarray(x)
    dt = x.dtype if x.dtype in (np.float32, np.float64) else np.float64
    p = np.pad(x.astype(np.float64), 1, mode="symmetric")
    h, w_ = x.shape
    acc = np.zeros((h, w_), dtype=np.float64)
    for dy in range(3):
        for dx in range(3):
            acc += p[dy:dy + h, dx:dx + w_]
    s = acc.astype(dt)
    return s / dt.type(9) if dt == np.float32 else s / 9


# ---------------------------------------------------------------------------
# Synthetic rasters: generated by hdem_synth.py (repo root; shared with bench.py, which
# must not depend on this package for its inputs), re-exported here for the tests.
# ---------------------------------------------------------------------------
from hdem_synth import GEN_BLOCK, GEN_SEED, synth_dem, synth_groves  # noqa: E402,F401
