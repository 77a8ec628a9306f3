"""
The BASELINE.json size (16384 x 16384) on the GPU.  The C priority-flood oracle
still finishes here (~30 s), so the sink fill and D8 are compared bit for bit in
full; the windowed operators are compared with the oracle on crops (they are
local: a crop with a halo of the window reach reproduces the same cells), and
through size-independent properties.
"""
import numpy as np
import pytest

from hydrodem_amd import backend
import oracle
from oracle import c_oracle


pytestmark = pytest.mark.gpu
N = 16384


@pytest.fixture(scope="module")
def dem(built):
    assert backend.device_count() >= 1
    return oracle.synth_dem(N, N)


def test_sinkfill_and_d8_full_size_bit_exact(dem):
    zd = backend.DeviceRaster.from_host(dem)
    wd, codes, st = backend.sinkfill_d8_dev(zd)               # the headline step
    assert st["converged"] and st["async_timed_out"] == 0
    w = wd.to_host()
    want = c_oracle.sinkfill_pflood(dem)
    assert np.array_equal(w, want)
    assert np.array_equal(codes.to_host(), c_oracle.d8(want))
    assert (w >= dem).all() and np.array_equal(w[0], dem[0]) and np.array_equal(w[:, 0], dem[:, 0])
    # idempotent: filling the filled surface certifies without lowering anything
    _, st2 = backend.sinkfill_dev(wd, out=backend.DeviceRaster.empty(dem.shape, np.float32))
    d = backend.d8_dev(wd).to_host()
    assert np.array_equal(d, c_oracle.d8(want))
    # a filled surface has no interior pit: code 0 only where no neighbour is lower
    assert set(np.unique(d)) <= {0, 1, 2, 4, 8, 16, 32, 64, 128}
    for r in (zd, wd):
        r.free()


def test_groves_and_boxmean_full_size_on_crops(dem):
    groves = oracle.synth_groves(N, N)
    img = backend.DeviceRaster.from_host(dem)
    gd = backend.DeviceRaster.from_host(groves)
    out = backend.groves_dev(img, gd, iterations=3)
    full = out.to_host()
    box = backend.boxmean3_dev(out).to_host()
    rng = np.random.default_rng(8)
    halo = 21
    for _ in range(6):
        y0, x0 = int(rng.integers(0, N - 300)), int(rng.integers(0, N - 300))
        sl = (slice(y0, y0 + 300), slice(x0, x0 + 300))
        want = c_oracle.groves_ref(dem[sl], groves[sl], 3)
        inner = (slice(halo, 300 - halo), slice(halo, 300 - halo))
        err = np.abs(full[sl][inner] - want[inner])
        assert (err > 1e-4).sum() <= 2, "groves differs from the reference restatement"
        # box mean + round of the GPU groves output: exact, one cell of halo
        wb = c_oracle.boxmean3(full[sl], True)
        assert np.array_equal(box[sl][1:-1, 1:-1], wb[1:-1, 1:-1])
    # raster corners: the untouched 7-cell ring of the quadratic filter
    assert np.array_equal(full[:7], dem[:7]) and np.array_equal(full[:, -7:], dem[:, -7:])
    for r in (img, gd, out):
        r.free()


def test_destripe_full_size_properties():
    rng = np.random.default_rng(2)
    z = oracle.synth_dem(N, N, pits=False)
    # one plane wave between frequencies of the grid (a peak exactly on one bin has no
    # neighbour and IsolatedPoints drops it, as in the reference), away from the terrain's
    ky, kx, amp = 3000.4, 5000.3, 1.0
    y = np.arange(N, dtype=np.float64)[:, None]
    x = np.arange(N, dtype=np.float64)[None, :]
    phase = rng.uniform(0, 2 * np.pi)
    stripes = amp * np.sin(2 * np.pi * (ky * y + kx * x) / N + phase)
    dem = (z + stripes).astype(np.float32)
    out, mask = backend.fourier_destripe(dem, return_mask=True)
    assert np.isfinite(out).all()
    # the wave's frequencies are in the mask (shifted coordinates, both mirror images) ...
    assert mask[N // 2 - 3000, N // 2 - 5000] == 1 and mask[N // 2 + 3000, N // 2 + 5000] == 1
    # ... and most of the wave is gone (what leaks past the 13 x 13 dilation stays)
    s = np.sin(2 * np.pi * (ky * y + kx * x) / N + phase)
    assert abs(2 * ((dem.astype(np.float64) - out) * s).mean() - amp) < 0.15
    assert abs(2 * ((out - z) * s).mean()) < 0.15
    # the zero frequency is untouched
    assert abs(out.mean(dtype=np.float64) - dem.mean(dtype=np.float64)) < 1e-3


def test_sinkfill_and_d8_at_the_column_count_of_the_largest_config(built):
    """BASELINE config 5 cuts 65536 x 65536 into row blocks of 8192 x 65536: the width
    (row stride, tile columns, 32-bit offsets inside a window) at full size, on a band
    that the C oracle still fills in seconds."""
    h, w = 700, 65536 + 37
    z = oracle.synth_dem(h, w)
    wd, codes, st = backend.sinkfill_d8_dev(backend.DeviceRaster.from_host(z))
    want = c_oracle.sinkfill_pflood(z)
    assert st["converged"] == 1
    assert np.array_equal(wd.to_host(), want)
    assert np.array_equal(codes.to_host(), c_oracle.d8(want))
    wd.free()
    codes.free()


def test_lagoons_full_size_on_crops():
    """LagoonsDetection at 16384^2 against the oracle on crops: the chain is local
    (reach 1 + 5 + 8 cells), so a crop reproduces every cell further than that from
    its cut sides; corner crops keep the raster's own border, which must match too."""
    from oracle import hdem_oracle_lagoons as L
    hs = np.round(oracle.synth_dem(N, N))
    hs[::97, ::89] = -32768.0                                  # voids
    hs[5000:5400, 7000:7600] = 212.0                           # a lake across tile seams
    hs[:40, :300] = 95.0                                       # and one on the raster's border
    mask, fixed, values = [r.to_host() for r in
                           backend.lagoons_detection_dev(backend.DeviceRaster.from_host(hs))]
    assert mask.sum() > 400 * 600 // 2
    reach, c = 14, 320
    rng = np.random.default_rng(3)
    spots = [(0, 0), (0, N - c), (N - c, 0), (N - c, N - c), (4900, 6900), (5250, 7450)]
    spots += [(int(rng.integers(0, N - c)), int(rng.integers(0, N - c))) for _ in range(3)]
    for y0, x0 in spots:
        sl = (slice(y0, y0 + c), slice(x0, x0 + c))
        want_mask, stages = L.lagoons_detection(hs[sl].copy())
        inner = (slice(0 if y0 == 0 else reach, c if y0 + c == N else c - reach),
                 slice(0 if x0 == 0 else reach, c if x0 + c == N else c - reach))
        assert np.array_equal(fixed[sl][inner], stages["CorrectNANValues"][inner], equal_nan=True)
        assert np.array_equal(values[sl][inner], stages["TidyingLagoons"][inner])
        assert np.array_equal(mask[sl][inner] != 0, want_mask[inner] != 0)
