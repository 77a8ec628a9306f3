"""
Parity of the Fourier destripe chain on the HIP path, through the C
ABI, against the golden vectors of the imported reference and the CPU oracle.

Bars.  Byte masks: bit-exact, except cells whose decision margin |q - 4 mean| is
within 1e-4 of q (the transform is rocFFT's, not fftpack's: magnitudes differ in
the last bits) -- counted, none in the fixtures.  Destriped elevations: <= 1e-4 m
(complex64 both ways on dem - mean; the reference's inverse runs in complex128).
"""
import numpy as np
import pytest
from scipy import fftpack

import hydrodem_amd as hd
from hydrodem_amd import backend
from oracle import hdem_oracle_fourier as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def fz(golden, built):
    assert backend.device_count() >= 1
    return golden("fourier.npz")


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_blanks_pass_matches_the_reference(fz, tag):
    mag = fz[f"{tag}_mag"]
    s1, _ = F.quarter_slices(*mag.shape)
    found, q_mod = hd.BlanksFourier(window_size=55).apply(mag[s1].copy())
    assert found.dtype == np.float64 and np.array_equal(found, fz[f"{tag}_found1"])
    assert np.array_equal(q_mod, fz[f"{tag}_q1_mod"])
    det = hd.DetectBlanksFourier().apply(mag[s1].copy())
    assert np.array_equal(det, fz[f"{tag}_det1"])


def test_mask_stencils_match_the_reference(fz):
    m = fz["st_mask"].astype(np.float64)
    same = m.copy()
    iso = hd.IsolatedPoints(window_size=3).apply(same)
    assert iso is same and np.array_equal(iso, fz["st_iso"])          # in place, like the reference
    assert np.array_equal(hd.ExpandFilter(window_size=13).apply(iso), fz["st_exp"])
    assert np.array_equal(hd.ExpandFilter(window_size=5).apply(m), fz["st_exp5"])
    for tag in ("even", "odd"):
        iso = hd.IsolatedPoints(window_size=3).apply(fz[f"{tag}_det1"].astype(np.float64))
        assert np.array_equal(iso, fz[f"{tag}_iso1"])
        assert np.array_equal(hd.ExpandFilter(window_size=13).apply(iso), fz[f"{tag}_exp1"])
        mask = hd.MaskFourier().apply(fz[f"{tag}_mag"][F.quarter_slices(*fz[f"{tag}_mag"].shape)[0]].copy())
        assert np.array_equal(mask, fz[f"{tag}_exp1"])


@pytest.mark.parametrize("shape", [(8, 8), (150, 168), (141, 155), (97, 1031), (512, 768)])
def test_fft_matches_fftpack(shape):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(shape) * 10 + 100).astype(np.float32)
    want = fftpack.fft2(x)
    got = hd.FourierTransform().apply(x)
    assert got.dtype == np.complex64 and got.shape == shape
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale
    back = hd.FourierITransform().apply(got)
    assert np.abs(back.real - x).max() <= 2e-6 * np.abs(x).max()
    assert np.abs(back.imag).max() <= 2e-6 * np.abs(x).max()


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_destripe_matches_the_reference(fz, tag):
    dem = fz[f"{tag}_dem"]
    f = hd.DetectApplyFourier(keep_mask=True)
    out = f.apply(dem)
    assert out.dtype == np.float32 and out.shape == dem.shape
    assert np.array_equal(f.mask, fz[f"{tag}_mask"])
    assert np.abs(out - fz[f"{tag}_result"]).max() <= TOL
    # same through the pieces, as the reference's orchestration would call them
    init = hd.FourierInitial()
    mag = init.apply(dem)
    assert np.abs(mag - fz[f"{tag}_mag"]).max() <= 2e-6 * fz[f"{tag}_mag"].max()
    assert np.array_equal(hd.FourierProcessQuarters(fz[f"{tag}_mag"]).apply(None), fz[f"{tag}_mask"])
    assert np.array_equal(hd.DetectApplyFourier().apply(dem), out)


def test_destripe_on_a_larger_raster_against_the_oracle():
    dem = F.synth_striped_dem(700, 1030, seed=3,
                              stripes=((0.31, 0.07, 1.2), (0.12, -0.38, 0.8), (0.05, 0.45, 0.6)))
    want, want_mask, stages = F.detect_apply_fourier(dem)
    out, mask = backend.fourier_destripe(dem, return_mask=True)
    diff = mask != want_mask
    if diff.any():
        # only threshold-borderline decisions may differ (and what the dilation makes of them)
        borderline = sum(int((np.abs(g) <= 1e-4 * np.abs(g).max()).sum())
                         for pair in stages["margins"] for g in pair)
        assert borderline > 0 and diff.sum() <= 330 * borderline
    else:
        assert np.abs(out - want).max() <= TOL
    # the plane waves are gone: their spectral lines dropped by > 100x
    spec_in, spec_out = np.abs(fftpack.fft2(dem)), np.abs(fftpack.fft2(out))
    for fy, fx in ((0.31, 0.07), (0.12, -0.38), (0.05, 0.45)):
        ky, kx = int(round(fy * 700)) % 700, int(round(fx * 1030)) % 1030
        win = (slice(max(ky - 2, 0), ky + 3), slice(max(kx - 2, 0), kx + 3))
        assert spec_out[win].max() < 0.01 * spec_in[win].max()


@pytest.mark.parametrize("shape", [(512, 1024), (256, 256)])
def test_destripe_power_of_two_sizes_take_the_split_inverse(shape, monkeypatch):
    """For power-of-two sizes the inverse transform runs as two batched 1-D passes with
    own transposes (the last one fused with the abs) instead of rocFFT's 2-D plan: same
    mask as the oracle, elevations within the bar, and the same result as the 2-D plan
    (HDEM_FFT_2D_INVERSE) to rounding."""
    dem = F.synth_striped_dem(*shape, seed=9)
    want, want_mask, _ = F.detect_apply_fourier(dem)
    out, mask = backend.fourier_destripe(dem, return_mask=True)
    assert np.array_equal(mask, want_mask)
    assert np.abs(out - want).max() <= TOL
    monkeypatch.setenv("HDEM_FFT_2D_INVERSE", "1")
    out2 = backend.fourier_destripe(dem)
    assert np.abs(out - out2).max() <= 2e-5


def test_destripe_rejects_rasters_whose_quadrants_are_smaller_than_the_window():
    with pytest.raises(hd.WindowSizeHighError) as e:
        hd.DetectApplyFourier().apply(np.zeros((120, 200), dtype=np.float32))
    assert str(e.value) == "Window size: 55 cannot be higher than grid dimensions: (50, 90)"
    with pytest.raises(hd.NumpyArrayExpectedError):
        hd.DetectApplyFourier().apply([[1.0]])
    with pytest.raises(hd.WindowSizeEvenError):
        hd.ExpandFilter(window_size=4).apply(np.zeros((9, 9)))


def test_destripe_device_chain():
    dem = F.synth_striped_dem(256, 300, seed=1)
    chain = hd.ComposedFilter()
    chain.filters = [hd.DetectApplyFourier(), hd.PostProcessingFinal()]
    got = chain.apply(dem)
    assert np.array_equal(got, hd.PostProcessingFinal().apply(hd.DetectApplyFourier().apply(dem)))
