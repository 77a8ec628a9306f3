"""
Oracle of the Fourier destripe chain against the golden vectors the
imported reference produced (tests/golden/make_golden_fourier.py): every stage
bit for bit -- the restatement uses the same scipy.fftpack transform, so even
the final raster is identical.  CPU only.
"""
import numpy as np
import pytest

from oracle import hdem_oracle_fourier as F


@pytest.fixture(scope="module")
def fz(golden):
    return golden("fourier.npz")


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_every_stage_matches_the_reference(fz, tag):
    dem = fz[f"{tag}_dem"]
    spec, mag = F.fourier_initial(dem)
    assert spec.dtype == np.complex64 and np.array_equal(mag, fz[f"{tag}_mag"])
    s1, s2 = F.quarter_slices(*dem.shape)
    found, q_mod, margin = F.blanks_fourier(mag[s1])
    assert np.array_equal(found, fz[f"{tag}_found1"])
    assert np.array_equal(q_mod, fz[f"{tag}_q1_mod"])
    # no decision of this fixture sits on the threshold (float32-pairwise vs float64 sums)
    assert np.min(np.abs(margin) / np.maximum(mag[s1], 1e-30)) > 1e-4
    d1, _ = F.detect_blanks_fourier(mag[s1])
    d2, _ = F.detect_blanks_fourier(mag[s2])
    assert np.array_equal(d1, fz[f"{tag}_det1"]) and np.array_equal(d2, fz[f"{tag}_det2"])
    assert d2.sum() > 0                                     # both quadrants carry peaks
    iso = F.isolated_points(d1)
    assert np.array_equal(iso, fz[f"{tag}_iso1"]) and iso.sum() < d1.sum()
    assert np.array_equal(F.expand(iso), fz[f"{tag}_exp1"])
    out, mask, _ = F.detect_apply_fourier(dem)
    assert np.array_equal(mask, fz[f"{tag}_mask"])
    assert out.dtype == np.float64 and np.array_equal(out, fz[f"{tag}_result"])
    # the destripe took the plane waves out
    assert np.abs(out - dem).max() > 0.5


def test_mask_stencils_match_the_reference(fz):
    m = fz["st_mask"].astype(np.float64)
    iso = F.isolated_points(m)
    assert np.array_equal(iso, fz["st_iso"])
    assert iso[0, 5] == 1 and iso[39, 46] == 1 and iso[17, 0] == 1   # border cells untouched
    assert np.array_equal(F.expand(iso), fz["st_exp"])
    assert np.array_equal(F.expand(m, 5), fz["st_exp5"])


def test_mask_is_point_mirrored():
    ny, nx = 150, 168
    m1 = np.zeros((ny // 2 - 10, nx // 2 - 10)); m1[3, 4] = 1
    m2 = np.zeros_like(m1); m2[7, 9] = 1
    full = F.assemble_mask(ny, nx, m1, m2)
    assert full.sum() == 4
    assert full[3, 4] == 1 and full[ny - 1 - 3, nx - 1 - 4] == 1
    X = nx // 2 + 10 + 9
    assert full[7, X] == 1 and full[ny - 1 - 7, nx - 1 - X] == 1
