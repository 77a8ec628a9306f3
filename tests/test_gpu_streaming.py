"""
Band streaming on the GPU: a raster processed band by band through
pinned buffers and two streams gives the whole-raster result -- bit for bit for
the box mean and D8, to 1e-4 m for groves (per-strip reference level).
"""
import numpy as np
import pytest

from hydrodem_amd import backend, streaming as S
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(built):
    assert backend.device_count() >= 1


@pytest.mark.parametrize("band_rows,depth", [(100, 2), (257, 3), (1000, 1), (5000, 2)])
def test_streamed_boxmean_and_d8_equal_the_whole_raster(band_rows, depth):
    z = oracle.synth_dem(1111, 640)
    whole = backend.DeviceRaster.from_host(z)
    want_box = backend.boxmean3_dev(whole).to_host()
    want_d8 = backend.d8_dev(whole).to_host()
    out = np.empty_like(z)
    with S.BandStream(z.shape, band_rows=band_rows, depth=depth, **S.boxmean_op()) as bs:
        bs.run([z], out)
    assert np.array_equal(out, want_box)
    codes = np.empty(z.shape, dtype=np.uint8)
    with S.BandStream(z.shape, band_rows=band_rows, depth=depth, **S.d8_op()) as bs:
        bs.run([z], codes)
        bs.run([z], codes)                                # a stream can be reused
    assert np.array_equal(codes, want_d8)


def test_streamed_groves_from_a_memmap_with_callable_io(tmp_path):
    h, w = 900, 512
    img, mask = oracle.synth_dem(h, w, pits=False), oracle.synth_groves(h, w)
    src = np.memmap(tmp_path / "img.f32", dtype=np.float32, mode="w+", shape=(h, w))
    src[:] = img
    src.flush()
    dst = np.memmap(tmp_path / "out.f32", dtype=np.float32, mode="w+", shape=(h, w))
    reads, writes = [], []

    def read_mask(lo, hi, view):                          # what a GDAL window read would do
        reads.append((lo, hi))
        view[:] = mask[lo:hi]

    def write(r0, r1, view):
        writes.append((r0, r1))
        dst[r0:r1] = view

    with S.BandStream((h, w), band_rows=128, depth=2, **S.groves_op(3)) as bs:
        bs.run([src, read_mask], write)
    want = backend.groves_dev(backend.DeviceRaster.from_host(img),
                              backend.DeviceRaster.from_host(mask)).to_host()
    assert np.abs(np.asarray(dst) - want).max() <= 1e-4
    assert writes == [(r, min(r + 128, h)) for r in range(0, h, 128)]      # in order, once
    assert reads[0] == (0, 128 + 21) and reads[1] == (128 - 21, 256 + 21)
