"""
Band streaming on the GPU: a raster processed band by band through
pinned buffers, one stream and one host thread per slot, against the CPU oracle of the
whole raster -- bit for bit for the box mean and D8, to 1e-4 m for groves (the bar of
SURVEY 8d; cells whose highlight sits on the 1.5 m threshold counted) -- and against the
whole-raster GPU run (same bits for the box mean and D8).
"""
import numpy as np
import pytest

from hydrodem_amd import backend, streaming as S
import oracle
from oracle import c_oracle

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
    assert np.array_equal(out, c_oracle.boxmean3(z, True))              # the oracle, not only the GPU
    codes = np.empty(z.shape, dtype=np.uint8)
    with S.BandStream(z.shape, band_rows=band_rows, depth=depth, **S.d8_op()) as bs:
        bs.run([z], codes)
        bs.run([z], codes)                                # a stream can be reused
    assert np.array_equal(codes, want_d8)
    assert np.array_equal(codes, c_oracle.d8(z))


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
    ref = c_oracle.groves_ref(img, mask, 3)                            # reference restatement
    assert (np.abs(np.asarray(dst) - ref) > 1e-4).sum() <= 2
    assert writes == [(r, min(r + 128, h)) for r in range(0, h, 128)]      # in order, once
    assert sorted(reads)[:2] == [(0, 128 + 21), (128 - 21, 256 + 21)]


def test_stream_over_a_memmap_of_many_band_sets_against_the_oracle(tmp_path):
    """A file-backed raster of 40 bands through 3 slots (13 rounds of the slot ring), three
    operators, each against the CPU oracle of the whole raster."""
    h, w = 2500, 700
    z = oracle.synth_dem(h, w)
    src = np.memmap(tmp_path / "z.f32", dtype=np.float32, mode="w+", shape=(h, w))
    src[:] = z
    src.flush()
    for kwargs, dtype, want in ((S.boxmean_op(), np.float32, c_oracle.boxmean3(z, True)),
                                (S.d8_op(), np.uint8, c_oracle.d8(z))):
        dst = np.memmap(tmp_path / "o.bin", dtype=dtype, mode="w+", shape=(h, w))
        with S.BandStream((h, w), band_rows=64, depth=3, **kwargs) as bs:
            bs.run([src], dst)
        assert np.array_equal(np.asarray(dst), want)
    mask = oracle.synth_groves(h, w)
    out = np.empty((h, w), dtype=np.float32)
    with S.BandStream((h, w), band_rows=64, depth=3, **S.groves_op(3)) as bs:
        bs.run([src, mask], out)
        serial = np.empty_like(out)
        bs.run([src, mask], serial, threads=False)            # same bits without the threads
    assert np.array_equal(out, serial)
    assert (np.abs(out - c_oracle.groves_ref(z, mask, 3)) > 1e-4).sum() <= 2


def test_a_failing_reader_stops_the_stream():
    z = oracle.synth_dem(600, 300)

    def read(lo, hi, view):
        if lo > 200:
            raise OSError("read failed")
        view[:] = z[lo:hi]

    with S.BandStream(z.shape, band_rows=100, depth=2, **S.boxmean_op()) as bs:
        with pytest.raises(OSError, match="read failed"):
            bs.run([read], np.empty_like(z))
        out = np.empty_like(z)
        bs.run([z], out)                                      # and the stream is still usable
    assert np.array_equal(out, c_oracle.boxmean3(z, True))
