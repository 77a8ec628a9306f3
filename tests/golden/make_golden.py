"""
Regenerates tests/golden/*.npz.  RUNS ONLY IN THE BUILD CONTAINER, where the
reference checkout is mounted read-only at /root/reference; nothing here is
needed (or available) on the GPU box -- the tests read the committed .npz.

What it stores is data: seeded inputs and the outputs the *imported reference
operators* produce for them, plus crops of rasters the reference's own test
suite holds (tests/resources/tests_expected.zip, read with Pillow).

    python tests/golden/make_golden.py
"""
import io
import os
import sys
import zipfile

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from filters.custom_filters import (QuadraticFilter, GrovesCorrection,  # noqa: E402
                                    GrovesCorrectionsIter,
                                    PostProcessingFinal)
from filters.extension_filters import Convolve, Around  # noqa: E402
import sliding_window as ref_sw  # noqa: E402
from oracle.hdem_oracle_np import synth_dem  # noqa: E402


def tif(zf, name):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(zf.read(name))))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in arrays.items()},
          os.path.getsize(path) // 1024, "KiB")


def main():
    rng = np.random.default_rng(424242)

    # --- A3 QuadraticFilter ------------------------------------------------
    dem = synth_dem(56, 72)
    q15 = QuadraticFilter(window_size=15).apply(dem)
    dem_srtm = synth_dem(48, 40, variant="srtm")
    q15s = QuadraticFilter(window_size=15).apply(dem_srtm)
    q5 = QuadraticFilter(window_size=5).apply(dem[:20, :24].copy())
    dem64 = dem[:40, :40].astype(np.float64) + 1e-9
    q15_64 = QuadraticFilter(window_size=15).apply(dem64)
    save("quadratic.npz", dem=dem, q15=q15, dem_srtm=dem_srtm, q15s=q15s,
         q5_in=dem[:20, :24].copy(), q5=q5, dem64=dem64, q15_64=q15_64)

    # --- A4 GrovesCorrection / GrovesCorrectionsIter -------------------------
    img = synth_dem(64, 80)
    # bumps so that highlight > 1.5 happens inside and outside the mask
    bump = rng.random(img.shape) < 0.04
    img = (img + np.where(bump, rng.uniform(1.0, 6.0, img.shape), 0.0)
           ).astype(np.float32)
    groves_bool = rng.random(img.shape) < 0.35
    g1 = GrovesCorrection(groves_bool)
    out1 = g1.apply(img)
    part = g1.partial_results
    out3 = GrovesCorrectionsIter(groves_bool, iterations=3).apply(img)
    save("groves.npz", img=img, groves=groves_bool.astype(np.uint8),
         out1=out1, smooth1=part[0], highlight1=part[1],
         mask1=np.asarray(part[3]).astype(np.uint8), out3=out3)

    # --- A5 PostProcessingFinal = Convolve + Around ---------------------------
    x32 = synth_dem(37, 53)
    xint = np.round(synth_dem(33, 47) * 1.0).astype(np.float32)  # .5 ties
    x64 = (synth_dem(29, 31).astype(np.float64)
           + rng.standard_normal((29, 31)) * 1e-3)
    save("boxmean.npz",
         x32=x32, conv32=Convolve().apply(x32),
         final32=PostProcessingFinal().apply(x32),
         xint=xint, convint=Convolve().apply(xint),
         finalint=PostProcessingFinal().apply(xint),
         x64=x64, conv64=Convolve().apply(x64),
         final64=PostProcessingFinal().apply(x64),
         around_in=np.array([[0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 3.5001]],
                            dtype=np.float32),
         around_out=Around().apply(np.array(
             [[0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 3.5001]],
             dtype=np.float32)))

    # --- A6 SlidingWindow family: every window the reference yields -----------
    grid = np.arange(81).reshape((9, 9))
    ones = (rng.random((9, 9)) < 0.3).astype(np.int64)
    cases = {
        "SlidingWindow_3": (ref_sw.SlidingWindow, (grid, 3), {}),
        "SlidingWindow_5": (ref_sw.SlidingWindow, (grid, 5), {}),
        "SlidingWindow_ones_3": (ref_sw.SlidingWindow, (ones, 3), {"iter_over_ones": True}),
        "SlidingIgnoreBorder_3": (ref_sw.SlidingIgnoreBorder, (grid, 3), {}),
        "CircularWindow_5": (ref_sw.CircularWindow, (grid, 5), {}),
        "InnerWindow_5_3": (ref_sw.InnerWindow, (grid, 5, 3), {}),
        "NoCenterWindow_3": (ref_sw.NoCenterWindow, (grid, 3), {}),
        "IgnoreBorderInnerSliding_5_3": (ref_sw.IgnoreBorderInnerSliding, (grid, 5),
                                         {"inner_size": 3}),
    }
    arrays = {"grid": grid, "ones": ones}
    for name, (cls, args, kw) in cases.items():
        wins, idx = [], []
        for w, c in cls(*args, **kw):
            wins.append(w)
            idx.append(c)
        arrays[name + "_windows"] = np.stack(wins)
        arrays[name + "_centres"] = np.array(idx)
        arrays[name + "_getitem"] = cls(*args, **kw)[4, 4]
    save("sliding.npz", **arrays)

    # --- rasters held by the reference's own test suite ---------------------
    zf = zipfile.ZipFile(os.path.join(REF, "tests/resources/tests_expected.zip"))
    fc = tif(zf, "expected/fourier_corrected_expected.tif")
    sp = tif(zf, "expected/srtm_processed.tif")
    final = tif(zf, "expected/final_dem.tif")
    # Derived known answer for ONE GrovesCorrection pass: the
    # groves class raster is a missing blob, but fourier_corrected -> (groves
    # x3) -> srtm_processed differ only where the correction fired, so
    # M = |delta| > 1e-3 is the effective mask.  Crop keeps the file small.
    d = np.abs(fc.astype(np.float64) - sp.astype(np.float64)) > 1e-3
    ys, xs = np.nonzero(d)
    print("groves-changed cells in the pair:", d.sum())
    # densest 192x192 crop
    best, by, bx = -1, 0, 0
    for y0 in range(0, fc.shape[0] - 192 + 1, 16):
        for x0 in range(0, fc.shape[1] - 192 + 1, 16):
            c = int(d[y0:y0 + 192, x0:x0 + 192].sum())
            if c > best:
                best, by, bx = c, y0, x0
    sl = (slice(by, by + 192), slice(bx, bx + 192))
    print("crop", by, bx, "changed cells", best)
    save("ref_rasters.npz", fourier_corrected=fc[sl], srtm_processed=sp[sl],
         crop=np.array([by, bx]), final_dem=final)


if __name__ == "__main__":
    main()
