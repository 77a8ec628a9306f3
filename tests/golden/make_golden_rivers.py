"""
Regenerates tests/golden/rivers.npz.  RUNS ONLY IN THE BUILD CONTAINER (see
make_golden.py): imports the reference's river filters
(custom_filters.py:128-199,770-831) and stores seeded inputs with their outputs.

Inputs: (1) a seeded DEM with integer plateaus (ties: several window minima at
once) and random-walk river masks, windows 3 and 5; (2) two rasters of the
reference's own suite (tests/resources/tests_expected.zip, read with Pillow) used
as *inputs*: ``hsheds_nan_values_expected`` as the DEM and ``rivers_processed``
(a real rasterised river network, 0/1) as the river raster, ``lagoons_expected``
for the lagoon mask.  The reference's own input/expected pair for RouteRivers is
not available (inputs zip is a missing blob) and rivers_routed_expected /
rivers_processed do not close with each other (431 cells; different dates).

    python tests/golden/make_golden_rivers.py
"""
import io
import os
import sys
import warnings
import zipfile

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))

from filters.custom_filters import (RouteRivers, ProcessRivers,  # noqa: E402
                                    ClipLagoonsRivers, MaskPositives)


def tif(zf, name):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(zf.read(name))))


def walks(rng, shape, count, steps):
    m = np.zeros(shape, dtype=np.float32)
    for _ in range(count):
        j, i = rng.integers(0, shape[0]), rng.integers(0, shape[1])
        for _ in range(steps):
            m[j, i] = 1
            j = int(np.clip(j + rng.integers(-1, 2), 0, shape[0] - 1))
            i = int(np.clip(i + rng.integers(0, 2), 0, shape[1] - 1))
    return m


def stages(rivers, hsheds, lagoons_mask):
    chain = ProcessRivers(hsheds)
    x = rivers
    got = []
    for f in chain.filters:
        x = f.apply(x)
        got.append(np.array(x))
    clipped = ClipLagoonsRivers(lagoons_mask, got[3]).apply(got[3])
    return got, clipped


def main():
    warnings.simplefilter("ignore")
    out = {}
    rng = np.random.default_rng(4242)
    y, x = np.mgrid[0:70, 0:90]
    dem = (120 - 0.3 * x + 4 * np.sin(y / 7.0) + rng.normal(0, 0.6, (70, 90))).astype(np.float32)
    dem[20:40, 30:60] = np.round(dem[20:40, 30:60])       # plateaus: tied minima
    dem[5, 5] = np.nan                                     # a window with a NaN marks nothing
    mask = walks(rng, dem.shape, 6, 80)
    mask[0, :] = 1                                         # ring cells are never centres
    mask[10, 10] = 2                                       # int(2) != 1: skipped
    mask[11, 11] = 1.7                                     # truncates to 1: visited
    out.update(dem=dem, mask=mask,
               routed3=RouteRivers(window_size=3, dem=dem).apply(mask),
               routed5=RouteRivers(window_size=5, dem=dem).apply(mask))
    lag = (rng.random(dem.shape) < 0.2).astype(np.int64)
    rivers = mask * rng.integers(1, 4, dem.shape)          # rasterised attribute values
    got, clipped = stages(rivers, np.nan_to_num(dem, nan=100.0), lag)
    out.update(seed_rivers=rivers, seed_lagoons=lag, seed_positives=got[0], seed_expand=got[1],
               seed_routed=got[2], seed_closing=got[3], seed_clipped=clipped)
    print("seeded", dem.shape, "river cells", int(mask.sum()), "routed3", int(out["routed3"].sum()),
          "routed5", int(out["routed5"].sum()), "clipped", clipped.dtype, int(clipped.sum()))

    zf = zipfile.ZipFile(os.path.join(REF, "tests/resources/tests_expected.zip"))
    hs = tif(zf, "expected/hsheds_nan_values_expected.tif")
    rv = tif(zf, "expected/rivers_processed.tif")
    lagm = MaskPositives().apply(tif(zf, "expected/lagoons_expected.tif"))
    got, clipped = stages(rv, hs, lagm)
    out.update(ref_hsheds=hs, ref_rivers=rv.astype(np.uint8), ref_lagoons_mask=lagm.astype(np.uint8),
               ref_expand=got[1].astype(np.uint8), ref_routed=got[2].astype(np.uint8),
               ref_closing=got[3].astype(np.uint8), ref_clipped=clipped.astype(np.uint8))
    assert clipped.dtype == np.int64 and got[2].dtype == np.float64 and got[3].dtype == bool
    print("reference rasters", hs.shape, "river cells", int(rv.sum()), "routed", int(got[2].sum()),
          "closing", int(got[3].sum()), "clipped", int(clipped.sum()))
    path = os.path.join(HERE, "rivers.npz")
    np.savez_compressed(path, **out)
    print("rivers.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
