"""
Regenerates tests/golden/assembly.npz.  RUNS ONLY IN THE BUILD CONTAINER (see
make_golden.py): the final assembly of the reference's orchestration --
`hydro_dem_process.py:80-88` (``_prepare_final_terms``: AdditionFilter, SubtractionFilter,
two ProductFilters) and `:147-149` (the three-term sum, PostProcessingFinal) -- evaluated
with the imported reference's filter classes on seeded inputs of the types the pipeline
holds at that point: SRTM float64 (after the groves passes: float32 * int64), HydroSHEDS
float32, lagoon values float64, the two masks int64.  `hydro_dem_process.py` itself cannot be
imported here (GDAL), so the five calls are spelled out in its order.

    python tests/golden/make_golden_assembly.py
"""
import os
import sys

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))

from filters.custom_filters import (SubtractionFilter, ProductFilter,  # noqa: E402
                                    AdditionFilter, PostProcessingFinal,
                                    MaskPositives, MaskNegatives, MaskTallGroves)


def main():
    rng = np.random.default_rng(20240901)
    shape = (97, 131)
    y, x = np.mgrid[0:shape[0], 0:shape[1]]
    base = 100 + 0.05 * x + 3 * np.sin(y / 9.0)
    srtm = base + rng.normal(0, 0.4, shape)                        # float64, not float32 values
    hsheds = (base - 1.5 + rng.normal(0, 0.3, shape)).astype(np.float32)
    mask_lagoons = np.zeros(shape, dtype=np.int64)
    mask_lagoons[20:40, 30:70] = 1
    lagoons_values = mask_lagoons * np.round(base - 2.0)           # float64, 0 outside
    rivers = (rng.random(shape) < 0.04).astype(np.int64)
    rivers[60, :] = 1
    rivers[mask_lagoons == 1] = 0                                   # ClipLagoonsRivers left them apart
    # hydro_dem_process.py:80-88
    mask_rivers_lagoons = AdditionFilter(addend=mask_lagoons).apply(rivers)
    not_rivers_lagoons = SubtractionFilter(minuend=1).apply(mask_rivers_lagoons)
    first_term = ProductFilter(factor=srtm).apply(not_rivers_lagoons)
    third_term = ProductFilter(factor=hsheds).apply(rivers)
    # :147-149
    dem_complete = first_term + lagoons_values + third_term
    final_dem = PostProcessingFinal().apply(dem_complete)
    assert final_dem.dtype == np.float64 and first_term.dtype == np.float64
    out = dict(srtm=srtm, hsheds=hsheds, mask_lagoons=mask_lagoons.astype(np.uint8),
               lagoons_values=lagoons_values, rivers=rivers.astype(np.uint8),
               not_rivers_lagoons=not_rivers_lagoons.astype(np.uint8), first_term=first_term,
               third_term=third_term, dem_complete=dem_complete, final_dem=final_dem)
    # the mask chains on a float32 raster (custom_filters.py:465-534)
    probe = (srtm - base).astype(np.float32) * 4
    out.update(probe=probe, positives=MaskPositives().apply(probe).astype(np.uint8),
               negatives=MaskNegatives().apply(probe).astype(np.uint8),
               tall=MaskTallGroves().apply(probe).astype(np.uint8))
    path = os.path.join(HERE, "assembly.npz")
    np.savez_compressed(path, **out)
    print("assembly.npz", os.path.getsize(path) // 1024, "KiB; river cells", int(rivers.sum()),
          "final range", final_dem.min(), final_dem.max())


if __name__ == "__main__":
    main()
