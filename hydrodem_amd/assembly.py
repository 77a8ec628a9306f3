"""
The final assembly of the reference's orchestration, device resident.

`hydro_dem_process.py:60-91` (``_prepare_final_terms``) and `:147-149` combine the three
branches with the element-wise filters and smooth the sum:

    mask_rivers_lagoons = AdditionFilter(addend=lagoons.mask_lagoons).apply(rivers)
    not_rivers_lagoons  = SubtractionFilter(minuend=1).apply(mask_rivers_lagoons)
    first_term          = ProductFilter(factor=srtm).apply(not_rivers_lagoons)
    third_term          = ProductFilter(factor=lagoons.hsheds_nan_fixed).apply(rivers)
    dem_complete        = first_term + lagoons.lagoons_values + third_term
    final_dem           = PostProcessingFinal().apply(dem_complete)

Called like that -- unchanged, through the drop-in package -- every ``apply`` is a host
array in, a host array out.  :func:`final_dem` is the same sequence of the same filter
objects on device rasters (``apply_device``): each input goes up once, the eight
operators run in HBM, the result comes down once.  The arithmetic is the reference's
(float64 where the pipeline holds float64), so the result is bit for bit the reference's
(`tests/golden/assembly.npz`).
"""

import numpy as np

from . import backend
from .filters.custom_filters import PostProcessingFinal
from .filters.simple_filters import AdditionFilter, ProductFilter, SubtractionFilter


def _mask(a):
    a = np.asarray(a)
    return a.astype(np.uint8) if a.dtype != np.uint8 else a


def _elevations(a):
    a = np.asarray(a)
    return a if a.dtype in (np.float32, np.float64) else a.astype(np.float64)


def final_dem(srtm, mask_lagoons, hsheds_nan_fixed, lagoons_values, rivers, ctx=None,
              keep_terms=False):
    """``final_dem`` of `hydro_dem_process.py:147-149` from the results of the three
    branches: ``srtm`` (`image_srtm.py:199`), ``lagoons.mask_lagoons`` /
    ``.hsheds_nan_fixed`` / ``.lagoons_values`` (`custom_filters.py:656-660`) and ``rivers``
    (`image_hsheds.py:203-205`), host arrays.  Returns the host array the reference
    returns (the elevations' float type); with ``keep_terms`` also the three terms."""
    up = lambda a: backend.DeviceRaster.from_host(a, ctx=ctx)      # noqa: E731
    rasters = []

    def own(r):
        rasters.append(r)
        return r

    try:
        d_srtm, d_hs = own(up(_elevations(srtm))), own(up(_elevations(hsheds_nan_fixed)))
        d_values = own(up(_elevations(lagoons_values)))
        d_lagoons, d_rivers = own(up(_mask(mask_lagoons))), own(up(_mask(rivers)))
        both = own(AdditionFilter(addend=d_lagoons).apply_device(d_rivers))
        neither = own(SubtractionFilter(minuend=1).apply_device(both))
        first = own(ProductFilter(factor=d_srtm).apply_device(neither))
        third = own(ProductFilter(factor=d_hs).apply_device(d_rivers))
        partial = own(AdditionFilter(addend=first).apply_device(d_values))
        complete = own(AdditionFilter(addend=partial).apply_device(third))
        result = own(PostProcessingFinal().apply_device(complete))
        out = result.to_host()
        if keep_terms:
            return out, (first.to_host(), d_values.to_host(), third.to_host())
        return out
    finally:
        for r in rasters:
            r.free()
