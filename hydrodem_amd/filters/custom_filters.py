"""
Stencil operators of the hot path, backed by the HIP library.

Reference twins (`cguerrero/hydrodem/filters/custom_filters.py`):
``QuadraticFilter`` :202-257, ``MaskTallGroves`` :513-534,
``GrovesCorrection`` :664-732, ``GrovesCorrectionsIter`` :735-767,
``PostProcessingFinal`` :1104-1125 -- same class names, constructor
signatures, mutable operand attributes and error classes.

Fourier destripe branch: ``ExpandFilter`` :76-125,
``IsolatedPoints`` :320-366, ``BlanksFourier`` :369-429,
``DetectBlanksFourier`` :432-462, ``MaskFourier`` :537-561, ``FourierInitial``
:834-877, ``FourierProcessQuarters`` :880-1050, ``DetectApplyFourier``
:1053-1101.

Lagoon branch: ``MajorityFilter`` :22-73, ``CorrectNANValues``
:260-317, ``MaskNegatives`` / ``MaskPositives`` :465-510, ``TidyingLagoons``
:564-610, ``LagoonsDetection`` :613-661.

River branch (host side only -- ``RouteRivers`` is a serial scan by
construction, SURVEY section 2 #9 / 8e "not shardable"; no kernel is wanted):
``RouteRivers`` :128-199, ``ProcessRivers`` :770-798, ``ClipLagoonsRivers``
:801-831.  They exist so that `image_hsheds.py:6-7,203-205` imports and runs
unchanged when ``filters`` resolves here.

New operators (the reference has neither): ``SinkFill`` and
``D8FlowDirection``, shaped like every other ``Filter``.

Module namespace.  The reference's ``custom_filters`` is also where its callers
pick up the element-wise and SciPy wrappers (`image_srtm.py:7-8` takes
``BinaryClosing`` from here, `hydro_dem_process.py:20-21` ``AdditionFilter``),
because `custom_filters.py:9-19` imports them at module level.  The imports
below bind the same complete set of names.

Storage type.  The device path stores rasters as float32 (what GDAL hands the
reference, `image_srtm.py:125`).  The reference drifts to float64 after the
first groves pass (float32 * int64); the values agree to <= 1 float32 ulp and
every reference stencil re-reads its input through ``astype('float32')``
anyway (`sliding_window.py:132`).
"""

import copy
from collections import Counter  # noqa: F401  (name of the reference module's namespace)

import numpy as np

from . import Filter, ComposedFilter, ComposedFilterResults
from .simple_filters import (LowerThan, BooleanToInteger, GreaterThan,  # noqa: F401
                             ProductFilter, SubtractionFilter, AdditionFilter)
from .extension_filters import (BitwiseXOR, BinaryErosion, Around,  # noqa: F401
                                BinaryClosing, GreyDilation, Convolve,
                                FourierITransform, FourierTransform,
                                FourierShift, FourierIShift, AbsoluteValues)
from ..sliding_window import (SlidingWindow, CircularWindow,  # noqa: F401
                              NoCenterWindow, IgnoreBorderInnerSliding)
from .. import backend


class QuadraticFilter(Filter):  # pylint: disable=too-few-public-methods
    """Least-squares quadratic smoothing over a ``window_size`` square window;
    the ring of ``window_size // 2`` cells is returned unchanged
    (custom_filters.py:202-257).  Window validation raises the same
    ``WindowSizeHighError`` / ``WindowSizeEvenError`` as the reference's
    ``SlidingWindow`` constructor (sliding_window.py:150-156)."""

    auto_device = True      # device form == host form for a float32 raster

    def __init__(self, *, window_size):
        self.window_size = window_size

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        return backend.quadratic(image_to_filter, self.window_size)

    def apply_device(self, raster):
        return backend.quadratic_dev(raster, self.window_size)


class MaskTallGroves(ComposedFilter):  # pylint: disable=too-few-public-methods
    """``(image > 1.5) * 1`` (custom_filters.py:513-534).  Host NumPy; inside
    ``GrovesCorrection`` this algebra runs in the fused kernel's epilogue."""

    def __init__(self):  # pylint: disable=super-init-not-called
        self.filters = [GreaterThan(value=1.5), BooleanToInteger()]


class GrovesCorrection(ComposedFilter):  # pylint: disable=too-few-public-methods
    """One groves correction pass (custom_filters.py:664-732):

        smooth = QuadraticFilter(15)(img); hl = img - smooth
        m = groves_class * (hl > 1.5);     out = hl * (1 - m) + smooth

    evaluated by ONE fused HIP kernel when ``groves_class`` is a 0 / 1 mask (what
    `image_srtm.py:177-178` passes: a ``binary_closing`` result); a class raster with
    other values takes the reference's algebra literally -- ``m = class * tall`` --
    member by member, with only the quadratic filter on the GPU.  ``filters`` keeps
    the reference's five
    members so that callers can still re-bind their operands
    (``filters[3].factor`` is the groves class, ``filters[0].window_size`` the
    window, ``filters[2].filters[0].value`` the tall-grove threshold); they
    are read at ``apply`` time.  ``partial_results`` is filled only with
    ``keep_partial_results=True`` (it costs an extra quadratic pass)."""

    @property
    def auto_device(self):
        """Device form == host form for a float32 raster -- with a 0 / 1 class raster."""
        return self._is_mask(self._params()[0])

    def __init__(self, groves_class, keep_partial_results=False):  # pylint: disable=super-init-not-called
        self.partial_results = []
        self.keep_partial_results = keep_partial_results
        self.filters = [QuadraticFilter(window_size=15), SubtractionFilter(),
                        MaskTallGroves(), ProductFilter(factor=groves_class),
                        SubtractionFilter(minuend=1)]

    def _params(self):
        return (self.filters[3].factor, self.filters[0].window_size,
                self.filters[2].filters[0].value)

    @staticmethod
    def _is_mask(groves_class):
        """Only 0 and 1 in the class raster (then the fused kernel's `class != 0` is the
        reference's product with the class)."""
        g = np.asarray(groves_class)
        if g.dtype == bool or g.size == 0:
            return True
        if g.dtype.kind in "ui":                       # one or two SIMD passes, no temporaries
            return bool(g.max() <= 1 and (g.dtype.kind == "u" or g.min() >= 0))
        return not np.any((g != 0) & (g != 1))

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        groves_class, window, thr = self._params()
        if not self._is_mask(groves_class):
            # custom_filters.py:724-732 as written: highlight * (1 - class * tall) + smooth
            smooth = backend.quadratic(image_to_filter, window)
            highlight = image_to_filter - smooth
            product = groves_class * ((highlight > thr) * 1)
            if self.keep_partial_results:
                self.partial_results = [smooth, highlight, (highlight > thr) * 1, product,
                                        1 - product]
            return highlight * (1 - product) + smooth
        if self.keep_partial_results:
            img = np.ascontiguousarray(image_to_filter, dtype=np.float32)
            smooth = backend.quadratic(img, window)
            highlight = img - smooth
            tall = (highlight > thr) * 1
            product = groves_class * tall
            self.partial_results = [smooth, highlight, tall, product, 1 - product]
        return backend.groves(image_to_filter, groves_class, window, thr, 1)

    def apply_device(self, raster):
        groves_class, window, thr = self._params()
        if not self._is_mask(groves_class):
            raise NotImplementedError("the fused groves kernel takes a 0 / 1 class raster")
        with backend.DeviceRaster.from_host(backend.mask_bytes(groves_class),
                                            dtype=np.uint8, ctx=raster.ctx) as g:
            out = backend.groves_dev(raster, g, window, thr, 1)
            raster.ctx.synchronize()
        return out


class GrovesCorrectionsIter(ComposedFilter):  # pylint: disable=too-few-public-methods
    """``iterations`` chained ``GrovesCorrection`` passes
    (custom_filters.py:735-767).  When the members are untouched the whole
    chain is one C call that ping-pongs two device buffers."""

    auto_device = True      # device form == host form for a float32 raster

    def __init__(self, groves_class, iterations=3):  # pylint: disable=super-init-not-called
        self.filters = []
        for _ in range(iterations):
            self.filters.append(GrovesCorrection(groves_class))

    def _uniform(self):
        if not self.filters or not all(type(f) is GrovesCorrection and
                                       not f.keep_partial_results
                                       for f in self.filters):
            return None
        p0 = self.filters[0]._params()
        for f in self.filters[1:]:
            p = f._params()
            if p[0] is not p0[0] or p[1:] != p0[1:]:
                return None
        if not GrovesCorrection._is_mask(p0[0]):
            return None                     # member by member: the reference's algebra
        return p0

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        params = self._uniform()
        if params is None:
            return ComposedFilter.apply(self, image_to_filter)
        groves_class, window, thr = params
        return backend.groves(image_to_filter, groves_class, window, thr,
                              len(self.filters))

    def apply_device(self, raster):
        params = self._uniform()
        if params is None:
            return ComposedFilter.apply_device(self, raster)
        groves_class, window, thr = params
        with backend.DeviceRaster.from_host(backend.mask_bytes(groves_class),
                                            dtype=np.uint8, ctx=raster.ctx) as g:
            out = backend.groves_dev(raster, g, window, thr, len(self.filters))
            raster.ctx.synchronize()
        return out


class PostProcessingFinal(ComposedFilter):  # pylint: disable=too-few-public-methods
    """3x3 box mean then round to 1 m (custom_filters.py:1104-1125).  With the
    default members the two run as one fused kernel (float32 or float64)."""

    auto_device = True      # device form == host form for a float32 raster

    def __init__(self):  # pylint: disable=super-init-not-called
        self.filters = [Convolve(), Around()]

    def _fused(self):
        return (len(self.filters) == 2 and type(self.filters[0]) is Convolve
                and self.filters[0]._is_box3() and type(self.filters[1]) is Around)

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        if self._fused():
            return backend.boxmean3(image_to_filter, do_round=True)
        content = image_to_filter
        for filter_ in self.filters:
            content = filter_.apply(content)
        return content

    def apply_device(self, raster):
        if not self._fused():
            raise NotImplementedError("device chain needs the default members")
        return backend.boxmean3_dev(raster, do_round=True)


class SinkFill(Filter):  # pylint: disable=too-few-public-methods
    """Depression filling (new operator; normative definition SURVEY 8a A1).

    ``W = max(Z, spill elevation)``: the greatest fixed point of
    ``W[c] = max(Z[c], min(W[c], min8(W[n] + epsilon)))`` with the one-cell
    border pinned to Z ("interior only, border untouched", the convention of
    `sliding_window.py:187-192`); nodata (NaN) cells stay NaN and act as
    outlets.  ``epsilon = 0`` gives flats and is bit-reproducible.

    Attributes
    ----------
    epsilon : float
        Planchon-Darboux gradient added per step (metres), default 0.
    max_rounds : int
        Worklist-round limit; 0 = library default.  ``NotConvergedError`` when
        it is hit.
    stats : dict
        rounds / tile_visits / tiles of the last ``apply``.
    """

    auto_device = True      # device form == host form for a float32 raster

    def __init__(self, *, epsilon=0.0, max_rounds=0):
        self.epsilon = epsilon
        self.max_rounds = max_rounds
        self.stats = {}

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        out, self.stats = backend.sinkfill(image_to_filter, self.epsilon,
                                           self.max_rounds, return_stats=True)
        return out

    def apply_device(self, raster):
        out, self.stats = backend.sinkfill_dev(raster, self.epsilon, self.max_rounds)
        return out


class D8FlowDirection(Filter):  # pylint: disable=too-few-public-methods
    """D8 steepest-descent direction (new operator): ESRI codes
    E=1, SE=2, S=4, SW=8, W=16, NW=32, N=64, NE=128, 0 = no lower neighbour or
    border cell; drop = (z_c - z_k) / distance evaluated in float32 as
    ``(z_c - z_k) * w_k``, w = 1 or float32(0.70710678); ties go to the first
    neighbour in window order NW, N, NE, W, E, SW, S, SE (the order
    ``np.nonzero`` gives `custom_filters.py:193-195`).  Returns uint8."""

    auto_device = True      # device form == host form for a float32 raster

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return backend.d8(image_to_filter)

    def apply_device(self, raster):
        return backend.d8_dev(raster)


class HydroConditioning(ComposedFilter):  # pylint: disable=too-few-public-methods
    """``SinkFill`` then ``D8FlowDirection`` as one device-resident chain (the
    pair BASELINE.json's metric is quoted on).  ``filled`` keeps the filled
    DEM of the last ``apply``; the return value is the D8 grid."""

    def __init__(self, *, epsilon=0.0):
        super().__init__()
        self.filters = [SinkFill(epsilon=epsilon), D8FlowDirection()]
        self.filled = None

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        fill = self.filters[0]
        with backend.DeviceRaster.from_host(image_to_filter, dtype=np.float32) as z:
            # one call: the certifying pass of the fill writes the flow directions
            filled, codes, fill.stats = backend.sinkfill_d8_dev(
                z, eps=fill.epsilon, max_rounds=fill.max_rounds)
            with filled, codes:
                self.filled = filled.to_host()
                return codes.to_host()

    def apply_batch(self, rasters):
        """Fill + D8 of several rasters in ONE pass of the solver: ``[(filled, codes), ...]``.

        Small rasters (SRTM / HydroSHEDS tiles are 1201^2 ... 6000^2 cells) leave most of the
        GPU idle -- a fill is a chain of dependent tile visits -- so those of one width are
        stacked into one canvas with a nodata row between neighbours.  That changes no result: the cells next to nodata are pinned exactly as
        a raster's ring is (the operator's own rule), so every raster fills as it does alone;
        only the D8 codes of each raster's ring are put back to 0 afterwards (on the canvas
        they have neighbours).  The arrays returned are views of the downloaded canvases (one
        pair of canvases per width).  Bit-equal to ``apply`` raster by raster
        (``tests/test_gpu_parity.py::test_batch_of_rasters_fills_like_each_alone``)."""
        rasters = [np.asarray(r, dtype=np.float32) for r in rasters]
        for r in rasters:
            Filter.apply(self, r)
            if r.ndim != 2:
                raise ValueError("apply_batch takes 2-D rasters")
        if not rasters:
            return []
        rasters = [np.ascontiguousarray(r) for r in rasters]
        ctx = backend.context()
        lib, hnd = ctx.lib, ctx.handle
        fill = self.filters[0]
        out = [None] * len(rasters)
        # one canvas per width (tiles of a survey share theirs): a raster then is one
        # contiguous block of its canvas and goes up and comes down in one plain copy each
        # (pitched copies from pageable memory run row by row: 30 x slower)
        by_width = {}
        for k, r in enumerate(rasters):
            by_width.setdefault(r.shape[1], []).append(k)
        stats = None
        for width, members in by_width.items():
            rows = sum(rasters[k].shape[0] for k in members) + len(members) - 1
            with backend.DeviceRaster.empty((rows, width), np.float32, ctx) as z:
                ctx.check(lib.hdem_memset_dev(hnd, z.ptr, 0xff, z.nbytes))      # all NaN
                tops, y = [], 0
                for k in members:
                    r = rasters[k]
                    ctx.check(lib.hdem_memcpy_h2d(hnd, z.ptr + y * width * 4, r.ctypes.data, r.nbytes))
                    tops.append(y)
                    y += r.shape[0] + 1
                filled, codes, st = backend.sinkfill_d8_dev(z, eps=fill.epsilon,
                                                            max_rounds=fill.max_rounds)
                stats = st if stats is None else {
                    key: (min(stats[key], st[key]) if key == "converged" else stats[key] + st[key])
                    if isinstance(st[key], int) else st[key] for key in st}
                with filled, codes:
                    # the canvas comes down in two copies (page-locked blocks of the library,
                    # backend.host_empty); what is handed out are views of it, one per raster
                    # -- many small copies into fresh pageable arrays were measured at 200 ms
                    # for 16 tiles, the driver pinning and unpinning their pages
                    w_all, d_all = filled.to_host(), codes.to_host()
                for k, y0 in zip(members, tops):
                    h = rasters[k].shape[0]
                    d = d_all[y0:y0 + h]
                    d[0], d[-1], d[:, 0], d[:, -1] = 0, 0, 0, 0
                    out[k] = (w_all[y0:y0 + h], d)
        fill.stats = stats or {}
        return out


# ---------------------------------------------------------------------------
# Fourier destripe
# ---------------------------------------------------------------------------
class ExpandFilter(Filter):  # pylint: disable=too-few-public-methods
    """1 at the centres whose ``window_size`` square minus its four corners holds a
    cell > 0, 0 elsewhere -- including the ring of ``window_size // 2`` cells where
    the window does not fit (custom_filters.py:76-125)."""

    def __init__(self, *, window_size):
        self.window_size = window_size

    def apply(self, image_to_filter):
        return backend.expand(image_to_filter, self.window_size, np.float64)


class IsolatedPoints(Filter):  # pylint: disable=too-few-public-methods
    """Cells equal to 1 with no other cell > 0 in their window become 0; like the
    reference (custom_filters.py:320-366) this writes into its input and returns
    it.  Only centres whose window fits are looked at."""

    def __init__(self, *, window_size):
        self.window_size = window_size

    def apply(self, image_to_filter):
        img = image_to_filter
        g = np.asarray(img, dtype=np.float32)                # the window snapshot
        ones = np.trunc(g) == 1
        code = np.where(ones, 1, np.where(g > 0, 2, 0)).astype(np.uint8)
        res = backend.isolated_points(code, self.window_size)
        r = self.window_size // 2
        inner = np.zeros(g.shape, dtype=bool)
        inner[r:g.shape[0] - r, r:g.shape[1] - r] = True
        sel = ones & inner
        img[sel] = res[sel].astype(img.dtype)
        return img


class BlanksFourier(Filter):  # pylint: disable=too-few-public-methods
    """Peaks of a spectrum magnitude: cells above 4x the mean of their 55 x 55
    neighbourhood, its inner 5 x 5 and everything past the array edge left out.
    Returns (mask, image with the peaks zeroed) (custom_filters.py:369-429).
    Any odd window from 7 to 201; 55, the one the reference's pipeline uses, has a
    kernel instance of its own."""

    def __init__(self, *, window_size):
        self.window_size = window_size

    def apply(self, image_to_filter):
        return backend.blanks_fourier(image_to_filter, self.window_size)


class DetectBlanksFourier(Filter):  # pylint: disable=too-few-public-methods
    """Two ``BlanksFourier`` passes, the second on the image without the first
    pass's peaks; the masks are added (custom_filters.py:432-462)."""

    def apply(self, quarter_fourier):
        total = np.zeros(np.shape(quarter_fourier))
        blanks = BlanksFourier(window_size=55)
        for _ in (0, 1):
            found, quarter_fourier = blanks.apply(quarter_fourier)
            total += found
        return total


class MaskFourier(ComposedFilter):  # pylint: disable=too-few-public-methods
    """detect -> drop isolated points -> expand (custom_filters.py:537-561)."""

    def __init__(self):  # pylint: disable=super-init-not-called
        self.filters = [DetectBlanksFourier(), IsolatedPoints(window_size=3),
                        ExpandFilter(window_size=13)]


class FourierInitial(ComposedFilterResults):  # pylint: disable=too-few-public-methods
    """fft2 -> fftshift -> abs; keeps the shifted spectrum
    (custom_filters.py:834-877)."""

    def __init__(self):
        super().__init__()
        self.filters = [FourierTransform(), FourierShift(), AbsoluteValues()]
        self.fourier_shift = None

    def apply(self, image_to_filter):
        result = super().apply(image_to_filter)
        self.fourier_shift = self.results["FourierShift"]
        return result


class FourierProcessQuarters(Filter):  # pylint: disable=too-few-public-methods
    """Mask of the frequencies to drop, from the magnitude of the shifted
    spectrum: ``MaskFourier`` on the two upper quadrants (10-cell margin to the
    axes), point-mirrored into the lower half (custom_filters.py:880-1050).
    ``apply`` ignores its argument, as the reference does."""

    def __init__(self, fft_transform_abs):
        self.fft_transform_abs = fft_transform_abs
        self._ny, self._nx = fft_transform_abs.shape
        self._mid_y, self._y_odd = divmod(self._ny, 2)
        self._mid_x, self._x_odd = divmod(self._nx, 2)
        self.pair_mid = self._mid_y, self._mid_x
        self._margin = 10

    def apply(self, image_to_filter):  # pylint: disable=unused-argument
        ny, nx, my, mx, m = self._ny, self._nx, self._mid_y, self._mid_x, self._margin
        mag = self.fft_transform_abs
        first = MaskFourier().apply(np.array(mag[:my - m, :mx - m]))
        second = MaskFourier().apply(np.array(mag[:my - m, mx + m + self._x_odd:nx]))
        q1 = np.zeros(self.pair_mid)
        q2 = np.zeros(self.pair_mid)
        q1[:my - m, :mx - m] = first
        q2[:my - m, m:mx] = second
        full = np.zeros((ny, nx))
        full[:my, :mx] = q1
        full[:my, mx + self._x_odd:] = q2
        full[my + self._y_odd:, :mx] = q2[::-1, ::-1]
        full[my + self._y_odd:, mx + self._x_odd:] = q1[::-1, ::-1]
        return full


class DetectApplyFourier(ComposedFilter):  # pylint: disable=too-few-public-methods
    """Destripe: find the bright isolated frequencies of the spectrum, zero them,
    transform back (custom_filters.py:1053-1101).  One device-resident pass
    (``hdem_fourier_destripe_f32``): the shifts are index maps, the quadrant masks
    are applied straight to the spectrum.  Returns float32 (the reference: float64
    from a complex128 inverse; values agree to ~1e-5 m).  With
    ``keep_mask=True`` the full mask is kept in ``.mask`` afterwards."""

    auto_device = True      # device form == host form for a float32 raster

    def __init__(self, keep_mask=False):  # pylint: disable=super-init-not-called
        self.initial = FourierInitial()
        self.fft_transform_abs = None
        self.keep_mask = keep_mask
        self.mask = None
        self.filters = []

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        if self.keep_mask:
            out, mask = backend.fourier_destripe(image_to_filter, return_mask=True)
            self.mask = mask.astype(np.float64)
            return out
        return backend.fourier_destripe(image_to_filter)

    def apply_device(self, raster):
        return backend.fourier_destripe_dev(raster)


# ---------------------------------------------------------------------------
# HydroSHEDS / lagoon branch
# ---------------------------------------------------------------------------
class MajorityFilter(Filter):  # pylint: disable=too-few-public-methods
    """The value that fills more than 70 % of ``window_size**2 - 1`` cells of the
    circular window (square minus corners, the centre counts), else 0; the ring
    where the window does not fit stays 0 (custom_filters.py:22-73)."""

    def __init__(self, *, window_size):
        self.window_size = window_size

    def apply(self, image_to_filter):
        img = backend.DeviceRaster.from_host(
            np.ascontiguousarray(image_to_filter, dtype=np.float32))
        return backend.widened_to_host(backend.majority_dev(img, self.window_size), np.float64)

    def apply_device(self, raster):
        return backend.majority_dev(raster, self.window_size)


class CorrectNANValues(Filter):  # pylint: disable=too-few-public-methods
    """Voids (cells < 0) become the mean of their non-negative neighbours; like the
    reference (custom_filters.py:260-317) this writes into its input and returns
    it.  Odd windows 3 to 11 (the reference's pipeline uses 3)."""

    def __init__(self, *, window_size=3):
        self.window_size = window_size

    def apply(self, image_to_filter):
        dem = image_to_filter
        g = np.ascontiguousarray(dem, dtype=np.float32)
        fixed = backend.correct_nan_dev(backend.DeviceRaster.from_host(g),
                                        window_size=self.window_size)
        if g is dem:
            # (every cell that is not repaired comes back bit for bit)
            fixed.ctx.check(fixed.ctx.lib.hdem_memcpy_d2h(fixed.ctx.handle, g.ctypes.data,
                                                          fixed.ptr, g.nbytes))
            return dem
        r = int(self.window_size) // 2
        sel = g < 0
        sel[:r] = sel[g.shape[0] - r:] = False
        sel[:, :r] = sel[:, g.shape[1] - r:] = False
        np.copyto(dem, fixed.to_host(), where=sel, casting="unsafe")
        return dem

    def apply_device(self, raster):
        return backend.correct_nan_dev(raster, window_size=self.window_size)


class MaskNegatives(ComposedFilter):  # pylint: disable=too-few-public-methods
    """1 where the image is negative (custom_filters.py:465-487)."""

    def __init__(self):  # pylint: disable=super-init-not-called
        self.filters = [LowerThan(value=0.0), BooleanToInteger()]


class MaskPositives(ComposedFilter):  # pylint: disable=too-few-public-methods
    """1 where the image is positive (custom_filters.py:490-510)."""

    def __init__(self):  # pylint: disable=super-init-not-called
        self.filters = [GreaterThan(value=0.0), BooleanToInteger()]


class TidyingLagoons(ComposedFilter):  # pylint: disable=too-few-public-methods
    """Erode twice, expand by 7, multiply with the input, 7 x 7 grey dilation
    (custom_filters.py:564-610).  ``apply`` runs the four steps in one
    device-resident call when the list is the stock one; ``filters`` stays
    inspectable and patchable like the reference's."""

    def __init__(self):  # pylint: disable=super-init-not-called
        self.filters = [BinaryErosion(iterations=2), ExpandFilter(window_size=7),
                        ProductFilter(), GreyDilation(size=(7, 7))]

    def _stock(self):
        f = self.filters
        return (len(f) == 4 and isinstance(f[0], BinaryErosion) and f[0].iterations == 2 and
                isinstance(f[1], ExpandFilter) and f[1].window_size == 7 and
                isinstance(f[2], ProductFilter) and isinstance(f[3], GreyDilation) and
                tuple(np.atleast_1d(f[3].size)) in ((7, 7), (7,)))

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        self.filters[2].factor = content = image_to_filter
        if self._stock():
            img = backend.DeviceRaster.from_host(
                np.ascontiguousarray(image_to_filter, dtype=np.float32))
            return backend.widened_to_host(backend.tidying_lagoons_dev(img), np.float64)
        for filter_ in self.filters:
            content = filter_.apply(content)
        return content

    def apply_device(self, raster):
        return backend.tidying_lagoons_dev(raster)


class LagoonsDetection(ComposedFilterResults):  # pylint: disable=too-few-public-methods
    """NaN repair -> majority (11) -> tidying -> mask of positives, keeping the
    intermediate results the orchestration reads (custom_filters.py:613-661).
    One device-resident call (``hdem_lagoons_detection_f32_dev``); like the
    reference the void repair is also written into the input array."""

    def __init__(self):
        super().__init__()
        self.filters = [CorrectNANValues(), MajorityFilter(window_size=11),
                        TidyingLagoons(), MaskPositives()]
        self.hsheds_nan_fixed = None
        self.mask_lagoons = None
        self.lagoons_values = None

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        g = np.ascontiguousarray(image_to_filter, dtype=np.float32)
        mask, fixed, values = backend.lagoons_detection_dev(backend.DeviceRaster.from_host(g))
        # CorrectNANValues works in place.  The kernel passes every other cell through bit for
        # bit, so a float32 raster simply receives the repaired one; any other type only its
        # repaired cells (its other values need not be float32 numbers).
        if g is image_to_filter:
            fixed.ctx.check(fixed.ctx.lib.hdem_memcpy_d2h(fixed.ctx.handle, g.ctypes.data,
                                                          fixed.ptr, g.nbytes))
        else:
            sel = g < 0
            sel[0] = sel[-1] = False
            sel[:, 0] = sel[:, -1] = False
            np.copyto(image_to_filter, fixed.to_host(), where=sel, casting="unsafe")
        self.hsheds_nan_fixed = image_to_filter
        self.lagoons_values = backend.widened_to_host(values, np.float64)
        self.mask_lagoons = backend.widened_to_host(mask, np.int64)
        self.results = {"CorrectNANValues": self.hsheds_nan_fixed,
                        "TidyingLagoons": self.lagoons_values,
                        "MaskPositives": self.mask_lagoons}
        return self.mask_lagoons


# ---------------------------------------------------------------------------
# River branch (host side; SURVEY section 2 #9: serial by construction)
# ---------------------------------------------------------------------------
class RouteRivers(Filter):  # pylint: disable=too-few-public-methods
    """Route a river mask downhill on a reference DEM
    (custom_filters.py:128-199).

    Every cell of the mask whose value truncates to 1 and whose
    ``window_size`` window fits is visited in raster order; all cells of its
    DEM window that hold the window minimum become river and are then raised
    to 10000 in the working copy of the DEM, so a later window sees the
    earlier ones' marks -- the one Gauss-Seidel stencil of the reference, and
    the reason it stays on the host.  The DEM is deep-copied at construction
    (`:163`) and read as float32 (`sliding_window.py:132`); a window holding a
    NaN marks nothing (``window == nan`` is never true).  Returns float64."""

    def __init__(self, *, window_size, dem):
        self.window_size = window_size
        self.dem = copy.deepcopy(dem)

    def apply(self, image_to_filter):
        mask = SlidingWindow(image_to_filter, window_size=self.window_size)
        work = SlidingWindow(self.dem, window_size=self.window_size).grid
        reach = self.window_size // 2
        routed = np.zeros(self.dem.shape)
        height, width = mask.grid.shape
        inner = mask.grid[reach:height - reach, reach:width - reach]
        rows, cols = np.nonzero(np.trunc(inner) == 1)           # raster order
        for row, col in zip(rows.tolist(), cols.tolist()):
            window = work[row:row + 2 * reach + 1, col:col + 2 * reach + 1]
            lowest = window == np.amin(window)
            routed[row:row + 2 * reach + 1, col:col + 2 * reach + 1][lowest] = 1
            window[lowest] = 10000
        return routed


class ProcessRivers(ComposedFilter):  # pylint: disable=too-few-public-methods
    """Mask of positives -> expand by 3 -> route on the HydroSHEDS DEM ->
    binary closing (custom_filters.py:770-798)."""

    def __init__(self, hsheds):  # pylint: disable=super-init-not-called
        self.filters = [MaskPositives(), ExpandFilter(window_size=3),
                        RouteRivers(window_size=3, dem=hsheds), BinaryClosing()]


class ClipLagoonsRivers(ComposedFilter):  # pylint: disable=too-few-public-methods
    """Rivers minus their intersection with the lagoons: ``rivers XOR
    (mask_lagoons * rivers)`` (custom_filters.py:801-831)."""

    def __init__(self, mask_lagoons, rivers_routed_closing):  # pylint: disable=super-init-not-called
        self.filters = [ProductFilter(factor=mask_lagoons),
                        BitwiseXOR(operand=rivers_routed_closing)]
