"""
Element-wise operators of the path (host side, NumPy).

Same names, constructor signatures and operand mutability as
`cguerrero/hydrodem/filters/simple_filters.py:7-275`.  On the groves path
these are not launched one by one: the fused HIP kernel behind
``GrovesCorrection`` evaluates the whole
subtract / threshold / mask / complement / blend algebra in its epilogue.
They stay here as thin NumPy expressions because the orchestration above the
seam (`hydro_dem_process.py:60-91`) composes them directly.

Every operator is one NumPy ufunc between the image and one operand held in an
attribute that callers may re-bind between calls (``filters[2].factor = ...``,
custom_filters.py:607,725); the classes below only say which ufunc, which
attribute, and on which side the image goes.

``apply(ndarray)`` is NumPy on the host, with NumPy's result types -- what the
reference returns.  ``apply_device(raster)`` is the same operator on a
:class:`hydrodem_amd.backend.DeviceRaster` (``hdem_elementwise_dev``), so that the
orchestration's algebra (`hydro_dem_process.py:60-91`) can stay in HBM between the
stencils; there masks are uint8 where NumPy holds bool / int64.  The
operand may be a scalar, a host array (uploaded for the call) or a device raster.
"""

import numpy as np

from . import Filter
from .. import backend


class _Elementwise(Filter):  # pylint: disable=too-few-public-methods
    ufunc = None            # numpy ufunc of two arguments
    operand = None          # name of the attribute holding the other argument
    image_first = True      # ufunc(image, operand) or ufunc(operand, image)
    type_checked = True     # Filter.apply raises NumpyArrayExpectedError for non-arrays

    def apply(self, image_to_filter):
        if self.type_checked:
            Filter.apply(self, image_to_filter)
        other = getattr(self, self.operand)
        pair = (image_to_filter, other) if self.image_first else (other, image_to_filter)
        return type(self).ufunc(*pair)

    device_op = None        # backend.EW_* code of the operator

    def apply_device(self, raster):
        """The operator on a device raster; the caller keeps ``raster`` and owns the
        result."""
        other = getattr(self, self.operand)
        if isinstance(other, backend.DeviceRaster) or np.ndim(other) == 0:
            return backend.elementwise_dev(self.device_op, raster, other)
        host = np.asarray(other)
        if host.dtype == bool or (host.dtype.kind in "iu" and host.size and
                                  0 <= host.min() and host.max() <= 255):
            host = host.astype(np.uint8)                   # masks travel as bytes
        elif host.dtype != np.float64:
            host = host.astype(np.float32)
        with backend.DeviceRaster.from_host(host, ctx=raster.ctx) as operand:
            out = backend.elementwise_dev(self.device_op, raster, operand)
            raster.ctx.synchronize()                       # the operand is freed on exit
        return out


class LowerThan(_Elementwise):
    """``image < value`` -> bool grid (simple_filters.py:7-50)."""
    ufunc, operand, device_op = np.less, "value", backend.EW_LT

    def __init__(self, *, value):
        self.value = value


class GreaterThan(_Elementwise):
    """``image > value`` -> bool grid (simple_filters.py:53-96)."""
    ufunc, operand, device_op = np.greater, "value", backend.EW_GT

    def __init__(self, *, value):
        self.value = value


class BooleanToInteger(_Elementwise):
    """bool -> integer grid by ``* 1`` (simple_filters.py:99-131)."""
    ufunc, operand, device_op = np.multiply, "_one", backend.EW_MUL     # (``* 1``)
    _one = 1


class ProductFilter(_Elementwise):
    """``factor * image``; ``factor`` is a scalar or a grid and may be
    re-bound after construction (simple_filters.py:134-180;
    custom_filters.py:607)."""
    ufunc, operand, image_first, device_op = np.multiply, "factor", False, backend.EW_MUL

    def __init__(self, factor=1):
        self.factor = factor


class AdditionFilter(_Elementwise):
    """``addend + image`` (simple_filters.py:183-229)."""
    ufunc, operand, image_first, device_op = np.add, "addend", False, backend.EW_ADD

    def __init__(self, addend=0):
        self.addend = addend


class SubtractionFilter(_Elementwise):
    """``minuend - image``.  Like the reference it does not type-check its
    operand (simple_filters.py:232-275), and ``minuend`` is re-bound by
    ``GrovesCorrection`` (custom_filters.py:725)."""
    ufunc, operand, image_first, type_checked = np.subtract, "minuend", False, False
    device_op = backend.EW_RSUB

    def __init__(self, *, minuend=0.0):
        self.minuend = minuend
