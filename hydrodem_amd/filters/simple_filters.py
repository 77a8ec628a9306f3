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
"""

import numpy as np

from . import Filter


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


class LowerThan(_Elementwise):
    """``image < value`` -> bool grid (simple_filters.py:7-50)."""
    ufunc, operand = np.less, "value"

    def __init__(self, *, value):
        self.value = value


class GreaterThan(_Elementwise):
    """``image > value`` -> bool grid (simple_filters.py:53-96)."""
    ufunc, operand = np.greater, "value"

    def __init__(self, *, value):
        self.value = value


class BooleanToInteger(_Elementwise):
    """bool -> integer grid by ``* 1`` (simple_filters.py:99-131)."""
    ufunc, operand = np.multiply, "_one"
    _one = 1


class ProductFilter(_Elementwise):
    """``factor * image``; ``factor`` is a scalar or a grid and may be
    re-bound after construction (simple_filters.py:134-180;
    custom_filters.py:607)."""
    ufunc, operand, image_first = np.multiply, "factor", False

    def __init__(self, factor=1):
        self.factor = factor


class AdditionFilter(_Elementwise):
    """``addend + image`` (simple_filters.py:183-229)."""
    ufunc, operand, image_first = np.add, "addend", False

    def __init__(self, addend=0):
        self.addend = addend


class SubtractionFilter(_Elementwise):
    """``minuend - image``.  Like the reference it does not type-check its
    operand (simple_filters.py:232-275), and ``minuend`` is re-bound by
    ``GrovesCorrection`` (custom_filters.py:725)."""
    ufunc, operand, image_first, type_checked = np.subtract, "minuend", False, False

    def __init__(self, *, minuend=0.0):
        self.minuend = minuend
