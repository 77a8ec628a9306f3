"""
Element-wise operators of the path (host side, NumPy).

Same names, constructor signatures and operand mutability as
`cguerrero/hydrodem/filters/simple_filters.py:7-275`.  On the groves path
these are not launched one by one: the fused HIP kernel behind
``GrovesCorrection`` evaluates the whole
subtract / threshold / mask / complement / blend algebra in its epilogue.
They stay here as thin NumPy expressions because the orchestration above the
seam (`hydro_dem_process.py:60-91`) composes them directly.
"""

from . import Filter


class LowerThan(Filter):  # pylint: disable=too-few-public-methods
    """``image < value`` -> bool grid (simple_filters.py:7-50)."""

    def __init__(self, *, value):
        self.value = value

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return image_to_filter < self.value


class GreaterThan(Filter):  # pylint: disable=too-few-public-methods
    """``image > value`` -> bool grid (simple_filters.py:53-96)."""

    def __init__(self, *, value):
        self.value = value

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return image_to_filter > self.value


class BooleanToInteger(Filter):  # pylint: disable=too-few-public-methods
    """bool -> integer grid by ``* 1`` (simple_filters.py:99-131)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return image_to_filter * 1


class ProductFilter(Filter):  # pylint: disable=too-few-public-methods
    """``factor * image``; ``factor`` is a scalar or a grid and may be
    re-bound after construction (simple_filters.py:134-180;
    custom_filters.py:607)."""

    def __init__(self, factor=1):
        self.factor = factor

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return self.factor * image_to_filter


class AdditionFilter(Filter):  # pylint: disable=too-few-public-methods
    """``addend + image`` (simple_filters.py:183-229)."""

    def __init__(self, addend=0):
        self.addend = addend

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return self.addend + image_to_filter


class SubtractionFilter(Filter):  # pylint: disable=too-few-public-methods
    """``minuend - image``.  Like the reference it does not type-check its
    operand (simple_filters.py:232-275), and ``minuend`` is re-bound by
    ``GrovesCorrection`` (custom_filters.py:725)."""

    def __init__(self, *, minuend=0.0):
        self.minuend = minuend

    def apply(self, subtracting):  # pylint: disable=arguments-differ
        return self.minuend - subtracting
