"""
The SciPy/NumPy wrapper filters of the built paths, backed by the HIP library:
``Convolve`` and ``Around`` (`cguerrero/hydrodem/filters/extension_filters.py:
133-184,98-130`) and, for the Fourier destripe branch,
``AbsoluteValues`` :63-95, ``FourierTransform`` / ``FourierITransform`` /
``FourierShift`` / ``FourierIShift`` :348-480.

The transforms run on the GPU (rocFFT) in complex64 -- what scipy.fftpack
returns for the float32 rasters of the pipeline; for other input types the
reference would compute in double.  The shifts and the absolute value are data
movement / one NumPy expression on the host; inside ``DetectApplyFourier`` none
of them is executed as a separate step (index maps in the kernels).

Lagoon branch: ``BinaryErosion`` :187-235, ``BinaryClosing``
:238-293, ``GreyDilation`` :296-345 (scipy.ndimage semantics: cross structure by
default, border value 0, ``mode='reflect'`` for the grey dilation; centred odd
structures / sizes) and ``BitwiseXOR`` :12-60.
"""

import copy


import numpy as np

from . import Filter
from .. import backend


class Around(Filter):  # pylint: disable=too-few-public-methods
    """``np.around`` (round half to even) element-wise
    (extension_filters.py:98-130)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return backend.around(image_to_filter).reshape(image_to_filter.shape)


class Convolve(Filter):  # pylint: disable=too-few-public-methods
    """``scipy.ndimage.convolve(x, weights) / weights.size`` with SciPy's
    defaults: ``mode='reflect'``, origin 0, double accumulation, output in the
    input dtype (extension_filters.py:133-184).

    The default ``ones((3, 3))`` takes the LDS-tiled 3x3 kernel (float32 and
    float64); other odd weights up to 15 x 15 take the general kernel (float32
    rasters in float32, everything else in float64, as SciPy does).
    """

    auto_device = True      # device form == host form for a float32 raster

    def __init__(self, weights=np.ones((3, 3))):
        self.weights = weights

    def _is_box3(self):
        w = np.asarray(self.weights)
        return w.shape == (3, 3) and bool(np.all(w == 1))

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        if self._is_box3():
            return backend.boxmean3(image_to_filter, do_round=False)
        return backend.convolve(image_to_filter, np.asarray(self.weights))

    def apply_device(self, raster):
        if not self._is_box3():
            raise NotImplementedError(
                "device-resident Convolve supports the 3x3 ones() weights")
        return backend.boxmean3_dev(raster, do_round=False)


class AbsoluteValues(Filter):  # pylint: disable=too-few-public-methods
    """``np.absolute`` (extension_filters.py:63-95)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return np.absolute(image_to_filter)


class FourierTransform(Filter):  # pylint: disable=too-few-public-methods
    """2-D discrete Fourier transform (extension_filters.py:348-379)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return backend.fft2(image_to_filter)


class FourierITransform(Filter):  # pylint: disable=too-few-public-methods
    """Inverse 2-D transform, normalised by the size
    (extension_filters.py:382-414)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return backend.fft2(image_to_filter, inverse=True)


class FourierShift(Filter):  # pylint: disable=too-few-public-methods
    """Zero frequency to the centre (extension_filters.py:417-447)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return np.fft.fftshift(image_to_filter)


class FourierIShift(Filter):  # pylint: disable=too-few-public-methods
    """Inverse of :class:`FourierShift` (extension_filters.py:450-480)."""

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return np.fft.ifftshift(image_to_filter)


class BitwiseXOR(Filter):  # pylint: disable=too-few-public-methods
    """``np.bitwise_xor(operand, image)``; the operand is deep-copied at
    construction (extension_filters.py:12-60)."""

    def __init__(self, *, operand):
        self.operand = copy.deepcopy(operand)

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        return np.bitwise_xor(self.operand, image_to_filter)


def _mask_raster(image):
    return backend.DeviceRaster.from_host(backend.mask_bytes(image))


class BinaryErosion(Filter):  # pylint: disable=too-few-public-methods
    """``scipy.ndimage.binary_erosion(image, iterations=...)`` -> bool grid
    (extension_filters.py:187-235)."""

    def __init__(self, *, iterations):
        self.iterations = iterations

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        out = backend.binary_erosion_dev(_mask_raster(image_to_filter), self.iterations)
        return out.to_host().view(np.bool_)                   # (the kernels write 0 / 1)


class BinaryClosing(Filter):  # pylint: disable=too-few-public-methods
    """``scipy.ndimage.binary_closing(image, structure=...)`` -> bool grid
    (extension_filters.py:238-293)."""

    def __init__(self, *, structure=None):
        self.structure = structure

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        out = backend.binary_closing_dev(_mask_raster(image_to_filter), self.structure)
        return out.to_host().view(np.bool_)


class GreyDilation(Filter):  # pylint: disable=too-few-public-methods
    """``scipy.ndimage.grey_dilation(image, size=...)``: flat maximum filter
    (extension_filters.py:296-345).  float32 and float64 rasters are dilated in their own
    type (a maximum: exact either way); other types go through float64, which holds every
    integer the reference's callers pass, and come back in the input's dtype."""

    def __init__(self, *, size):
        self.size = size

    def apply(self, image_to_filter):
        super().apply(image_to_filter)
        src = np.asarray(image_to_filter)
        work = np.float32 if src.dtype == np.float32 else np.float64
        img = backend.DeviceRaster.from_host(np.ascontiguousarray(src, dtype=work))
        out = backend.grey_dilation_dev(img, self.size).to_host()
        img.free()
        return out.astype(src.dtype, copy=False)
