"""
The two SciPy/NumPy wrapper filters that sit on the named path, backed by HIP
kernels: ``Convolve`` and ``Around``
(`cguerrero/hydrodem/filters/extension_filters.py:133-184,98-130`).

The other wrappers of that module (morphology, XOR, FFT) belong to the
lagoon/river and Fourier branches, which are outside this build's scope
(SURVEY section 8f); they are not re-declared here.
"""

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
    float64); other odd weights up to 15 x 15 take the general kernel, which
    computes in float32 storage.
    """

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
