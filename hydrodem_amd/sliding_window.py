"""
Host mirror of the reference's stencil-window protocol
(`cguerrero/hydrodem/sliding_window.py:12-793`).

This is the definition of the border / dtype semantics every GPU stencil of
the path reproduces:

* windows are float32 snapshots of the grid taken at construction
  (`sliding_window.py:128-132`) -- all reference stencils are Jacobi;
* centres run in raster order over the cells where the full window fits,
  ``range(ws // 2, n - ws // 2)`` on both axes (`:187-192`): the ring of
  ``ws // 2`` cells is never visited ("interior only, border untouched");
* masked positions of a window are NaN and consumers use NaN-aware reductions;
* ``iter_over_ones`` visits only centres whose value truncates to 1 (`:193`).

It is host logic for small grids, golden tests and callers that still iterate
windows in Python; the hot path does not go through it.  Written from the
behaviour above, with two deliberate differences from the reference: the NaN
mask is built once (the reference re-appends its index list on every
``__iter__`` / ``__getitem__``, `:185,250`), and ``__getitem__`` refuses a
centre one past the last valid one, which the reference's off-by-one lets
through with a truncated window (`:268-271`).
"""

from itertools import product

import numpy as np

from .exceptions import (WindowSizeHighError, WindowSizeEvenError,
                         CenterCloseBorderError, NumpyArrayExpectedError,
                         InnerSizeError)


class SlidingWindow:
    """Square sliding window over a 2-D grid: ``for window, (j, i) in
    SlidingWindow(grid, window_size)`` and ``sliding[j, i]``."""

    def __init__(self, grid, window_size, iter_over_ones=False):
        self.grid = grid
        self.window_size = window_size
        self._indices_nan = []
        self._mask_ready = False
        self.iter_over_ones = iter_over_ones

    # -- validated attributes ------------------------------------------------
    @property
    def grid(self):
        return self._grid

    @grid.setter
    def grid(self, value):
        if not isinstance(value, np.ndarray):
            raise NumpyArrayExpectedError(value)
        self._grid = value.astype('float32')

    @property
    def window_size(self):
        return self._window_size

    @window_size.setter
    def window_size(self, value):
        if any(value > n for n in self.grid.shape):
            raise WindowSizeHighError(value, self.grid.shape)
        if value % 2 != 1:
            raise WindowSizeEvenError(value)
        self._window_size = int(value)

    # -- masking hook (cooperative: subclasses extend and call super) ---------
    def _masked_positions(self):
        """(row, col) window positions to blank with NaN; none here."""
        return []

    def _prepare_mask(self):
        if not self._mask_ready:
            self._indices_nan = list(dict.fromkeys(self._masked_positions()))
            self._mask_ready = True

    def _snapshot(self, j, i):
        half = self.window_size // 2
        window = self.grid[j - half:j + half + 1, i - half:i + half + 1].copy()
        for pos in self._indices_nan:
            window[pos] = np.nan
        return window

    # -- protocol ------------------------------------------------------------
    def __iter__(self):
        self._prepare_mask()
        ny, nx = self.grid.shape
        half = self.window_size // 2
        for j, i in product(range(half, ny - half), range(half, nx - half)):
            if self.iter_over_ones and int(self.grid[j, i]) != 1:
                continue
            yield self._snapshot(j, i), (j, i)

    def __getitem__(self, coords):
        self._prepare_mask()
        j, i = coords
        ny, nx = self.grid.shape
        half = self.window_size // 2
        if not (half <= j < ny - half and half <= i < nx - half):
            raise CenterCloseBorderError(coords, window_size=self.window_size)
        return self._snapshot(j, i)


class SlidingIgnoreBorder(SlidingWindow):
    """Same iteration over a copy of the grid padded with ``ws // 2`` NaN cells
    on every side, so every original cell gets a window
    (`sliding_window.py:303-418`).  Yielded indices are in padded
    coordinates, like the reference."""

    def __init__(self, grid, window_size, *args, **kwargs):
        super().__init__(grid, window_size, *args, **kwargs)
        half = self.window_size // 2
        self.grid = np.pad(self.grid, half, mode='constant',
                           constant_values=np.nan)


class CircularWindow(SlidingWindow):
    """The four corners of every window are NaN (`sliding_window.py:421-499`)."""

    def _masked_positions(self):
        last = self.window_size - 1
        return [(0, 0), (0, last), (last, 0), (last, last)] + \
            list(super()._masked_positions())


class InnerWindow(SlidingWindow):
    """An ``inner_size`` square around the centre is NaN, the centre itself
    excepted (`sliding_window.py:502-653`)."""

    def __init__(self, grid, window_size, inner_size, *args, **kwargs):
        self.inner_size = inner_size
        super().__init__(grid, window_size, *args, **kwargs)

    def _masked_positions(self):
        if self.inner_size > self.window_size:
            raise InnerSizeError(self.inner_size, self.window_size)
        centre, reach = self.window_size // 2, self.inner_size // 2
        span = range(centre - reach, centre + reach + 1)
        inner = [(r, c) for r, c in product(span, span)
                 if not r == c == centre]
        return inner + list(super()._masked_positions())


class NoCenterWindow(SlidingWindow):
    """The centre of every window is NaN (`sliding_window.py:656-736`)."""

    def _masked_positions(self):
        centre = self.window_size // 2
        return [(centre, centre)] + list(super()._masked_positions())


class IgnoreBorderInnerSliding(SlidingIgnoreBorder, InnerWindow,
                               NoCenterWindow):
    """NaN padding + hollow inner square + no centre, by cooperative
    inheritance (`sliding_window.py:739-793`); the window the Fourier blank
    detector averages (`custom_filters.py:417-421`)."""
