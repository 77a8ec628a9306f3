"""
Error types of the operator boundary.

Mirrors the error convention of the reference package
(`cguerrero/hydrodem/exceptions.py:6-73`): one base class whose ``str()`` is the
message, and one subclass per validation failure.  The message texts are part
of the contract: the reference's own tests assert them
(`cguerrero/tests/test_sliding_window.py:92-93,110-111,130-133`).

Two additions that the reference does not have, both raised by the HIP
backend only:

* :class:`BackendError` -- the HIP shared library is missing or a HIP/RCCL
  call failed.  The product path never falls back to a CPU implementation, so
  this is loud by design.
* :class:`NotConvergedError` -- sink-fill hit ``max_rounds`` before reaching
  the fixed point.
"""


class HydroDEMException(Exception):
    """Base class; ``str(exc)`` is the message (exceptions.py:6-22)."""

    def __init__(self, msg=""):
        self._msg = msg
        super().__init__(msg)

    def __str__(self):
        return self._msg


class WindowSizeHighError(HydroDEMException):
    """Window larger than the grid (exceptions.py:25-32)."""

    def __init__(self, window_size, grid_dimensions=""):
        super().__init__(
            msg=f'Window size: {window_size} cannot be higher than grid '
                f'dimensions: {grid_dimensions}')


class WindowSizeEvenError(HydroDEMException):
    """Even window size (exceptions.py:35-42)."""

    def __init__(self, window_size):
        super().__init__(
            msg=f'Window size: {window_size} cannot be an even number')


class CenterCloseBorderError(HydroDEMException):
    """Requested window centre does not leave room for a full window
    (exceptions.py:45-53)."""

    def __init__(self, center_window, window_size):
        super().__init__(
            msg=f'Center of window: {center_window} too close of border. '
                f'Window size: {window_size}')


class NumpyArrayExpectedError(HydroDEMException):
    """Operand is not a ``numpy.ndarray`` (exceptions.py:56-63)."""

    def __init__(self, provided):
        super().__init__(
            msg=f'Expected numpy ndarray type. Provided: {type(provided)}')


class InnerSizeError(HydroDEMException):
    """Inner window not smaller than the outer one.

    The reference declares this with a one-argument constructor but raises it
    with two (exceptions.py:66-73 vs sliding_window.py:646), which would be a
    ``TypeError`` there; both arities are accepted here.
    """

    def __init__(self, inner_size, window_size=None):
        if window_size is None:
            super().__init__(
                msg=f'Expected numpy ndarray type. Provided: '
                    f'{type(inner_size)}')
        else:
            super().__init__(
                msg=f'Inner size: {inner_size} must be odd and lower than '
                    f'window size: {window_size}')


class BackendError(HydroDEMException):
    """HIP backend unavailable or a device call failed (no reference twin)."""


class NotConvergedError(HydroDEMException):
    """Sink-fill stopped at its round limit before the fixed point."""
