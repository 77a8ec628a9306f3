"""Flat-import shim: ``from sliding_window import ...`` -> hydrodem_amd.sliding_window."""
from hydrodem_amd.sliding_window import (SlidingWindow, SlidingIgnoreBorder,  # noqa: F401
                                         CircularWindow, InnerWindow, NoCenterWindow,
                                         IgnoreBorderInnerSliding)
