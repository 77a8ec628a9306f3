"""Flat-import shim: ``filters.simple_filters`` -> hydrodem_amd.filters.simple_filters."""
from hydrodem_amd.filters.simple_filters import *  # noqa: F401,F403
import hydrodem_amd.filters.simple_filters as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
