"""Flat-import shim: ``filters.extension_filters`` -> hydrodem_amd.filters.extension_filters."""
from hydrodem_amd.filters.extension_filters import *  # noqa: F401,F403
import hydrodem_amd.filters.extension_filters as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
