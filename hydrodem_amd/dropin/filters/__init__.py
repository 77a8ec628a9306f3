"""Flat-import shim: ``from filters import Filter`` -> hydrodem_amd.filters."""
from hydrodem_amd.filters import Filter, ComposedFilter, ComposedFilterResults  # noqa: F401
