"""Package-style half of the drop-in: ``cguerrero.hydrodem.filters...``,
``cguerrero.hydrodem.sliding_window`` and ``cguerrero.hydrodem.exceptions`` (the
reference's tests import these, `tests/test_filter.py:10`,
`tests/test_sliding_window.py:4-8`) resolve to ``hydrodem_amd``; every other
``cguerrero.*`` module (``utils_dem``, ``config_loader`` ...) keeps coming from
the reference tree further down ``sys.path``."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
