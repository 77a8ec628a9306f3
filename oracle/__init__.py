"""
CPU oracle package.  TEST INFRASTRUCTURE ONLY -- see the headers of
``hdem_oracle_np.py`` and ``hdem_oracle.c``.  Importable from ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg; never
from ``hydrodem_amd``.
"""
from .hdem_oracle_np import *  # noqa: F401,F403
from . import c_oracle  # noqa: F401
