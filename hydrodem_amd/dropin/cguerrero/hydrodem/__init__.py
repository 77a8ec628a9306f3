"""See ``cguerrero/__init__.py``: this directory comes first in the package
path, the reference's ``cguerrero/hydrodem`` directory stays behind it."""
import os as _os
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
# filters/, sliding_window.py and exceptions.py are the flat shims one level up
__path__.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
