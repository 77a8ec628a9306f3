"""Flat-import shim: ``from exceptions import ...`` -> hydrodem_amd.exceptions."""
from hydrodem_amd.exceptions import *  # noqa: F401,F403
from hydrodem_amd.exceptions import (HydroDEMException, WindowSizeHighError,  # noqa: F401
                                     WindowSizeEvenError, CenterCloseBorderError,
                                     NumpyArrayExpectedError, InnerSizeError,
                                     BackendError, NotConvergedError)
