"""
hydrodem_amd -- MI355X (gfx950) implementation of HydroDEM's raster hot path.

Layout
    csrc/            hand-written HIP kernels + the C ABI (include/hydrodem_hip.h)
    backend.py       ctypes loader, status -> exception mapping, DeviceRaster
    filters/         host mirror of the reference's ``filters`` package
    sliding_window   host mirror of the reference's stencil-window protocol
    exceptions       error classes of the seam
    partition        row-block decomposition + halo exchange (multi-GPU)
    dropin/          directory to put first on ``sys.path`` so that the
                     reference's flat imports (``from filters import ...``)
                     resolve here
"""

from .exceptions import (HydroDEMException, WindowSizeHighError,  # noqa: F401
                         WindowSizeEvenError, CenterCloseBorderError,
                         NumpyArrayExpectedError, InnerSizeError, BackendError,
                         NotConvergedError)
from .filters import Filter, ComposedFilter, ComposedFilterResults  # noqa: F401
from .filters.custom_filters import (QuadraticFilter, MaskTallGroves,  # noqa: F401
                                     GrovesCorrection, GrovesCorrectionsIter,
                                     PostProcessingFinal, SinkFill,
                                     D8FlowDirection, HydroConditioning,
                                     ExpandFilter, IsolatedPoints, BlanksFourier,
                                     DetectBlanksFourier, MaskFourier, FourierInitial,
                                     FourierProcessQuarters, DetectApplyFourier,
                                     MajorityFilter, CorrectNANValues, MaskNegatives,
                                     MaskPositives, TidyingLagoons, LagoonsDetection,
                                     RouteRivers, ProcessRivers, ClipLagoonsRivers)
from .filters.extension_filters import (Convolve, Around, AbsoluteValues,  # noqa: F401
                                        FourierTransform, FourierITransform,
                                        FourierShift, FourierIShift, BitwiseXOR,
                                        BinaryErosion, BinaryClosing, GreyDilation)
from .filters.simple_filters import (LowerThan, GreaterThan, BooleanToInteger,  # noqa: F401
                                     ProductFilter, AdditionFilter,
                                     SubtractionFilter)

__version__ = "0.1.0"
