"""
Operator protocol of the hot path: ``Filter`` / ``ComposedFilter`` /
``ComposedFilterResults``.

Same class names, constructor shapes and ``apply(ndarray) -> ndarray``
contract as the reference (`cguerrero/hydrodem/filters/__init__.py:10-127`),
so orchestration code written against the reference keeps working when
``filters`` resolves to this package (see ``hydrodem_amd/dropin``).

One addition: every GPU-backed filter also implements
``apply_device(raster)`` on a :class:`hydrodem_amd.backend.DeviceRaster`, and
the two composed classes chain through it, so a chain pays one host->device and
one device->host copy instead of one pair per member.

``apply(ndarray)`` keeps the reference's contract -- result types and in-place side
effects included.  It only takes the device chain by itself when every member says
(``auto_device``) that its device form returns exactly what its host form returns
for a float32 raster; the element-wise operators (NumPy result types: bool, int64,
float64) and the filters that write into their input (``CorrectNANValues``,
``IsolatedPoints``) do not, and run member by member as in the reference.
``apply_device`` is the explicit route: everything stays in HBM, masks are uint8,
elevations keep the type they were uploaded with.
"""

from abc import ABC, abstractmethod

from numpy import ndarray

from ..exceptions import NumpyArrayExpectedError


class Filter(ABC):  # pylint: disable=too-few-public-methods
    """Base operator.  Subclasses call ``super().apply(x)`` for the type
    check (filters/__init__.py:23-39)."""

    @abstractmethod
    def apply(self, image_to_filter):
        if not isinstance(image_to_filter, ndarray):
            raise NumpyArrayExpectedError(image_to_filter)


def _device_chain(filters):
    """True when every member can run device-resident and returns there what its
    host form returns (see the module docstring)."""
    return bool(filters) and all(
        callable(getattr(f, "apply_device", None)) and getattr(f, "auto_device", False)
        for f in filters)


class LazyResults(dict):
    """``results`` of a device-resident :class:`ComposedFilterResults` chain: stages stay
    in HBM (``.device[name]``) and are copied to the host the first time somebody reads
    them by class name, as callers of the reference do (custom_filters.py:658-660,876)."""

    def __init__(self):
        super().__init__()
        self.device = {}

    def __missing__(self, key):
        if key not in self.device:
            raise KeyError(key)
        self[key] = value = self.device[key].to_host()
        return value

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.device

    def keys(self):
        return list(dict.fromkeys(list(dict.keys(self)) + list(self.device)))

    # every other read goes through keys() / __getitem__ as well, so that a stage that is
    # still only on the device is there for get(), iteration, items(), values() and len()
    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def downloaded(self):
        """Names of the stages that have a host copy already."""
        return list(dict.keys(self))

    def release(self, keep=None):
        """Free the device copies (host copies already made stay).  A raster stored under
        two names -- a member whose ``apply_device`` hands its input on -- is freed once;
        ``keep``: a raster the caller goes on using (the last stage ``apply_device``
        returned) is left alone."""
        seen = set()
        for raster in self.device.values():
            if raster is keep or id(raster) in seen:
                continue
            seen.add(id(raster))
            raster.free()
        self.device = {}


class ComposedFilter(Filter):  # pylint: disable=too-few-public-methods
    """Left-to-right chain over ``self.filters``
    (filters/__init__.py:42-80)."""

    def __init__(self):
        self.filters = []

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        if (_device_chain(self.filters) and image_to_filter.ndim == 2
                and image_to_filter.dtype == "float32"):
            from ..backend import DeviceRaster
            with DeviceRaster.from_host(image_to_filter) as raster:
                with self.apply_device(raster) as result:
                    return result.to_host()
        stage = image_to_filter
        for member in self.filters:
            stage = member.apply(stage)
        return stage

    def apply_device(self, raster):
        """Chain on a device-resident raster; the caller keeps ownership of
        ``raster`` and receives a new one."""
        stage = raster
        for member in self.filters:
            following = member.apply_device(stage)
            if stage is not raster and stage is not following:
                stage.free()        # intermediate of this chain
            stage = following
        return stage


class ComposedFilterResults(Filter):  # pylint: disable=too-few-public-methods
    """Chain that also keeps every stage in ``results[ClassName]``
    (filters/__init__.py:83-127); callers read stages by class name
    (custom_filters.py:658-660,876)."""

    def __init__(self):
        self.filters = []
        self.results = {}

    def apply(self, image_to_filter):
        Filter.apply(self, image_to_filter)
        stage = image_to_filter
        for member in self.filters:
            stage = member.apply(stage)
            self.results[type(member).__name__] = stage
        return stage

    def apply_device(self, raster):
        """The chain on a device-resident raster.  Every stage stays in HBM;
        ``results[ClassName]`` downloads a stage when it is first read
        (:class:`LazyResults`; ``results.release()`` frees the device copies).  The
        caller keeps ``raster``; the returned raster is the last stage, owned by
        ``results``."""
        lazy = LazyResults()
        stage = raster
        for member in self.filters:
            stage = member.apply_device(stage)
            lazy.device[type(member).__name__] = stage
        self.results = lazy
        return stage
