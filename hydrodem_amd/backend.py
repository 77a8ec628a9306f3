"""
ctypes binding of ``libhydrodem_hip.so`` (C ABI: ``include/hydrodem_hip.h``).

This is the whole host side of the boundary: load the library, map status
codes to the reference's exception classes, and move NumPy arrays across.
There is deliberately no CPU implementation behind it -- when the library or
a GPU is missing every operator raises :class:`BackendError`.
"""

import ctypes
import importlib.util
import os
import sys
import threading
import weakref

import numpy as np

from .exceptions import (BackendError, HydroDEMException, NotConvergedError,
                         WindowSizeEvenError, WindowSizeHighError)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhydrodem_hip.so")

# status codes (include/hydrodem_hip.h)
OK, BAD_ARG, WINDOW_EVEN, WINDOW_HIGH, HIP_ERR, NOT_CONVERGED, NO_DEVICE, OOM = range(8)

# kernel ids for the timing query
K_D8, K_FILL_INIT, K_FILL_TILE, K_BOXMEAN, K_GROVES, K_CONVOLVE, K_COPY, K_FILL_ROUND = range(8)
K_BLOCKMAX, K_FFT, K_FOURIER_ROWSUM, K_FOURIER_DETECT, K_FOURIER_MASK, K_FOURIER_POINT = range(8, 14)
K_LAGOON, K_MAJORITY, K_FILL_COARSE, K_FILL_FLAT, K_ELEMENTWISE = 14, 15, 16, 17, 18
K_FILL_HUB = 19

# element-wise operators and raster types of hdem_elementwise_dev
EW_MUL, EW_ADD, EW_RSUB, EW_GT, EW_LT, EW_NONZERO = range(6)
_EW_TYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2,
             np.dtype(np.int64): 3}

FILL_INIT, FILL_WARM, FILL_ACT_TOP, FILL_ACT_BOTTOM = 0, 1, 2, 4
FILL_GHOST_TOP, FILL_GHOST_BOTTOM, FILL_SYNC_ONLY, FILL_NO_VERIFY = 0x10, 0x20, 0x40, 0x80
FILL_RESUME, FILL_GHOST_GIVEN, FILL_NO_COARSE = 0x100, 0x200, 0x400
FILL_DEFER = 0x800


class KernelStat(ctypes.Structure):
    _fields_ = [("launches", ctypes.c_int64), ("ms", ctypes.c_double),
                ("units", ctypes.c_int64)]


class FillStats(ctypes.Structure):
    _fields_ = [("rounds", ctypes.c_int32), ("converged", ctypes.c_int32),
                ("tile_visits", ctypes.c_int64), ("tiles", ctypes.c_int64),
                ("tile_h", ctypes.c_int32), ("tile_w", ctypes.c_int32),
                ("visits_flat", ctypes.c_int32), ("async_timed_out", ctypes.c_int32),
                ("iterations", ctypes.c_int64), ("visits_unchanged", ctypes.c_int64),
                ("visits_requeued", ctypes.c_int64), ("round_visits", ctypes.c_int64),
                ("pending", ctypes.c_int64), ("partial_residency", ctypes.c_int32),
                ("flat_unchanged", ctypes.c_int32),
                ("deferred_visits", ctypes.c_int64), ("deferred_unchanged", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_c = ctypes
_vp, _i, _f = _c.c_void_p, _c.c_int, _c.c_float
# name -> argtypes; every function returns int except the three noted
SIGNATURES = {
    "hdem_device_count": [_c.POINTER(_i)],
    "hdem_init": [_i, _c.POINTER(_vp)],
    "hdem_shutdown": [_vp],
    "hdem_set_stream": [_vp, _vp],
    "hdem_synchronize": [_vp],
    "hdem_set_fill_slice_us": [_vp, _i],
    "hdem_set_fill_seam_words": [_vp, _vp],
    "hdem_fill_seam_apply_dev": [_vp, _vp, _i, _i, _vp, _vp, ctypes.c_int64, _vp],
    "hdem_set_fill_coarse_start": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_fill_hub_prepare_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_fill_hub_raster_dev": [_vp, _vp],
    "hdem_set_fill_hub_levels": [_vp, _vp],
    "hdem_malloc": [_vp, _c.c_size_t, _c.POINTER(_vp)],
    "hdem_free": [_vp, _vp],
    "hdem_trim": [_vp, _c.POINTER(_c.c_size_t)],
    "hdem_memcpy_h2d": [_vp, _vp, _vp, _c.c_size_t],
    "hdem_memcpy_d2h": [_vp, _vp, _vp, _c.c_size_t],
    "hdem_memcpy_d2d": [_vp, _vp, _vp, _c.c_size_t],
    "hdem_memset_dev": [_vp, _vp, _i, _c.c_size_t],
    "hdem_host_alloc": [_vp, _c.c_size_t, _c.POINTER(_vp)],
    "hdem_host_free": [_vp, _vp],
    "hdem_memcpy_h2d_async": [_vp, _vp, _vp, _c.c_size_t],
    "hdem_memcpy_d2h_async": [_vp, _vp, _vp, _c.c_size_t],
    "hdem_profile_enable": [_vp, _i],
    "hdem_profile_reset": [_vp],
    "hdem_profile_get": [_vp, _i, _c.POINTER(KernelStat)],
    "hdem_d8_f32": [_vp, _vp, _i, _i, _vp],
    "hdem_d8_f32_dev": [_vp, _vp, _i, _i, _vp],
    "hdem_sinkfill_f32": [_vp, _vp, _i, _i, _f, _i, _vp, _c.POINTER(FillStats)],
    "hdem_sinkfill_f32_dev": [_vp, _vp, _i, _i, _f, _i, _i, _vp,
                              _c.POINTER(FillStats)],
    "hdem_blockmax_f32_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_copy_rate_dev": [_vp, _vp, _vp, _c.c_size_t],
    "hdem_elementwise_dev": [_vp, _i, _vp, _i, _vp, _i, _c.c_double, _c.c_int64, _vp, _i],
    "hdem_fourier_destripe_f32": [_vp, _vp, _i, _i, _vp, _vp],
    "hdem_fourier_destripe_f32_dev": [_vp, _vp, _i, _i, _vp, _vp],
    "hdem_blanks_fourier_f32_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_isolated_points_u8_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_expand_u8_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_fft2_c2c_f32_dev": [_vp, _vp, _i, _i, _i],
    "hdem_fft2_c2c_f64_dev": [_vp, _vp, _i, _i, _i],
    "hdem_correct_nan_f32_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_majority_f32_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_binary_erosion_u8_dev": [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp],
    "hdem_binary_closing_u8_dev": [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp],
    "hdem_grey_dilation_f32_dev": [_vp, _vp, _i, _i, _i, _i, _vp],
    "hdem_grey_dilation_f64_dev": [_vp, _vp, _i, _i, _i, _i, _vp],
    "hdem_tidying_lagoons_f32_dev": [_vp, _vp, _i, _i, _vp],
    "hdem_lagoons_detection_f32_dev": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "hdem_sinkfill_d8_f32_dev": [_vp, _vp, _i, _i, _f, _i, _i, _vp, _vp,
                                 _c.POINTER(FillStats)],
    "hdem_boxmean3_f32": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_boxmean3_f64": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_boxmean3_f32_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_boxmean3_f64_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_convolve_f32": [_vp, _vp, _i, _i, _vp, _i, _i, _vp],
    "hdem_convolve_f64": [_vp, _vp, _i, _i, _vp, _i, _i, _vp],
    "hdem_around_f32": [_vp, _vp, _c.c_int64, _vp],
    "hdem_around_f64": [_vp, _vp, _c.c_int64, _vp],
    "hdem_quadratic_f32": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_quadratic_f32_dev": [_vp, _vp, _i, _i, _i, _vp],
    "hdem_groves_f32": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "hdem_groves_f32_dev": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
}
OTHER_SYMBOLS = {"hdem_last_error": _c.c_char_p, "hdem_version": _i}

_lib = None
_lock = threading.Lock()


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    ``libamdhip64.so`` (soname ``libamdhip64.so.7``, same as ``/opt/rocm``'s) and look
    it up by file name, so if this library pulled in the system runtime first, a later
    ``import torch`` would load a second runtime and report no usable GPU.  Loading
    torch's copy first (without importing torch) makes both bind to the same one.
    ``HYDRODEM_HIP_RUNTIME=system`` keeps the system runtime (processes that never
    import torch)."""
    if os.environ.get("HYDRODEM_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for base in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(base, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load_library(path=None):
    """dlopen the HIP library and declare every prototype.  Needs no GPU."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        _share_torch_hip_runtime()
        p = path or os.environ.get("HYDRODEM_HIP_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise BackendError(
                f"HIP library not found at {p}: build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` "
                f"(there is no CPU fallback)")
        try:
            lib = ctypes.CDLL(p)
        except OSError as exc:
            raise BackendError(f"cannot load {p}: {exc}") from exc
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        lib.hdem_last_error.argtypes = []
        lib.hdem_last_error.restype = ctypes.c_char_p
        lib.hdem_version.argtypes = []
        lib.hdem_version.restype = ctypes.c_int
        if path is None:
            _lib = lib
        return lib


def _raise(lib, rc, window=None, shape=None):
    msg = (lib.hdem_last_error() or b"").decode("utf-8", "replace")
    if rc == WINDOW_EVEN:
        raise WindowSizeEvenError(window)
    if rc == WINDOW_HIGH:
        raise WindowSizeHighError(window, shape)
    if rc == NOT_CONVERGED:
        raise NotConvergedError(msg)
    if rc == BAD_ARG:
        raise ValueError(msg)
    if rc == OOM:
        raise MemoryError(msg)
    raise BackendError(msg or f"hydrodem_hip status {rc}")


class Context:
    """One ``hdem_ctx`` (device, stream, workspace).  Use :func:`context`."""

    def __init__(self, device=0):
        self.lib = load_library()
        handle = ctypes.c_void_p()
        rc = self.lib.hdem_init(int(device), ctypes.byref(handle))
        if rc != OK:
            _raise(self.lib, rc)
        self.handle = handle
        self.device = int(device)

    def check(self, rc, **kw):
        if rc != OK:
            _raise(self.lib, rc, **kw)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hdem_shutdown(self.handle)
            self.handle = None

    def synchronize(self):
        self.check(self.lib.hdem_synchronize(self.handle))

    def set_fill_seam_words(self, dptr):
        """Three device ints for FILL_DEFER calls (``hdem_set_fill_seam_words``); 0: none."""
        self.check(self.lib.hdem_set_fill_seam_words(self.handle, ctypes.c_void_p(int(dptr) or None)))

    def fill_seam_apply(self, w_ptr, H, W, recv_top_ptr, recv_bot_ptr, pending, words_ptr):
        """``hdem_fill_seam_apply_dev``: received rows into the ghost rows, seam words set."""
        vp = lambda p: ctypes.c_void_p(int(p) or None)
        self.check(self.lib.hdem_fill_seam_apply_dev(self.handle, vp(w_ptr), int(H), int(W),
                                                     vp(recv_top_ptr), vp(recv_bot_ptr),
                                                     int(pending), vp(words_ptr)))

    def set_fill_slice_us(self, microseconds):
        """Time slice of the asynchronous sink-fill phase (0 = to convergence)."""
        self.check(self.lib.hdem_set_fill_slice_us(self.handle, int(microseconds)))

    def set_fill_coarse_start(self, coarse_ptr, ch, cw, block, row_map_ptr=None):
        """Device pointers; see ``hdem_set_fill_coarse_start``."""
        self.check(self.lib.hdem_set_fill_coarse_start(
            self.handle, ctypes.c_void_p(coarse_ptr or 0), int(ch), int(cw), int(block),
            ctypes.c_void_p(row_map_ptr or 0)))

    def fill_hub_prepare(self, z_ptr, h, w, flags, w_ptr):
        """``hdem_fill_hub_prepare_dev``: d of a row block into the interior of ``w``."""
        self.check(self.lib.hdem_fill_hub_prepare_dev(
            self.handle, ctypes.c_void_p(z_ptr), int(h), int(w), int(flags), ctypes.c_void_p(w_ptr)))

    def fill_hub_raster(self, out_ptr):
        """``hdem_fill_hub_raster_dev``: the prepared block's hub raster into ``out_ptr``."""
        self.check(self.lib.hdem_fill_hub_raster_dev(self.handle, ctypes.c_void_p(out_ptr)))

    def set_fill_hub_levels(self, levels_ptr):
        self.check(self.lib.hdem_set_fill_hub_levels(self.handle, ctypes.c_void_p(levels_ptr or 0)))

    def set_stream(self, stream_ptr):
        self.check(self.lib.hdem_set_stream(self.handle,
                                            ctypes.c_void_p(stream_ptr or 0)))

    def trim(self):
        """Hand the context's cached device blocks and scratch buffers back to the
        device (``hdem_trim``); returns the bytes released."""
        n = ctypes.c_size_t(0)
        self.check(self.lib.hdem_trim(self.handle, ctypes.byref(n)))
        return int(n.value)

    # -- timing ----------------------------------------------------------
    def profile(self, on=True):
        self.check(self.lib.hdem_profile_enable(self.handle, int(bool(on))))

    def profile_reset(self):
        self.check(self.lib.hdem_profile_reset(self.handle))

    def profile_get(self, kernel_id):
        st = KernelStat()
        self.check(self.lib.hdem_profile_get(self.handle, kernel_id,
                                             ctypes.byref(st)))
        return {"launches": st.launches, "ms": st.ms, "units": st.units}


_contexts = {}


def context(device=None):
    """Per-process context of ``device`` (default: ``HYDRODEM_DEVICE`` or 0)."""
    if device is None:
        device = int(os.environ.get("HYDRODEM_DEVICE", "0"))
    with _lock:
        ctx = _contexts.get(device)
    if ctx is None:
        ctx = Context(device)
        with _lock:
            _contexts[device] = ctx
    return ctx


def device_count():
    lib = load_library()
    n = ctypes.c_int(0)
    lib.hdem_device_count(ctypes.byref(n))
    return n.value


_DTYPES = {np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.uint8),
           np.dtype(np.complex64), np.dtype(np.complex128), np.dtype(np.int64)}


class _HostBlocks:
    """Page-locked host memory for the arrays the host entry points return.

    A result that comes back into a fresh ``np.empty`` pays for its pages three times: the
    faults when they are first written (taken on 16 threads by the library, still ~10 ms per
    GiB), a device-to-host copy that stages through the driver at 40 instead of 56 GB/s, and
    ~50 ms per GiB of unmapping when the array is dropped -- more than the kernels and the
    bus together for every operator here.  Results of 32 MiB and more are therefore NumPy
    arrays over page-locked blocks that return here when the array (and every view of it)
    has gone, and are handed out again for the next result of that size.  The cache is capped
    (``HDEM_HOST_POOL_MIB``, default 8 GiB; 0 = plain ``np.empty``), and so is what may be
    page-locked at any one time, arrays in the caller's hands included
    (``HDEM_HOST_PINNED_MAX_MIB``, default 4 x the cache): beyond it results are plain
    ``np.empty`` arrays.  A block the cache has no room for is unlocked by the next call that
    asks for memory here, not by the finalizer that returned it -- that one runs on whatever
    thread drops the last reference and makes no device call."""

    MIN_BYTES = 32 << 20

    def __init__(self):
        self.lock = threading.Lock()
        self.spare = {}                      # nbytes -> [address, ...]
        self.cached = 0
        self.outstanding = 0                 # page-locked bytes in callers' hands
        self.to_free = []                    # blocks waiting to be unlocked
        mib = os.environ.get("HDEM_HOST_POOL_MIB")
        self.cap = (int(mib) << 20) if mib is not None else (8 << 30)
        mib = os.environ.get("HDEM_HOST_PINNED_MAX_MIB")
        self.pinned_max = (int(mib) << 20) if mib is not None else 4 * self.cap

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        count = int(np.prod(shape, dtype=np.int64))
        nbytes = count * dtype.itemsize
        if nbytes < self.MIN_BYTES or self.cap <= 0:
            return np.empty(shape, dtype)
        try:
            ctx = context()
            with self.lock:
                doomed, self.to_free = self.to_free, []
                stack = self.spare.get(nbytes)
                addr = stack.pop() if stack else None
                if addr is not None:
                    self.cached -= nbytes
                elif self.outstanding + self.cached + nbytes > self.pinned_max:
                    addr = 0                     # enough is page-locked already
                if addr != 0:
                    self.outstanding += nbytes
            for old in doomed:
                ctx.lib.hdem_host_free(ctx.handle, ctypes.c_void_p(old))
            if addr == 0:
                return np.empty(shape, dtype)
            if addr is None:
                ptr = ctypes.c_void_p()
                if ctx.lib.hdem_host_alloc(ctx.handle, nbytes, ctypes.byref(ptr)) != 0 or not ptr.value:
                    with self.lock:
                        self.outstanding -= nbytes
                    return np.empty(shape, dtype)
                addr = ptr.value
            buf = (ctypes.c_char * nbytes).from_address(addr)
            root = np.frombuffer(buf, dtype=dtype, count=count)
            weakref.finalize(root, self._give_back, addr, nbytes)
            return root.reshape(shape)
        except Exception:  # pylint: disable=broad-except
            return np.empty(shape, dtype)

    def _give_back(self, addr, nbytes):
        # (a finalizer: any thread, possibly during interpreter shutdown -- no device call)
        with self.lock:
            self.outstanding -= nbytes
            if self.cached + nbytes <= self.cap:
                self.spare.setdefault(nbytes, []).append(addr)
                self.cached += nbytes
            else:
                self.to_free.append(addr)


_host_blocks = _HostBlocks()


def host_empty(shape, dtype):
    """An uninitialised host array for a result of the library (see :class:`_HostBlocks`)."""
    return _host_blocks.empty(shape, dtype)


class DeviceRaster:
    """A 2-D raster resident in HBM (owning unless wrapped)."""

    def __init__(self, ctx, ptr, shape, dtype, owner=True, keepalive=None):
        self.ctx, self.ptr, self.shape = ctx, ptr, tuple(shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner
        self._keepalive = keepalive

    @property
    def nbytes(self):
        return int(np.prod(self.shape)) * self.dtype.itemsize

    @classmethod
    def empty(cls, shape, dtype, ctx=None):
        ctx = ctx or context()
        dtype = np.dtype(dtype)
        if dtype not in _DTYPES:
            raise ValueError(f"unsupported raster dtype {dtype}")
        ptr = ctypes.c_void_p()
        nbytes = int(np.prod(shape)) * dtype.itemsize
        ctx.check(ctx.lib.hdem_malloc(ctx.handle, nbytes, ctypes.byref(ptr)))
        return cls(ctx, ptr.value, shape, dtype)

    @classmethod
    def from_host(cls, array, dtype=None, ctx=None):
        a = np.ascontiguousarray(array, dtype=dtype or (
            array.dtype if array.dtype in _DTYPES else np.float32))
        if a.ndim != 2:
            raise ValueError(f"expected a 2-D raster, got shape {a.shape}")
        r = cls.empty(a.shape, a.dtype, ctx)
        r.ctx.check(r.ctx.lib.hdem_memcpy_h2d(r.ctx.handle, r.ptr,
                                              a.ctypes.data, a.nbytes))
        return r

    @classmethod
    def wrap(cls, ptr, shape, dtype, ctx=None, keepalive=None):
        """Non-owning view of device memory someone else allocated (e.g. a
        torch tensor: ``wrap(t.data_ptr(), t.shape, np.float32, keepalive=t)``)."""
        return cls(ctx or context(), int(ptr), shape, dtype, owner=False,
                   keepalive=keepalive)

    def to_host(self):
        out = host_empty(self.shape, self.dtype)
        self.ctx.check(self.ctx.lib.hdem_memcpy_d2h(self.ctx.handle,
                                                    out.ctypes.data, self.ptr,
                                                    out.nbytes))
        return out

    def free(self):
        if self._owner and self.ptr:
            self.ctx.lib.hdem_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:  # pylint: disable=broad-except
            pass


# ---------------------------------------------------------------------------
# device-resident operators (thin: one C call each)
# ---------------------------------------------------------------------------

def _need(r, dtype):
    if r.dtype != np.dtype(dtype):
        raise ValueError(f"expected a {np.dtype(dtype)} raster, got {r.dtype}")


def d8_dev(z, out=None):
    _need(z, np.float32)
    out = out or DeviceRaster.empty(z.shape, np.uint8, z.ctx)
    c = z.ctx
    c.check(c.lib.hdem_d8_f32_dev(c.handle, z.ptr, z.shape[0], z.shape[1], out.ptr))
    return out


def sinkfill_dev(z, eps=0.0, max_rounds=0, out=None, flags=FILL_INIT):
    _need(z, np.float32)
    out = out or DeviceRaster.empty(z.shape, np.float32, z.ctx)
    c = z.ctx
    st = FillStats()
    c.check(c.lib.hdem_sinkfill_f32_dev(c.handle, z.ptr, z.shape[0], z.shape[1],
                                        float(eps), int(max_rounds), int(flags),
                                        out.ptr, ctypes.byref(st)))
    return out, st.as_dict()


def blockmax_dev(z, block, out=None):
    """Block-maximum coarsening (``hdem_blockmax_f32_dev``)."""
    _need(z, np.float32)
    shape = (-(-z.shape[0] // block), -(-z.shape[1] // block))
    out = out or DeviceRaster.empty(shape, np.float32, z.ctx)
    c = z.ctx
    c.check(c.lib.hdem_blockmax_f32_dev(c.handle, z.ptr, z.shape[0], z.shape[1], int(block),
                                        out.ptr))
    return out


def elementwise_dev(op, image, operand, out_dtype=None, out=None):
    """``op(image, operand)`` cell by cell on device rasters (``hdem_elementwise_dev``);
    ``operand`` is a :class:`DeviceRaster` of the same shape or a scalar.  Result type:
    ``out_dtype`` or, like NumPy on the stored types, uint8 for comparisons / NONZERO and
    for a product of two masks, float64 as soon as one side is float64, else float32."""
    c = image.ctx
    raster = operand if isinstance(operand, DeviceRaster) else None
    if raster is not None and raster.shape != image.shape:
        raise ValueError(f"operand shape {raster.shape} != image shape {image.shape}")
    if image.dtype not in _EW_TYPES or (raster is not None and raster.dtype not in _EW_TYPES):
        raise ValueError("element-wise operators take float32, float64, uint8 or int64 rasters")
    if out_dtype is None:
        kinds = [image.dtype] + ([raster.dtype] if raster is not None else [])
        if op in (EW_GT, EW_LT, EW_NONZERO):
            out_dtype = np.uint8
        elif any(k == np.float64 for k in kinds):
            out_dtype = np.float64
        elif op == EW_MUL and all(k == np.uint8 for k in kinds) and \
                (raster is not None or float(operand) in (0.0, 1.0)):
            out_dtype = np.uint8
        else:
            out_dtype = np.float32
    out = out or DeviceRaster.empty(image.shape, out_dtype, c)
    n = int(np.prod(image.shape))
    c.check(c.lib.hdem_elementwise_dev(
        c.handle, int(op), image.ptr, _EW_TYPES[image.dtype],
        raster.ptr if raster is not None else None,
        _EW_TYPES[raster.dtype] if raster is not None else 0,
        0.0 if raster is not None else float(operand), n, out.ptr, _EW_TYPES[out.dtype]))
    return out


def copy_rate(ctx=None, nbytes=1 << 30, reps=5):
    """Measured device copy rate in GB/s (bytes moved through HBM = 2 x copied bytes per
    second): the achievable roof SURVEY 8d asks the kernels to be quoted against."""
    c = ctx or context()
    src = DeviceRaster.empty((nbytes // 4096, 1024), np.float32, c)
    dst = DeviceRaster.empty(src.shape, np.float32, c)
    try:
        c.check(c.lib.hdem_copy_rate_dev(c.handle, src.ptr, dst.ptr, src.nbytes))   # warm
        c.synchronize()
        was = c.profile_get(K_COPY)
        c.profile(True)
        for _ in range(reps):
            c.check(c.lib.hdem_copy_rate_dev(c.handle, src.ptr, dst.ptr, src.nbytes))
        now = c.profile_get(K_COPY)
        ms = now["ms"] - was["ms"]
        return 2.0 * (now["units"] - was["units"]) / max(ms, 1e-9) / 1e6
    finally:
        src.free()
        dst.free()


def fourier_destripe_dev(dem, out=None, mask=None):
    """DetectApplyFourier on a device raster; ``mask`` (uint8 raster) optionally
    receives the reference's ``masks_fourier`` (shifted coordinates)."""
    _need(dem, np.float32)
    c = dem.ctx
    out = out or DeviceRaster.empty(dem.shape, np.float32, c)
    if mask is not None:
        _need(mask, np.uint8)
    quarter = (dem.shape[0] // 2 - 10, dem.shape[1] // 2 - 10)
    c.check(c.lib.hdem_fourier_destripe_f32_dev(c.handle, dem.ptr, dem.shape[0], dem.shape[1],
                                                out.ptr, mask.ptr if mask is not None else None),
            window=55, shape=tuple(max(q, 0) for q in quarter))
    return out


def blanks_fourier_dev(q, found=None, window_size=55):
    """One BlanksFourier pass: returns the byte mask of cells above 4x their hollow
    mean; ``q`` is rewritten with those cells zeroed."""
    _need(q, np.float32)
    c = q.ctx
    found = found or DeviceRaster.empty(q.shape, np.uint8, c)
    c.check(c.lib.hdem_blanks_fourier_f32_dev(c.handle, q.ptr, q.shape[0], q.shape[1],
                                              int(window_size), found.ptr),
            window=window_size, shape=q.shape)
    return found


def isolated_points_dev(mask, window_size=3, out=None):
    _need(mask, np.uint8)
    c = mask.ctx
    out = out or DeviceRaster.empty(mask.shape, np.uint8, c)
    c.check(c.lib.hdem_isolated_points_u8_dev(c.handle, mask.ptr, mask.shape[0], mask.shape[1],
                                              int(window_size), out.ptr),
            window=window_size, shape=mask.shape)
    return out


def expand_dev(mask, window_size=13, out=None):
    _need(mask, np.uint8)
    c = mask.ctx
    out = out or DeviceRaster.empty(mask.shape, np.uint8, c)
    c.check(c.lib.hdem_expand_u8_dev(c.handle, mask.ptr, mask.shape[0], mask.shape[1],
                                     int(window_size), out.ptr),
            window=window_size, shape=mask.shape)
    return out


def fft2_dev(data, inverse=False):
    """In-place 2-D transform of a complex64 or complex128 raster; the inverse is
    unnormalised."""
    c = data.ctx
    if data.dtype == np.complex128:
        fn = c.lib.hdem_fft2_c2c_f64_dev
    else:
        _need(data, np.complex64)
        fn = c.lib.hdem_fft2_c2c_f32_dev
    c.check(fn(c.handle, data.ptr, data.shape[0], data.shape[1], int(bool(inverse))))
    return data


def widened_to_host(raster, dtype):
    """``raster.to_host().astype(dtype)`` with the conversion on the device: the wider array
    crosses the bus at 56 GB/s instead of being written by one host thread at ~10 (the
    reference hands back float64 / int64 where the kernels hold float32 / bytes)."""
    dtype = np.dtype(dtype)
    if raster.dtype == dtype:
        return raster.to_host()
    wide = elementwise_dev(EW_MUL, raster, 1.0, out_dtype=dtype)
    try:
        return wide.to_host()
    finally:
        wide.free()


def correct_nan_dev(dem, out=None, window_size=3):
    _need(dem, np.float32)
    c = dem.ctx
    out = out or DeviceRaster.empty(dem.shape, np.float32, c)
    c.check(c.lib.hdem_correct_nan_f32_dev(c.handle, dem.ptr, dem.shape[0], dem.shape[1],
                                           int(window_size), out.ptr),
            window=window_size, shape=dem.shape)
    return out


def majority_dev(img, window_size=11, out=None):
    _need(img, np.float32)
    c = img.ctx
    out = out or DeviceRaster.empty(img.shape, np.float32, c)
    c.check(c.lib.hdem_majority_f32_dev(c.handle, img.ptr, img.shape[0], img.shape[1],
                                        int(window_size), out.ptr),
            window=window_size, shape=img.shape)
    return out


def _structure_arg(structure):
    if structure is None:
        return None, 0, 0, None
    st = np.ascontiguousarray(np.asarray(structure) != 0, dtype=np.uint8)
    if st.ndim != 2:
        raise ValueError(f"expected a 2-D structure, got shape {st.shape}")
    return st.ctypes.data, st.shape[0], st.shape[1], st


def binary_erosion_dev(mask, iterations=1, structure=None, out=None):
    _need(mask, np.uint8)
    c = mask.ctx
    out = out or DeviceRaster.empty(mask.shape, np.uint8, c)
    tmp = DeviceRaster.empty(mask.shape, np.uint8, c) if iterations > 1 else None
    ptr, sh, sw, keep = _structure_arg(structure)
    try:
        c.check(c.lib.hdem_binary_erosion_u8_dev(c.handle, mask.ptr, mask.shape[0],
                                                 mask.shape[1], ptr, sh, sw, int(iterations),
                                                 tmp.ptr if tmp else None, out.ptr))
    finally:
        if tmp is not None:
            c.synchronize()
            tmp.free()
    del keep
    return out


def binary_closing_dev(mask, structure=None, out=None):
    _need(mask, np.uint8)
    c = mask.ctx
    out = out or DeviceRaster.empty(mask.shape, np.uint8, c)
    tmp = DeviceRaster.empty(mask.shape, np.uint8, c)
    ptr, sh, sw, keep = _structure_arg(structure)
    try:
        c.check(c.lib.hdem_binary_closing_u8_dev(c.handle, mask.ptr, mask.shape[0],
                                                 mask.shape[1], ptr, sh, sw, tmp.ptr, out.ptr))
    finally:
        c.synchronize()
        tmp.free()
    del keep
    return out


def grey_dilation_dev(img, size, out=None):
    """float32 or float64 raster; the result has the input's type."""
    c = img.ctx
    if img.dtype == np.float64:
        fn = c.lib.hdem_grey_dilation_f64_dev
    else:
        _need(img, np.float32)
        fn = c.lib.hdem_grey_dilation_f32_dev
    out = out or DeviceRaster.empty(img.shape, img.dtype, c)
    sy, sx = (size, size) if np.isscalar(size) else size
    c.check(fn(c.handle, img.ptr, img.shape[0], img.shape[1], int(sy), int(sx), out.ptr))
    return out


def tidying_lagoons_dev(img, out=None):
    _need(img, np.float32)
    c = img.ctx
    out = out or DeviceRaster.empty(img.shape, np.float32, c)
    c.check(c.lib.hdem_tidying_lagoons_f32_dev(c.handle, img.ptr, img.shape[0], img.shape[1],
                                               out.ptr), window=7, shape=img.shape)
    return out


def lagoons_detection_dev(hsheds):
    """(mask uint8, hsheds_nan_fixed, lagoons_values) device rasters."""
    _need(hsheds, np.float32)
    c = hsheds.ctx
    fixed = DeviceRaster.empty(hsheds.shape, np.float32, c)
    values = DeviceRaster.empty(hsheds.shape, np.float32, c)
    mask = DeviceRaster.empty(hsheds.shape, np.uint8, c)
    c.check(c.lib.hdem_lagoons_detection_f32_dev(c.handle, hsheds.ptr, hsheds.shape[0],
                                                 hsheds.shape[1], fixed.ptr, values.ptr,
                                                 mask.ptr), window=11, shape=hsheds.shape)
    return mask, fixed, values


def sinkfill_d8_dev(z, eps=0.0, max_rounds=0, out=None, codes=None, flags=FILL_INIT):
    """Sink fill + D8 of the filled surface (``hdem_sinkfill_d8_f32_dev``).
    Returns (filled raster, D8 raster, stats)."""
    _need(z, np.float32)
    c = z.ctx
    out = out or DeviceRaster.empty(z.shape, np.float32, c)
    codes = codes or DeviceRaster.empty(z.shape, np.uint8, c)
    st = FillStats()
    c.check(c.lib.hdem_sinkfill_d8_f32_dev(c.handle, z.ptr, z.shape[0], z.shape[1], float(eps),
                                           int(max_rounds), int(flags), out.ptr, codes.ptr,
                                           ctypes.byref(st)))
    return out, codes, st.as_dict()


def boxmean3_dev(x, do_round=True, out=None):
    out = out or DeviceRaster.empty(x.shape, x.dtype, x.ctx)
    c = x.ctx
    fn = (c.lib.hdem_boxmean3_f32_dev if x.dtype == np.float32
          else c.lib.hdem_boxmean3_f64_dev)
    if x.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError(f"box mean needs a float raster, got {x.dtype}")
    c.check(fn(c.handle, x.ptr, x.shape[0], x.shape[1], int(bool(do_round)), out.ptr))
    return out


def quadratic_dev(dem, window_size=15, out=None):
    _need(dem, np.float32)
    out = out or DeviceRaster.empty(dem.shape, np.float32, dem.ctx)
    c = dem.ctx
    c.check(c.lib.hdem_quadratic_f32_dev(c.handle, dem.ptr, dem.shape[0],
                                         dem.shape[1], int(window_size), out.ptr),
            window=window_size, shape=dem.shape)
    return out


def groves_dev(img, groves, window_size=15, threshold=1.5, iterations=3, out=None,
               scratch=None):
    """``scratch``: a float32 raster of the same shape for the ping-pong between
    passes (allocated and freed here when not given)."""
    _need(img, np.float32)
    _need(groves, np.uint8)
    c = img.ctx
    out = out or DeviceRaster.empty(img.shape, np.float32, c)
    if scratch is not None:
        _need(scratch, np.float32)
        c.check(c.lib.hdem_groves_f32_dev(c.handle, img.ptr, groves.ptr, img.shape[0],
                                          img.shape[1], int(window_size),
                                          float(threshold), int(iterations),
                                          scratch.ptr, out.ptr),
                window=window_size, shape=img.shape)
        return out
    scratch = DeviceRaster.empty(img.shape, np.float32, c) if iterations > 1 else None
    try:
        c.check(c.lib.hdem_groves_f32_dev(c.handle, img.ptr, groves.ptr, img.shape[0],
                                          img.shape[1], int(window_size),
                                          float(threshold), int(iterations),
                                          scratch.ptr if scratch else None, out.ptr),
                window=window_size, shape=img.shape)
    finally:
        if scratch is not None:
            c.synchronize()
            scratch.free()
    return out


# ---------------------------------------------------------------------------
# host-array operators (what Filter.apply binds): NumPy in, NumPy out
# ---------------------------------------------------------------------------

def _host2d(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.ndim != 2:
        raise ValueError(f"expected a 2-D raster, got shape {a.shape}")
    return a


def d8(z):
    c = context()
    z = _host2d(z, np.float32)
    out = host_empty(z.shape, np.uint8)
    c.check(c.lib.hdem_d8_f32(c.handle, z.ctypes.data, z.shape[0], z.shape[1],
                              out.ctypes.data))
    return out


def sinkfill(z, eps=0.0, max_rounds=0, return_stats=False):
    c = context()
    z = _host2d(z, np.float32)
    out = host_empty(z.shape, z.dtype)
    st = FillStats()
    c.check(c.lib.hdem_sinkfill_f32(c.handle, z.ctypes.data, z.shape[0], z.shape[1],
                                    float(eps), int(max_rounds), out.ctypes.data,
                                    ctypes.byref(st)))
    return (out, st.as_dict()) if return_stats else out


def boxmean3(x, do_round=True):
    c = context()
    if np.asarray(x).dtype == np.float32:
        x = _host2d(x, np.float32)
        fn = c.lib.hdem_boxmean3_f32
    else:
        x = _host2d(x, np.float64)
        fn = c.lib.hdem_boxmean3_f64
    out = host_empty(x.shape, x.dtype)
    c.check(fn(c.handle, x.ctypes.data, x.shape[0], x.shape[1], int(bool(do_round)),
               out.ctypes.data))
    return out


def convolve(x, weights):
    """General odd weights: float32 rasters in float32, anything else in float64 -- the
    type scipy.ndimage.convolve works in for a float64 raster (extension_filters.py:183)."""
    c = context()
    if np.asarray(x).dtype == np.float32:
        x, fn = _host2d(x, np.float32), c.lib.hdem_convolve_f32
    else:
        x, fn = _host2d(x, np.float64), c.lib.hdem_convolve_f64
    w = _host2d(weights, np.float64)
    out = host_empty(x.shape, x.dtype)
    c.check(fn(c.handle, x.ctypes.data, x.shape[0], x.shape[1], w.ctypes.data, w.shape[0],
               w.shape[1], out.ctypes.data))
    return out


def around(x):
    c = context()
    if np.asarray(x).dtype == np.float32:
        a, fn = np.ascontiguousarray(x, dtype=np.float32), c.lib.hdem_around_f32
    else:
        a, fn = np.ascontiguousarray(x, dtype=np.float64), c.lib.hdem_around_f64
    out = host_empty(a.shape, a.dtype)
    if a.size:
        c.check(fn(c.handle, a.ctypes.data, a.size, out.ctypes.data))
    return out


def quadratic(dem, window_size=15):
    c = context()
    dem = _host2d(dem, np.float32)
    out = host_empty(dem.shape, dem.dtype)
    c.check(c.lib.hdem_quadratic_f32(c.handle, dem.ctypes.data, dem.shape[0],
                                     dem.shape[1], int(window_size), out.ctypes.data),
            window=window_size, shape=dem.shape)
    return out


def fourier_destripe(dem, return_mask=False):
    c = context()
    dem = _host2d(dem, np.float32)
    out = host_empty(dem.shape, dem.dtype)
    mask = host_empty(dem.shape, np.uint8) if return_mask else None
    quarter = (dem.shape[0] // 2 - 10, dem.shape[1] // 2 - 10)
    c.check(c.lib.hdem_fourier_destripe_f32(c.handle, dem.ctypes.data, dem.shape[0],
                                            dem.shape[1], out.ctypes.data,
                                            mask.ctypes.data if return_mask else None),
            window=55, shape=tuple(max(q, 0) for q in quarter))
    return (out, mask) if return_mask else out


def blanks_fourier(q, window_size=55):
    """(found float64 0/1, q * (1 - found)) like BlanksFourier.apply."""
    qd = DeviceRaster.from_host(_host2d(q, np.float32))
    found = blanks_fourier_dev(qd, window_size=window_size)
    return widened_to_host(found, np.float64), qd.to_host()


def isolated_points(mask, window_size=3):
    m = DeviceRaster.from_host(_host2d(mask, np.uint8))
    return isolated_points_dev(m, window_size).to_host()


def expand(mask, window_size=13, dtype=np.uint8):
    """ExpandFilter on a host raster (cells > 0 are set); the 0 / 1 result in ``dtype``."""
    g = np.asarray(mask)
    if g.dtype in (np.uint8, np.bool_):
        m = DeviceRaster.from_host(_host2d(g, g.dtype).view(np.uint8))
    else:
        m = DeviceRaster.from_host(_host2d(np.greater(g, 0), np.bool_).view(np.uint8))
    return widened_to_host(expand_dev(m, window_size), dtype)


def fft2(x, inverse=False):
    """fft2 / ifft2 (normalised) of a 2-D array: complex64 for float32 / complex64 input,
    complex128 for anything else -- scipy.fftpack's rule (extension_filters.py:379,414)."""
    single = np.asarray(x).dtype in (np.dtype(np.float32), np.dtype(np.complex64))
    a = _host2d(x, np.complex64 if single else np.complex128)
    d = fft2_dev(DeviceRaster.from_host(a), inverse).to_host()
    if not inverse:
        return d
    return d / (np.float32(a.size) if single else np.float64(a.size))


def mask_bytes(groves_class):
    """The class raster as the bytes the groves kernel reads (non-zero = grove).  One-byte
    dtypes go as they are -- at 16384^2 a ``!= 0`` and an ``astype`` are 100 ms of NumPy
    in front of 2 ms of kernels."""
    g = np.asarray(groves_class)
    if g.dtype in (np.uint8, np.bool_, np.int8):
        return _host2d(g, g.dtype).view(np.uint8)
    return _host2d(np.not_equal(g, 0), np.bool_).view(np.uint8)


def groves(img, groves_class, window_size=15, threshold=1.5, iterations=1):
    c = context()
    img = _host2d(img, np.float32)
    g = mask_bytes(groves_class)
    if g.shape != img.shape:
        raise ValueError(f"groves class shape {g.shape} != image shape {img.shape}")
    out = host_empty(img.shape, img.dtype)
    c.check(c.lib.hdem_groves_f32(c.handle, img.ctypes.data, g.ctypes.data,
                                  img.shape[0], img.shape[1], int(window_size),
                                  float(threshold), int(iterations), out.ctypes.data),
            window=window_size, shape=img.shape)
    return out


__all__ = [n for n in dir() if not n.startswith("_")]
_ = HydroDEMException  # re-exported for callers that catch the base class
