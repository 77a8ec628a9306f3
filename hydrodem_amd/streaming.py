"""
Raster I/O seam: run a windowed operator over a raster that lives on
the host -- an ndarray, a ``numpy.memmap`` of a file larger than RAM, or anything
the caller reads band by band, which is how a GDAL dataset hands out windows --
without ever holding the whole raster on the device, and with the transfers
hidden behind the kernels.

The reference reads and writes whole arrays (`utils_dem.py:17-40`,
``ReadAsArray()`` / ``WriteArray()``); only this hand-off changes, GDAL itself
is untouched: the caller's reader fills a page-locked band buffer
(``read(r0, r1, out_view)``), the writer receives a page-locked result view
(``write(r0, r1, view)``).

Schedule.  The raster is cut into bands of ``band_rows`` rows plus ``halo`` rows
of overlap on each side (the operator's reach: 1 for D8 and the 3 x 3 mean, 7 per
quadratic pass, 21 for three groves passes; the overlap is recomputed, never
exchanged).  ``depth`` slots, each with its own context = its own HIP stream, its
own pinned input / output buffers and device rasters, take the bands round robin:
slot k fills its input buffer, queues  H2D -> kernels -> D2H  on its stream, waits
for it and hands the result to the writer.  Every slot is driven by a host thread of
its own, so the host-side copies -- the largest cost once the kernels take a few
milliseconds: a 16384^2 float32 raster is 1 GiB each way through ``np.copyto`` --
overlap with each other and with the other slots' transfers and kernels (NumPy's
copies, ``ctypes`` calls and ``hdem_synchronize`` all release the GIL).  Caller
supplied ``read`` / ``write`` callables are serialised with a lock unless told
otherwise (a GDAL dataset handle is not thread safe), and results are written in band
order.  The raster's own first / last rows are the band's, so border semantics
(untouched rings, reflect) are the operator's own.

Not for the sink fill: its dependences are global (`partition.py` splits it).
"""

import ctypes
import threading

import numpy as np

from . import backend


def band_ranges(total_rows, band_rows, halo):
    """[(r0, r1, lo, hi)]: rows [r0, r1) are produced from input rows [lo, hi)."""
    if band_rows < 1 or halo < 0:
        raise ValueError("band_rows must be >= 1 and halo >= 0")
    out = []
    for r0 in range(0, total_rows, band_rows):
        r1 = min(r0 + band_rows, total_rows)
        out.append((r0, r1, max(r0 - halo, 0), min(r1 + halo, total_rows)))
    return out


class _Pinned:
    """A page-locked host buffer seen as a NumPy array."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        ptr = ctypes.c_void_p()
        ctx.check(ctx.lib.hdem_host_alloc(ctx.handle, nbytes, ctypes.byref(ptr)))
        self.ptr = ptr.value
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.ctx.lib.hdem_host_free(self.ctx.handle, ctypes.c_void_p(self.ptr))
            self.ptr = None


class _Slot:
    def __init__(self, device, rows, cols, in_dtypes, out_dtype, scratch_dtypes):
        self.ctx = backend.Context(device)
        self.host_in = [_Pinned(self.ctx, (rows, cols), dt) for dt in in_dtypes]
        self.host_out = _Pinned(self.ctx, (rows, cols), out_dtype)
        self.dev_in = [backend.DeviceRaster.empty((rows, cols), dt, self.ctx) for dt in in_dtypes]
        self.dev_out = backend.DeviceRaster.empty((rows, cols), out_dtype, self.ctx)
        self.scratch = [backend.DeviceRaster.empty((rows, cols), dt, self.ctx)
                        for dt in scratch_dtypes]

    def free(self):
        self.ctx.synchronize()
        for r in self.dev_in + [self.dev_out] + self.scratch:
            r.free()
        for p in self.host_in + [self.host_out]:
            p.free()
        self.ctx.close()


def _view(raster, rows):
    """First ``rows`` rows of a device raster as a non-owning raster."""
    return backend.DeviceRaster.wrap(raster.ptr, (rows, raster.shape[1]), raster.dtype,
                                     ctx=raster.ctx, keepalive=raster)


class BandStream:
    """Streams ``op`` over host rasters.

    ``op(inputs, out, scratch)``: ``inputs`` is a list of device rasters (one per
    source) holding the band with its halo, ``out`` a device raster of the same
    shape to fill, ``scratch`` device rasters of the same shape (``scratch_dtypes``)
    it may use; it must only queue work on ``inputs[0].ctx`` (every
    ``backend.*_dev`` call does) and must not synchronise.  ``halo``: rows of
    context ``op`` needs on each side for its output rows to equal the
    whole-raster result.
    """

    def __init__(self, shape, op, halo, in_dtypes=(np.float32,), out_dtype=np.float32,
                 scratch_dtypes=(), band_rows=2048, depth=2, device=None):
        self.shape = tuple(shape)
        self.op, self.halo = op, int(halo)
        self.bands = band_ranges(self.shape[0], band_rows, self.halo)
        rows = max(hi - lo for _, _, lo, hi in self.bands)
        device = int(backend.context(device).device)
        self.slots = [_Slot(device, rows, self.shape[1], in_dtypes, out_dtype, scratch_dtypes)
                      for _ in range(max(1, min(depth, len(self.bands))))]

    def close(self):
        for s in self.slots:
            s.free()
        self.slots = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def run(self, sources, sink, threads=True, serial_io=None):
        """``sources``: one per input -- an array-like (sliced by rows) or a callable
        ``read(lo, hi, out_view)``; ``sink``: an array-like or ``write(r0, r1, view)``
        (the view is only valid during the call).  ``threads``: one host thread per slot
        (else everything on the caller's thread, bands one after the other).
        ``serial_io``: never run two caller-supplied callables at once (default: yes when
        any is given; arrays and memmaps are copied concurrently either way).  A callable
        ``sink`` receives the bands in order; an array-like one is written by the slots'
        threads as they finish (1 GiB of float32 through one thread's ``__setitem__`` is
        50 ms -- more than everything else in a 16384^2 groves stream)."""
        user_io = any(callable(s) for s in sources) or callable(sink)
        io_lock = threading.Lock() if (user_io if serial_io is None else serial_io) else None

        def guarded(fn):
            if io_lock is None:
                return fn

            def locked(*args):
                with io_lock:
                    return fn(*args)
            return locked

        readers = [guarded(s) if callable(s) else
                   (lambda lo, hi, out, a=s: np.copyto(out, a[lo:hi])) for s in sources]
        write = guarded(sink) if callable(sink) else \
            (lambda r0, r1, v, a=sink: a.__setitem__(slice(r0, r1), v))
        lib = self.slots[0].ctx.lib
        turn = threading.Condition()
        state = {"next": 0, "error": None}
        ordered = callable(sink)

        def band(k, slot):
            r0, r1, lo, hi = self.bands[k]
            n = hi - lo
            c = slot.ctx
            for read, hbuf, dbuf in zip(readers, slot.host_in, slot.dev_in):
                read(lo, hi, hbuf.array[:n])
                c.check(lib.hdem_memcpy_h2d_async(c.handle, dbuf.ptr, hbuf.ptr,
                                                  n * self.shape[1] * hbuf.array.itemsize))
            self.op([_view(d, n) for d in slot.dev_in], _view(slot.dev_out, n),
                    [_view(d, n) for d in slot.scratch])
            c.check(lib.hdem_memcpy_d2h_async(c.handle, slot.host_out.ptr, slot.dev_out.ptr,
                                              n * self.shape[1] * slot.host_out.array.itemsize))
            c.synchronize()
            if not ordered:
                if state["error"] is None:
                    write(r0, r1, slot.host_out.array[r0 - lo:r1 - lo])
                return
            with turn:                                    # in band order
                while state["next"] != k and state["error"] is None:
                    turn.wait()
            if state["error"] is None:
                write(r0, r1, slot.host_out.array[r0 - lo:r1 - lo])
            with turn:
                state["next"] = k + 1
                turn.notify_all()

        def worker(index):
            try:
                for k in range(index, len(self.bands), len(self.slots)):
                    if state["error"] is not None:
                        return
                    band(k, self.slots[index])
            except BaseException as exc:                  # pylint: disable=broad-except
                with turn:
                    state["error"] = state["error"] or exc
                    turn.notify_all()

        if threads and len(self.slots) > 1:
            pool = [threading.Thread(target=worker, args=(i,)) for i in range(len(self.slots))]
            for t in pool:
                t.start()
            for t in pool:
                t.join()
        else:
            for k in range(len(self.bands)):
                if state["error"] is None:
                    try:
                        band(k, self.slots[k % len(self.slots)])
                    except BaseException as exc:          # pylint: disable=broad-except
                        state["error"] = exc
        if state["error"] is not None:
            raise state["error"]


# ---- the operators of the scope table as band operators -----------------------
# each returns the keyword arguments of BandStream that describe it
def groves_op(iterations=3, window_size=15, threshold=1.5):
    """GrovesCorrectionsIter; sources = [image float32, groves class uint8]."""
    def op(inputs, out, scratch):
        backend.groves_dev(inputs[0], inputs[1], window_size, threshold, iterations, out=out,
                           scratch=scratch[0])
    return dict(op=op, halo=iterations * (window_size // 2), in_dtypes=(np.float32, np.uint8),
                out_dtype=np.float32, scratch_dtypes=(np.float32,))


def boxmean_op(do_round=True):
    """PostProcessingFinal (3 x 3 mean, optional rounding)."""
    return dict(op=lambda inputs, out, scratch: backend.boxmean3_dev(inputs[0], do_round, out=out),
                halo=1, in_dtypes=(np.float32,), out_dtype=np.float32)


def d8_op():
    """D8 flow direction of an already filled surface."""
    return dict(op=lambda inputs, out, scratch: backend.d8_dev(inputs[0], out=out),
                halo=1, in_dtypes=(np.float32,), out_dtype=np.uint8)
