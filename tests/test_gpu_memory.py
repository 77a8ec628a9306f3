"""
hdem_malloc / hdem_free keep freed device blocks for the next request of about their size
(hdem_core.hip): the same block comes back, the wait for its last user is on the stream and
not on the host, and what one raster wrote is never seen through another.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from hydrodem_amd import backend
import oracle
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def test_a_freed_block_is_handed_out_again(built):
    ctx = backend.Context(0)
    a = backend.DeviceRaster.empty((1000, 1000), np.float32, ctx)
    pa = a.ptr
    a.free()
    b = backend.DeviceRaster.empty((1000, 1000), np.float32, ctx)
    assert b.ptr == pa                                     # exact size: the cached block
    c = backend.DeviceRaster.empty((1000, 1000), np.float32, ctx)
    assert c.ptr != pa                                     # ... which is out now
    b.free()
    d = backend.DeviceRaster.empty((990, 1000), np.float32, ctx)
    assert d.ptr == pa                                     # 1 % smaller: still that block
    d.free()
    e = backend.DeviceRaster.empty((500, 1000), np.float32, ctx)
    assert e.ptr != pa                                     # half the size: a block of its own
    for r in (c, e):
        r.free()
    ctx.close()


def test_results_survive_the_reuse_of_their_neighbours(built):
    """Operators enqueue, hand their scratch back and the next operator takes it over while
    the first may still be running: the answers must not care."""
    ctx = backend.Context(0)
    z = oracle.synth_dem(1500, 1300)
    want_w = c_oracle.sinkfill_pflood(z)
    want_d = c_oracle.d8(want_w)
    zd = backend.DeviceRaster.from_host(z, ctx=ctx)
    for _ in range(4):
        w, codes, _ = backend.sinkfill_d8_dev(zd)
        box = backend.boxmean3_dev(w)                      # allocates, runs behind the fill
        got_w, got_d = w.to_host(), codes.to_host()
        w.free()
        codes.free()
        again = backend.d8_dev(backend.DeviceRaster.from_host(got_w, ctx=ctx))   # reuses w's block
        assert np.array_equal(got_w, want_w) and np.array_equal(got_d, want_d)
        assert np.array_equal(again.to_host(), want_d)
        assert np.array_equal(box.to_host(), c_oracle.boxmean3(want_w, True))
        box.free()
        again.free()
    ctx.close()


def test_the_cache_can_be_switched_off(built):
    code = ("import numpy as np\n"
            "from hydrodem_amd import backend as B\n"
            "a = B.DeviceRaster.empty((2000, 2000), np.float32); p = a.ptr; a.free()\n"
            "keep = [B.DeviceRaster.empty((64, 64), np.uint8) for _ in range(8)]\n"
            "b = B.DeviceRaster.empty((2000, 2000), np.float32)\n"
            "print('same' if b.ptr == p else 'other')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HDEM_POOL_MIB="0", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    # (with the cache off the block goes back to the driver; whether hipMalloc returns the same
    # address again is the driver's business -- the call just has to work)
    assert out.stdout.strip() in ("same", "other")


def test_large_results_come_back_in_reused_page_locked_blocks(built):
    """Host results of 32 MiB and more are NumPy arrays over page-locked blocks that return
    to the library when the array and all its views have gone (backend._HostBlocks)."""
    import gc
    z = oracle.synth_dem(3000, 3000)                       # 36 MB
    zd = backend.DeviceRaster.from_host(z)
    a = zd.to_host()
    assert type(a) is np.ndarray and a.flags.writeable and np.array_equal(a, z)
    addr = a.ctypes.data
    view = a[100:200]
    del a
    gc.collect()
    b = zd.to_host()                                       # the view keeps the block
    assert b.ctypes.data != addr and np.array_equal(view, z[100:200])
    del view
    gc.collect()
    c = zd.to_host()
    assert c.ctypes.data == addr and np.array_equal(c, z)
    small = backend.DeviceRaster.from_host(z[:100]).to_host()      # below the threshold: np.empty
    assert np.array_equal(small, z[:100])
    # an operator's host form returns such an array too, and it behaves like any other
    d = backend.d8(c_oracle.sinkfill_pflood(z))
    assert d.dtype == np.uint8 and d.sum() > 0
    d2 = d.copy()
    d[:] = 0
    assert d2.sum() > 0


def test_trim_returns_the_cache_and_the_scratch(built):
    """hdem_trim: cached blocks and the scratch buffers the context keeps between calls go
    back to the device (a process that shares the GPU with another allocator), the context
    stays usable and grows them back."""
    ctx = backend.Context(0)
    z = oracle.synth_dem(1200, 1100)
    zd = backend.DeviceRaster.from_host(z, ctx=ctx)
    w, _ = backend.sinkfill_dev(zd)                        # hub buffers, workspace
    want = w.to_host()
    w.free()                                               # parked in the cache
    released = ctx.trim()
    assert released >= z.nbytes                            # at least the parked raster
    assert ctx.trim() == 0                                 # nothing left to give
    w2, _ = backend.sinkfill_dev(zd)
    assert np.array_equal(w2.to_host(), want)
    for r in (zd, w2):
        r.free()
    ctx.close()


def test_two_contexts_and_a_changed_stream(built):
    """The cache's contract is single-stream (include/hydrodem_hip.h): a raster of context A
    that context B works on is synchronised on B by the caller before A frees it -- then A may
    hand the block out again at once -- and a context whose stream was changed since a block
    was handed out waits for the device when the block comes back."""
    import torch
    a, b = backend.Context(0), backend.Context(0)
    z = oracle.synth_dem(900, 1000)
    want = c_oracle.boxmean3(z, True)
    for _ in range(3):
        za = backend.DeviceRaster.from_host(z, ctx=a)
        # B's operator reads A's raster (wrapped: B does not own it)
        seen = backend.DeviceRaster.wrap(za.ptr, za.shape, np.float32, ctx=b, keepalive=za)
        out_b = backend.boxmean3_dev(seen)
        b.synchronize()                                    # the caller's part of the contract
        block = za.ptr
        za.free()
        again = backend.DeviceRaster.from_host(np.zeros_like(z), ctx=a)     # takes the block over
        assert again.ptr == block
        assert np.array_equal(out_b.to_host(), want)
        again.free()
        out_b.free()
    # a block handed out under the context's own stream, freed under a caller's stream
    blk = backend.DeviceRaster.from_host(z, ctx=a)
    side = torch.cuda.Stream()
    a.set_stream(side.cuda_stream)
    res = backend.boxmean3_dev(blk)                        # runs on the caller's stream
    blk.free()                                             # stream changed: device-wide wait
    assert np.array_equal(res.to_host(), want)
    res.free()
    a.set_stream(None)
    a.close()
    b.close()
