import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hydrodem_amd as hd
from hydrodem_amd import backend as B
import hdem_synth
n = 16384
img, mask = hdem_synth.synth_dem(n, n, pits=False), hdem_synth.synth_groves(n, n)
for rep in range(3):
    t = time.time(); f = hd.GrovesCorrectionsIter(mask, iterations=3); a = f.apply(img); dt = time.time() - t
    print(f"GrovesCorrectionsIter(mask u8).apply(ndarray): {dt*1e3:.1f} ms")
t = time.time(); ok = hd.GrovesCorrection._is_mask(mask); print("is_mask", ok, f"{(time.time()-t)*1e3:.1f} ms")
t = time.time(); g = B.mask_bytes(mask); print("mask_bytes", f"{(time.time()-t)*1e3:.1f} ms", g is mask or g.base is mask)
mf = mask.astype(np.float32)
t = time.time(); a2 = hd.GrovesCorrectionsIter(mf, iterations=3).apply(img); print(f"float32 class raster: {(time.time()-t)*1e3:.1f} ms", np.array_equal(a, a2))
c = B.context()
for rep in range(2):
    t = time.time(); a3 = B.groves(img, mask, iterations=3); print(f"backend.groves: {(time.time()-t)*1e3:.1f} ms")
d = B.DeviceRaster.empty(img.shape, np.float32)
for rep in range(2):
    t = time.time(); c.check(c.lib.hdem_memcpy_h2d(c.handle, d.ptr, img.ctypes.data, img.nbytes)); print(f"H2D 1 GiB pageable: {(time.time()-t)*1e3:.1f} ms")
for rep in range(2):
    t = time.time(); o = np.empty_like(img); c.check(c.lib.hdem_memcpy_d2h(c.handle, o.ctypes.data, d.ptr, o.nbytes)); print(f"D2H 1 GiB into a fresh array: {(time.time()-t)*1e3:.1f} ms")
t = time.time(); c.check(c.lib.hdem_memcpy_d2h(c.handle, o.ctypes.data, d.ptr, o.nbytes)); print(f"D2H 1 GiB into a touched array: {(time.time()-t)*1e3:.1f} ms")
t = time.time(); del o, a3; print(f"freeing 2 GiB of host arrays: {(time.time()-t)*1e3:.1f} ms")
