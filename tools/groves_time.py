"""Time GrovesCorrectionsIter (3 iterations of the 15 x 15 quadratic filter + blend) on an
n x n raster and print a checksum (exploration).  usage: python tools/groves_time.py [n] [density]"""
import sys, os, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = B.context()
img = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n))
if len(sys.argv) > 2:
    g = (np.random.default_rng(1).random((n, n), dtype=np.float32) < float(sys.argv[2])).astype(np.uint8)
else:
    g = hdem_synth.synth_groves(n, n)
print("groves density", float(g.mean()))
gr = B.DeviceRaster.from_host(g)
out = B.DeviceRaster.empty((n, n), np.float32)
scr = B.DeviceRaster.empty((n, n), np.float32)
for rep in range(10):
    ctx.synchronize(); t = time.time()
    B.groves_dev(img, gr, 15, 1.5, 3, out=out, scratch=scr)
    ctx.synchronize(); dt = time.time() - t
    print(f"groves x3 {n}^2: {dt*1e3:.2f} ms  ({9*3*n*n/dt/1e9:.0f} GB/s at 27 B/cell)")
res = out.to_host()
print("crc", zlib.crc32(res.tobytes()), "changed", int((res != img.to_host()).sum()))
