"""Time MajorityFilter (window 11) on the n x n HydroSHEDS-like raster of the bench and
print a checksum of the result (exploration; HDEM_MAJORITY_FIRST_FORM=1 selects the first
kernel form).  usage: python tools/majority_time.py [n]"""
import sys, os, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = B.context()
hs = np.round(hdem_synth.synth_dem(n, n, pits=False))
hs[n // 3:n // 3 + n // 40, n // 2:n // 2 + n // 30] = 212.0          # a lake
src = B.DeviceRaster.from_host(hs.astype(np.float32))
out = B.DeviceRaster.empty((n, n), np.float32)
for rep in range(8):
    ctx.synchronize(); t = time.time()
    B.majority_dev(src, 11, out=out)
    ctx.synchronize(); dt = time.time() - t
    print(f"majority 11 {n}^2: {dt*1e3:.3f} ms  ({8*n*n/dt/1e9:.0f} GB/s algorithmic)")
res = out.to_host()
print("nonzero", int((res != 0).sum()), "crc", zlib.crc32(res.tobytes()))
