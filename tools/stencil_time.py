"""Time the radius-1 stencils (D8, 3 x 3 box mean + round) on an n x n raster (exploration).
usage: python tools/stencil_time.py [n]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = B.context()
z = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n))
codes = B.DeviceRaster.empty((n, n), np.uint8)
out = B.DeviceRaster.empty((n, n), np.float32)
def timed(f, reps=10):
    f(); ctx.synchronize(); best = 1e9
    for _ in range(reps):
        t = time.time(); f(); ctx.synchronize(); best = min(best, time.time() - t)
    return best
t = timed(lambda: B.d8_dev(z, out=codes)); print(f"d8      {t*1e3:.3f} ms  {5*n*n/t/1e9:.0f} GB/s")
t = timed(lambda: B.boxmean3_dev(z, True, out=out)); print(f"boxmean {t*1e3:.3f} ms  {8*n*n/t/1e9:.0f} GB/s")
