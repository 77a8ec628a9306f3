"""The headline step (fill + D8, 16384^2) a few times with its phases from the library's
own timers (exploration).  usage: python tools/step_phases.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = 16384
ctx = B.context()
zd = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n)); wd = B.DeviceRaster.empty((n, n), np.float32)
dd = B.DeviceRaster.empty((n, n), np.uint8)
for rep in range(6):
    ctx.profile(True); ctx.profile_reset()
    t = time.time(); B.sinkfill_d8_dev(zd, out=wd, codes=dd); ctx.synchronize(); dt = time.time() - t
    print(f"step {dt*1e3:.2f} ms: init {ctx.profile_get(B.K_FILL_INIT)['ms']:.3f} tile {ctx.profile_get(B.K_FILL_TILE)['ms']:.3f} coarse {ctx.profile_get(B.K_FILL_COARSE)['ms']:.3f} blockmax {ctx.profile_get(B.K_BLOCKMAX)['ms']:.3f}")
