"""The headline step (fill + D8, 16384^2) a few times with its phases from the library's
own timers (exploration).  usage: python tools/step_phases.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = B.context()
zd = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n)); wd = B.DeviceRaster.empty((n, n), np.float32)
dd = B.DeviceRaster.empty((n, n), np.uint8)
for rep in range(6):
    ctx.profile(True); ctx.profile_reset()
    t = time.time(); _, _, st = B.sinkfill_d8_dev(zd, out=wd, codes=dd); ctx.synchronize(); dt = time.time() - t
    g = lambda k: ctx.profile_get(k)['ms']
    print(f"step {dt*1e3:.2f} ms: hub {g(B.K_FILL_HUB):.3f} blockmax {g(B.K_BLOCKMAX):.3f} coarse {g(B.K_FILL_COARSE):.3f} "
          f"init/apply {g(B.K_FILL_INIT):.3f} tile {g(B.K_FILL_TILE):.3f} flat {g(B.K_FILL_FLAT):.3f} certify {g(B.K_FILL_ROUND):.3f}; visits {st['tile_visits']} "
          f"({st['tile_visits']/st['tiles']:.2f}/tile) unchanged {st['visits_unchanged']} flat {st['visits_flat']} iterations {st['iterations']}", flush=True)
