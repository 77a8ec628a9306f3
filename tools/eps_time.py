"""Gradient fill (eps > 0) + D8 at 16384^2: time and visits per tile (exploration).
usage: python tools/eps_time.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = B.context()
zd = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n)); wd = B.DeviceRaster.empty((n, n), np.float32); dd = B.DeviceRaster.empty((n, n), np.uint8)
for eps in (1e-3, 1e-4):
    for rep in range(3):
        ctx.synchronize(); t = time.perf_counter(); _, _, st = B.sinkfill_d8_dev(zd, eps=eps, out=wd, codes=dd); ctx.synchronize()
        print(f"eps {eps}: {1e3*(time.perf_counter()-t):.2f} ms, visits/tile {st['tile_visits']/st['tiles']:.2f}", flush=True)
