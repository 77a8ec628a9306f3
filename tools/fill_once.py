"""Time the sink fill of an n x n synthetic DEM a few times with the trace on
(exploration only).  usage: python tools/fill_once.py [n] [reps]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HDEM_FILL_TRACE", "1")
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = B.context()
zd = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n)); wd = B.DeviceRaster.empty((n, n), np.float32)
ref = None
for rep in range(reps):
    ctx.profile(True); ctx.profile_reset()
    t = time.time(); _, st = B.sinkfill_dev(zd, out=wd); ctx.synchronize(); dt = time.time() - t
    k = ctx.profile_get(B.K_FILL_TILE)
    print(f"fill {n}^2: wall {dt*1e3:.2f} ms, async kernel {k['ms']:.2f} ms, visits {st['tile_visits']} unchanged {st['visits_unchanged']}")
if n <= 4096:
    # (the tests hold the parity checks; this is for telling two builds apart)
    import zlib
    print("crc32 of the filled raster: %08x" % zlib.crc32(wd.to_host().tobytes()))
