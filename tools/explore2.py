"""Micro-timings of the sink-fill visit (exploration only)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = B.context()
z = oracle.synth_dem(n, n)
zd = B.DeviceRaster.from_host(z); wd = B.DeviceRaster.empty(z.shape, np.float32)
_, st = B.sinkfill_dev(zd, out=wd); ctx.synchronize()
print("full:", st)
for rep in range(3):
    ctx.profile(True); ctx.profile_reset()
    t = time.time()
    _, st = B.sinkfill_dev(zd, out=wd, flags=B.FILL_WARM | B.FILL_SYNC_ONLY)
    ctx.synchronize(); dt = time.time() - t
    k = ctx.profile_get(B.K_FILL_ROUND)
    print(f"verify-only pass: wall {dt*1e3:.3f} ms kernel {k['ms']:.3f} ms launches {k['launches']} visits {st['tile_visits']} -> {k['ms']*1e3*2048/max(st['tile_visits'],1):.2f} us per visit-slot; GB/s (8B/cell) {8*n*n/k['ms']/1e6:.0f}")
