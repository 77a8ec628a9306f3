"""debug: hub-start fill against the C oracle at several shapes (exploration)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
from oracle import c_oracle
import hdem_synth
cases = [(1500, 1300, False), (1500, 1300, True), (1054, 1054, False), (1056, 1055, False), (2048, 2048, False), (4096, 4096, False)]
for h, w, nan in cases:
    z = hdem_synth.synth_dem(h, w)
    if nan:
        z[700:720, 100:130] = np.nan
    want = c_oracle.sinkfill_pflood(z)
    zd = B.DeviceRaster.from_host(z)
    out, st = B.sinkfill_dev(zd)
    got = out.to_host()
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    print(h, w, nan, "mismatches", int(bad.sum()), "visits/tile %.2f" % (st["tile_visits"] / st["tiles"]), "rounds", st["rounds"], flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "got<want", int((got[bad] < want[bad]).sum()), "got>want", int((got[bad] > want[bad]).sum()))
        for y, x in list(zip(ys, xs))[:8]:
            print("   ", y, x, "tile", (y - 1) // 62, (x - 1) // 62, "in-tile", (y - 1) % 62, (x - 1) % 62, got[y, x], want[y, x], z[y, x])
    out.free(); zd.free()
