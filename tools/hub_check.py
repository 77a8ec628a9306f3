"""Hub-start fill against the C oracle at a few shapes, with the mismatching cells listed
(exploration / debugging).  usage: python tools/hub_check.py [small]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
from oracle import c_oracle
import hdem_synth
small = len(sys.argv) > 1
if small:
    os.environ["HDEM_HUB_MIN_TILES"] = "1"
    cases = [(127, 189, 2), (127, 189, 1), (127, 189, 0), (333, 1100, 1)]
else:
    cases = [(1500, 1300, 0), (1500, 1300, 3), (1054, 1054, 0), (2048, 2048, 0), (4096, 4096, 0)]
for h, w, holes in cases:
    z = hdem_synth.synth_dem(h, w)
    if holes >= 1:
        z[h // 2, w // 2] = np.nan
    if holes >= 2:
        z[60:70, 60:64] = np.nan
    if holes >= 3:
        z[620:760, 300:500] = np.nan
    want = c_oracle.sinkfill_pflood(z)
    zd = B.DeviceRaster.from_host(z)
    out, st = B.sinkfill_dev(zd)
    got = out.to_host()
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    print(h, w, holes, "mismatches", int(bad.sum()), "visits/tile %.2f" % (st["tile_visits"] / max(st["tiles"], 1)), "rounds", st["rounds"], flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "got<want", int((got[bad] < want[bad]).sum()), "got>want", int((got[bad] > want[bad]).sum()),
              "nan mismatch", int((np.isnan(got[bad]) != np.isnan(want[bad])).sum()))
        for y, x in list(zip(ys, xs))[:12]:
            print("   ", y, x, "tile", (y - 1) // 62, (x - 1) // 62, "in-tile", (y - 1) % 62, (x - 1) % 62, got[y, x], want[y, x], z[y, x])
    out.free(); zd.free()
