import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import oracle
from oracle import c_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
z = oracle.synth_dem(n, n)
want = c_oracle.sinkfill_pflood(z)
for flags in (B.FILL_SYNC_ONLY, 0):
    got, st = B.sinkfill(z, return_stats=True) if flags == 0 else (None, None)
    if flags:
        zd = B.DeviceRaster.from_host(z); wd = B.DeviceRaster.empty(z.shape, np.float32)
        _, st = B.sinkfill_dev(zd, out=wd, flags=flags); got = wd.to_host()
    bad = np.argwhere(got != want)
    print("flags", flags, "mismatch cells", len(bad), st)
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print(" rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
        ty, tx = (ys - 1) // 62, (xs - 1) // 62
        tiles = sorted(set(zip(ty.tolist(), tx.tolist())))
        print(" tiles", len(tiles), tiles[:20])
        for (y, x) in bad[:8]:
            print("  cell", y, x, "in-tile r,c", (y - 1) % 62 + 1, (x - 1) % 62 + 1, "got", got[y, x], "want", want[y, x], "z", z[y, x])
        print(" got>want:", (got[got != want] > want[got != want]).all())
