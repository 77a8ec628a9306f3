import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import oracle
from oracle import c_oracle
n = 1024
z = oracle.synth_dem(n, n)
want = c_oracle.sinkfill_pflood(z)
zd = B.DeviceRaster.from_host(z); wd = B.DeviceRaster.empty(z.shape, np.float32)
_, st = B.sinkfill_dev(zd, out=wd, flags=B.FILL_SYNC_ONLY); got = wd.to_host()
print(st)
# cells violating the fixed point: one more sweep changes them
new, ch = oracle.sinkfill_sweep(z, got)
viol = np.argwhere(new != got)
print("fixed-point violations:", len(viol))
for (y, x) in viol[:10]:
    r, c = (y - 1) % 62 + 1, (x - 1) % 62 + 1
    print(f"cell ({y},{x}) tile ({(y-1)//62},{(x-1)//62}) r={r} c={c} got={got[y,x]:.5f} new={new[y,x]:.5f} want={want[y,x]:.5f} z={z[y,x]:.5f}")
    print("   3x3 got:\n", got[y-1:y+2, x-1:x+2])
    print("   3x3 want:\n", want[y-1:y+2, x-1:x+2])
