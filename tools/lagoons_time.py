"""Per-kernel times of the lagoon chain on an n x n raster (exploration).
usage (under rocprofv3 --kernel-trace --stats for the split): python tools/lagoons_time.py [n]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = B.context()
hs = np.round(hdem_synth.synth_dem(n, n)); hs[::97, ::89] = -32768.0
d = B.DeviceRaster.from_host(hs)
for rep in range(3):
    ctx.profile(True); ctx.profile_reset()
    t = time.time()
    for r in B.lagoons_detection_dev(d): r.free()
    ctx.synchronize(); dt = time.time() - t
    print(f"lagoons {n}^2: wall {dt*1e3:.1f} ms, majority {ctx.profile_get(B.K_MAJORITY)['ms']:.2f} ms, other kernels {ctx.profile_get(B.K_LAGOON)['ms']:.2f} ms, expand {ctx.profile_get(B.K_FOURIER_MASK)['ms']:.2f} ms")
