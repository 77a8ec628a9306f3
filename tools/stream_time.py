"""PCIe-inclusive time of groves x3 on an n x n host raster: the whole-array host entry
point against the band stream (exploration).  usage: python tools/stream_time.py [n] [band_rows]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B, streaming as S
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
band = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
img, mask = hdem_synth.synth_dem(n, n, pits=False), hdem_synth.synth_groves(n, n)
for rep in range(2):
    t = time.time(); a = B.groves(img, mask, iterations=3); dt = time.time() - t
    print(f"whole-array host call: {dt*1e3:.1f} ms -> {n*n/dt/1e6:.0f} Mcells/s")
out = np.empty_like(img)
for depth in (1, 2, 3, 4, 6, 8):
    with S.BandStream(img.shape, band_rows=band, depth=depth, **S.groves_op(3)) as bs:
        for rep in range(2):
            t = time.time(); bs.run([img, mask], out); dt = time.time() - t
        print(f"band stream depth {depth} ({band} rows): {dt*1e3:.1f} ms -> {n*n/dt/1e6:.0f} Mcells/s, max diff {np.abs(out-a).max():.2e}")
# the copies alone, for scale
d = B.DeviceRaster.empty(img.shape, np.float32)
t = time.time(); B.context().check(B.context().lib.hdem_memcpy_h2d(B.context().handle, d.ptr, img.ctypes.data, img.nbytes)); dt = time.time() - t
print(f"pageable H2D of the raster alone: {dt*1e3:.1f} ms = {img.nbytes/dt/1e9:.1f} GB/s")
