"""Time the Fourier destripe on an n x n striped DEM, per kernel (exploration).
usage: python tools/fourier_time.py [n] [reps]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = B.context()
dem = hdem_synth.synth_striped_dem(n, n)
d = B.DeviceRaster.from_host(dem); out = B.DeviceRaster.empty(dem.shape, np.float32)
names = {B.K_FFT: "rocFFT c2c", B.K_FOURIER_DETECT: "detect",
         B.K_FOURIER_MASK: "mask kernels", B.K_FOURIER_POINT: "pointwise"}
for rep in range(reps):
    ctx.profile(True); ctx.profile_reset()
    t = time.time(); B.fourier_destripe_dev(d, out=out); ctx.synchronize(); dt = time.time() - t
    line = ", ".join(f"{v} {ctx.profile_get(k)['ms']:.2f} ms/{ctx.profile_get(k)['launches']}" for k, v in names.items())
    print(f"destripe {n}^2: wall {dt*1e3:.1f} ms -> {n*n/dt/1e6:.0f} Mcells/s; {line}")
qn = (n // 2 - 10) ** 2
k = ctx.profile_get(B.K_FOURIER_DETECT); print(f"detect: {9*qn*k['launches']/k['ms']/1e6:.0f} GB/s algorithmic (9 B/cell), {k['ms']/k['launches']:.3f} ms per launch")
k = ctx.profile_get(B.K_FFT); print(f"fft: {16*n*n*k['launches']/k['ms']/1e6:.0f} GB/s at one read + one write of the complex64 array per transform")
res = out.to_host()
print("max |out - dem|", float(np.abs(res - dem).max()))
