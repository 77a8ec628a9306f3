"""What the host forms of the mirrored filters cost at n x n around their kernels
(exploration).  usage: python tools/host_filters_time.py [n]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hydrodem_amd as hd
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dem = hdem_synth.synth_dem(n, n)
hs = hdem_synth.synth_hsheds(min(n, 8192), min(n, 8192)) if n <= 8192 else np.round(hdem_synth.synth_dem(n, n, pits=False))
cases = [
    ("SinkFill + D8 (HydroConditioning)", lambda: hd.HydroConditioning().apply(dem)),
    ("PostProcessingFinal f32", lambda: hd.PostProcessingFinal().apply(dem)),
    ("QuadraticFilter(15)", lambda: hd.QuadraticFilter(window_size=15).apply(dem)),
    ("DetectApplyFourier", lambda: hd.DetectApplyFourier().apply(dem)),
    ("MajorityFilter(11)", lambda: hd.MajorityFilter(window_size=11).apply(hs)),
    ("LagoonsDetection", lambda: hd.LagoonsDetection().apply(hs.copy())),
    ("CorrectNANValues", lambda: hd.CorrectNANValues().apply(hs.copy())),
]
for name, fn in cases:
    ts = []
    for rep in range(3):
        t = time.time(); r = fn(); ts.append(time.time() - t); del r
    print(f"{name:36s} {ts[0]*1e3:8.1f} {ts[1]*1e3:8.1f} {ts[2]*1e3:8.1f} ms")
