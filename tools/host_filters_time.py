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
groves = hdem_synth.synth_groves(n, n)
cases = [
    ("SinkFill + D8 (HydroConditioning)", lambda: hd.HydroConditioning().apply(dem)),
    ("PostProcessingFinal f32", lambda: hd.PostProcessingFinal().apply(dem)),
    ("QuadraticFilter(15)", lambda: hd.QuadraticFilter(window_size=15).apply(dem)),
    ("DetectApplyFourier", lambda: hd.DetectApplyFourier().apply(dem)),
    ("MajorityFilter(11)", lambda: hd.MajorityFilter(window_size=11).apply(hs)),
    ("LagoonsDetection", lambda: hd.LagoonsDetection().apply(hs.copy())),
    ("CorrectNANValues", lambda: hd.CorrectNANValues().apply(hs.copy())),
    ("BinaryClosing (uint8 class)", lambda: hd.BinaryClosing().apply(groves)),
    ("BinaryErosion(2)", lambda: hd.BinaryErosion(iterations=2).apply(groves)),
    ("TidyingLagoons", lambda: hd.TidyingLagoons().apply(hs)),
    ("ExpandFilter(7)", lambda: hd.ExpandFilter(window_size=7).apply(groves)),
    ("GreyDilation(7,7)", lambda: hd.GreyDilation(size=(7, 7)).apply(hs)),
    ("D8FlowDirection", lambda: hd.D8FlowDirection().apply(dem)),
    ("SinkFill", lambda: hd.SinkFill().apply(dem)),
    ("Convolve 3x3 ones", lambda: hd.Convolve(weights=np.ones((3, 3))).apply(dem)),
    ("Around", lambda: hd.Around().apply(dem)),
]
for name, fn in cases:
    ts = []
    for rep in range(3):
        t = time.time(); r = fn(); ts.append(time.time() - t); del r
    print(f"{name:36s} {ts[0]*1e3:8.1f} {ts[1]*1e3:8.1f} {ts[2]*1e3:8.1f} ms")
