"""Sink fill + D8 over raster sizes, variants and epsilon (exploration / evidence for
profiles/): one JSON record per case.  usage: python tools/sweep_sizes.py [out.json]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/size_sweep.json"
ctx = B.context()
rows = []
for n, variant, eps in ((2048, "rough", 0.0), (4096, "rough", 0.0), (8192, "rough", 0.0),
                        (16384, "rough", 0.0), (16384, "srtm", 0.0), (16384, "rough", 1e-3),
                        (32768, "rough", 0.0)):
    z = hdem_synth.synth_dem(n, n, variant=variant)
    zd = B.DeviceRaster.from_host(z, ctx=ctx)
    del z
    wd = B.DeviceRaster.empty((n, n), np.float32, ctx)
    dd = B.DeviceRaster.empty((n, n), np.uint8, ctx)
    st = {}
    def step():
        st.update(B.sinkfill_d8_dev(zd, eps=eps, out=wd, codes=dd)[2])
    step(); step(); ctx.synchronize()
    reps = 10 if n <= 16384 else 4
    t = time.perf_counter()
    for _ in range(reps):
        step()
    ctx.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    rec = {"size": n, "variant": variant, "eps": eps, "ms_per_step": ms, "Gcells_per_s": n * n / ms / 1e6,
           "visits_per_tile": st["tile_visits"] / max(st["tiles"], 1), "visits_flat": st["visits_flat"],
           "visits_unchanged": st["visits_unchanged"], "useful_frac": 8.0 * n * n / ms / 1e6 / 8000.0}
    rows.append(rec)
    print(rec, flush=True)
    for r in (zd, wd, dd):
        r.free()
json.dump(rows, open(out, "w"), indent=1)
