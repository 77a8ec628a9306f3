"""Fill + D8 over small raster sizes, hub start forced on / off (exploration: where the hub
start begins to pay).  usage: python tools/small_sweep.py"""
import sys, os, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from hydrodem_amd import backend as B
    import hdem_synth
    ctx = B.context()
    for n in (256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6000, 8192):
        zd = B.DeviceRaster.from_host(hdem_synth.synth_dem(n, n)); wd = B.DeviceRaster.empty((n, n), np.float32); dd = B.DeviceRaster.empty((n, n), np.uint8)
        for _ in range(3):
            B.sinkfill_d8_dev(zd, out=wd, codes=dd)
        ctx.synchronize(); t = time.perf_counter()
        for _ in range(10):
            _, _, st = B.sinkfill_d8_dev(zd, out=wd, codes=dd)
        ctx.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
        print(f"  {n:5d}^2: {ms:7.3f} ms  {st['tile_visits']/max(st['tiles'],1):5.2f} visits/tile ({st['tiles']} tiles)", flush=True)
        for r in (zd, wd, dd): r.free()
else:
    for label, env in (("hub forced on", {"HDEM_HUB_MIN_TILES": "1", "HDEM_HUB_MIN_TILES_NESTED": "1000000"}),
                       ("hub on, nested too", {"HDEM_HUB_MIN_TILES": "1", "HDEM_HUB_MIN_TILES_NESTED": "16"}),
                       ("defaults", {}),
                       ("hub off", {"HDEM_FILL_HUB": "0"})):
        print(label, flush=True)
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, **env), stderr=subprocess.DEVNULL)
