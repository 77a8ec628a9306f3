"""Exploratory timing of the device-resident operators (not part of the test
suite or the bench contract).  python tools/explore.py [sizes...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth

def main():
    sizes = [int(s) for s in sys.argv[1:]] or [4096]
    ctx = B.context()
    for n in sizes:
        t = time.time(); z = hdem_synth.synth_dem(n, n); tg = time.time() - t
        zd = B.DeviceRaster.from_host(z)
        wd = B.DeviceRaster.empty(z.shape, np.float32)
        dd = B.DeviceRaster.empty(z.shape, np.uint8)
        flags = int(os.environ.get("FILL_FLAGS", "0"))
        for rep in range(2):
            ctx.profile(True); ctx.profile_reset()
            ctx.synchronize(); t = time.time()
            _, st = B.sinkfill_dev(zd, out=wd, flags=flags)
            ctx.synchronize(); tf = time.time() - t
            t = time.time(); B.d8_dev(wd, out=dd); ctx.synchronize(); td = time.time() - t
            kt = ctx.profile_get(B.K_FILL_TILE); ki = ctx.profile_get(B.K_FILL_INIT)
            k8 = ctx.profile_get(B.K_D8); ks = ctx.profile_get(B.K_COPY)
            cells = n * n
            print(f"n={n} gen={tg:.1f}s fill={tf*1e3:.2f}ms d8={td*1e3:.3f}ms "
                  f"-> {cells/(tf+td)/1e6:.0f} Mcells/s | rounds={st['rounds']} "
                  f"visits={st['tile_visits']} ({st['tile_visits']/st['tiles']:.2f}/tile) iters={st['iterations']} unchanged={st['visits_unchanged']} requeued={st['visits_requeued']} "
                  f"tile_kernel={kt['ms']:.2f}ms launches={kt['launches']} "
                  f"GB/s={12*kt['units']/max(kt['ms'],1e-9)/1e6:.0f} "
                  f"init={ki['ms']:.3f}ms scan={ks['ms']:.3f}ms d8k={k8['ms']:.3f}ms "
                  f"d8GB/s={5*cells/max(k8['ms'],1e-9)/1e6:.0f}", flush=True)
        ctx.profile(False)
        # other kernels
        g = B.DeviceRaster.from_host(hdem_synth.synth_groves(n, n))
        od = B.DeviceRaster.empty(z.shape, np.float32)
        for rep in range(2):
            ctx.profile(True); ctx.profile_reset()
            B.boxmean3_dev(zd, True, out=od)
            B.groves_dev(zd, g, iterations=3, out=od)
            kb = ctx.profile_get(B.K_BOXMEAN); kg = ctx.profile_get(B.K_GROVES)
            cells = n * n
            print(f"  boxmean={kb['ms']:.3f}ms GB/s={8*cells/kb['ms']/1e6:.0f} | "
                  f"groves x3={kg['ms']:.3f}ms GB/s={27*cells/kg['ms']/1e6:.0f}", flush=True)
        ctx.profile(False)
        for r in (zd, wd, dd, g, od): r.free()

main()
