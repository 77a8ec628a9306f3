"""Gradient fill (eps > 0) from the eps = 0 fill plus a margin (exploration, host-side prototype):
W0 = fill(eps = 0); u = W0 + K (eps + ulp) on the free cells; fill(eps) WARM from u.  u bounds the
gradient fill when no run of "gentle" steps is longer than K; cells with a drop of more than the
margin to a neighbour settle on their first visit whatever the order.
usage: python tools/eps_two_stage.py [n] [variant]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variant = sys.argv[2] if len(sys.argv) > 2 else "rough"
ctx = B.context()
z = hdem_synth.synth_dem(n, n, variant=variant)
zd = B.DeviceRaster.from_host(z)
w0 = B.DeviceRaster.empty((n, n), np.float32)
wd = B.DeviceRaster.empty((n, n), np.float32)
for eps in (1e-3, 1e-4):
    ctx.synchronize(); t = time.perf_counter(); _, st = B.sinkfill_dev(zd, eps=eps, out=wd); ctx.synchronize()
    t_plain = time.perf_counter() - t
    want = wd.to_host()
    print(f"eps {eps}: plain {1e3*t_plain:.2f} ms, {st['tile_visits']/st['tiles']:.2f} visits/tile; "
          f"max excess over the eps=0 fill: ", end="", flush=True)
    ctx.synchronize(); t = time.perf_counter(); _, st0 = B.sinkfill_dev(zd, out=w0, flags=B.FILL_INIT | B.FILL_NO_VERIFY); ctx.synchronize()
    t0 = time.perf_counter() - t
    base = w0.to_host()
    ex = want - base
    print(f"{ex.max():.5f} m = {ex.max()/eps:.0f} steps, cells with excess {100*np.mean(ex > 0):.2f} %", flush=True)
    for K in (32, 128, 512):
        ulp = np.spacing(np.abs(base).astype(np.float32) * 2 + 1)
        u = (base + np.float32(K) * (np.float32(eps) + ulp) + ulp).astype(np.float32)
        u[0, :] = base[0, :]; u[-1, :] = base[-1, :]; u[:, 0] = base[:, 0]; u[:, -1] = base[:, -1]
        ok = bool((u >= want).all())
        ud = B.DeviceRaster.from_host(u, ctx=ctx)
        ctx.synchronize(); t = time.perf_counter(); _, st1 = B.sinkfill_dev(zd, eps=eps, out=ud, flags=B.FILL_WARM); ctx.synchronize()
        t1 = time.perf_counter() - t
        got = ud.to_host(); ud.free()
        print(f"   K {K}: bound holds {ok}; eps=0 stage {1e3*t0:.2f} ms ({st0['tile_visits']/st0['tiles']:.2f}/tile) + warm eps stage "
              f"{1e3*t1:.2f} ms ({st1['tile_visits']/st1['tiles']:.2f}/tile, rounds {st1['rounds']}); equal to the plain result: {np.array_equal(got, want)}", flush=True)
