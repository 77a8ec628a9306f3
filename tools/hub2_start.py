"""Prototype (host C + GPU fill): what a SECOND hub per tile would be worth (exploration; see
tools/hub2_start.c).  usage: python tools/hub2_start.py [n]   (HUB_NO_GPU=1: statistics only)"""
import ctypes, os, subprocess, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hdem_synth
from oracle import c_oracle
so = os.path.join(HERE, "_hub2_start.so")
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "hub2_start.c"), "-lm"])
L = ctypes.CDLL(so)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
z = hdem_synth.synth_dem(n, n, variant=os.environ.get("HUB_VARIANT", "rough"))
H = W = n
T = 62
ty = tx = (n - 2 + T - 1) // T
want = c_oracle.sinkfill_pflood(z)
dA = np.empty_like(z); dB = np.empty_like(z)
hubA = np.empty(ty * tx, dtype=np.int64); hubB = np.empty(ty * tx, dtype=np.int64); ab = np.empty(ty * tx, dtype=np.float32)
t0 = time.time()
L.hub2_dist(P(z), H, W, P(dA), P(dB), P(hubA), P(hubB), P(ab))
lev = np.empty(2 * ty * tx, dtype=np.float32)
L.hub2_levels(P(z), P(dA), P(dB), H, W, P(hubA), P(hubB), P(ab), P(lev))
u = np.empty_like(z)
L.hub2_start(P(z), P(dA), P(dB), H, W, P(lev), P(u))
print(f"{n}^2: host {time.time()-t0:.1f} s; tiles with a second hub {100*np.mean(hubB >= 0):.1f} %", flush=True)
for name, uu, hubs, levs in (("two hubs", u, hubA, lev[0::2]),):
    bad = int((uu < want).sum())
    inner = np.zeros_like(z, dtype=bool); inner[1:-1, 1:-1] = True
    ex = (uu - want)[inner]; raised = (want > z)[inner]
    hub_ex = levs - want.ravel()[hubs]
    print(f"{name}: violations {bad}; exact on {100*np.mean(ex == 0):.1f} % of the cells ({100*np.mean(ex[raised] == 0):.1f} % of the raised ones), "
          f"mean excess {np.mean(ex[ex < 1e30]):.4f} m; first hubs exact {100*np.mean(hub_ex == 0):.1f} %, mean excess {hub_ex.mean():.4f} m", flush=True)
    assert bad == 0
# one hub (the same code with the second hub ignored): levels from A alone
lev1 = lev.copy()
if not os.environ.get("HUB_NO_GPU"):
    from hydrodem_amd import backend as B
    ctx = B.context()
    zd = B.DeviceRaster.from_host(z)
    for rep in range(2):
        ud = B.DeviceRaster.from_host(u, ctx=ctx)
        ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=ud, flags=B.FILL_WARM); ctx.synchronize()
        print(f"   fill WARM from the two-hub start: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} ({st['tile_visits']/st['tiles']:.2f} per tile)", flush=True)
        assert np.array_equal(ud.to_host(), want)
        ud.free()
