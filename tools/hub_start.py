"""Prototype (host C + GPU fill): start values from a hub graph (exploration).

Every 62 x 62 tile gets a hub (its lowest cell) and d(c) = the minimax cost of the best path
from c to the hub inside the tile.  Hubs of neighbouring tiles are joined by the cheapest seam
crossing, max(d(a), d(b)) over adjacent cells a | b; the hub graph is filled exactly as a
(2 ty + 1) x (2 tx + 1) node-weighted raster.  u(c) = max(d(c), level(hub)) is an upper bound of
the fill: c -> hub inside the tile, hub -> raster ring along the graph.
usage: python tools/hub_start.py [n] [iters ...]   (iters 0 = exact d; k = k rounds of 4 scans)
       HUB_NO_GPU=1: statistics against the C oracle only"""
import ctypes, os, subprocess, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hdem_synth
from oracle import c_oracle

so = os.path.join(HERE, "_hub_start.so")
T = int(os.environ.get("HUB_T", "62"))                # edge of the hub tiles (the fill's own stay 62)
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", f"-DFT={T}", "-o", so,
                       os.path.join(HERE, "hub_start.c"), "-lm"])
L = ctypes.CDLL(so)
fp = ctypes.POINTER(ctypes.c_float); ip = ctypes.POINTER(ctypes.c_int64)
P = lambda a: a.ctypes.data_as(fp)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters_list = [int(a) for a in sys.argv[2:]] or [0]
variant = os.environ.get("HUB_VARIANT", "rough")
z = hdem_synth.synth_dem(n, n, variant=variant)
H = W = n
ty = tx = (n - 2 + T - 1) // T
t0 = time.time()
want = c_oracle.sinkfill_pflood(z)
print(f"{n}^2 {variant}: oracle {time.time()-t0:.1f} s; raised cells {100*np.mean(want > z):.1f} %", flush=True)
use_gpu = not os.environ.get("HUB_NO_GPU")
if use_gpu:
    from hydrodem_amd import backend as B
    ctx = B.context()
    zd = B.DeviceRaster.from_host(z)
    wd = B.DeviceRaster.empty(z.shape, np.float32)
    for rep in range(2):
        ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=wd); ctx.synchronize()
        print(f"ordinary fill: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} "
              f"({st['tile_visits']/st['tiles']:.2f} per tile) unchanged {st['visits_unchanged']} flat {st['visits_flat']}", flush=True)
    assert np.array_equal(wd.to_host(), want)
    ud = B.DeviceRaster.from_host(want, ctx=ctx)
    ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=ud, flags=B.FILL_WARM); ctx.synchronize()
    print(f"fill WARM from the answer: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']}", flush=True)
    ud.free()

if os.environ.get("HUB_MARGIN"):
    # paths to the hub may leave the tile by this many cells (exact distances over the grown tile)
    L.hub_set_margin(int(os.environ["HUB_MARGIN"]))
for iters in iters_list:
    t0 = time.time()
    d = np.empty_like(z); hub = np.empty(ty * tx, dtype=np.int64)
    L.hub_dist(P(z), H, W, iters, P(d), hub.ctypes.data_as(ip))
    cr = np.empty((2 * ty + 1, 2 * tx + 1), dtype=np.float32)
    L.hub_edges(P(z), P(d), H, W, hub.ctypes.data_as(ip), P(cr))
    lev = np.ascontiguousarray(c_oracle.sinkfill_pflood(cr)[1::2, 1::2])
    if os.environ.get("HUB_EXACT_LEVELS"):
        # what a graph with exact hub levels would be worth: the oracle's fill at every hub
        lev = np.ascontiguousarray(want.ravel()[hub].reshape(lev.shape).astype(np.float32))
    u = np.empty_like(z)
    L.hub_start(P(z), P(d), H, W, P(lev), P(u))
    bad = int((u < want).sum())
    inner = np.zeros_like(z, dtype=bool); inner[1:-1, 1:-1] = True
    ex = (u - want)[inner]
    raised = (want > z)[inner]
    hub_ex = lev.ravel() - want.ravel()[hub]
    print(f"iters {iters}: host {time.time()-t0:.1f} s; violations {bad}; exact on {100*np.mean(ex == 0):.1f} % of the cells "
          f"({100*np.mean(ex[raised] == 0):.1f} % of the raised ones), unreached {100*np.mean(u[inner] >= 3e38):.2f} %, mean excess "
          f"{np.mean(ex[ex < 1e30]):.4f} m, p90 {np.quantile(ex, 0.9):.3f}, p99 {np.quantile(ex, 0.99):.3f}; hubs exact "
          f"{100*np.mean(hub_ex == 0):.1f} %, mean excess {hub_ex.mean():.4f} m", flush=True)
    # per tile: share of tiles whose every cell is exact
    tex = (u - want)[1:1 + (ty - 1) * T, 1:1 + (tx - 1) * T].reshape(ty - 1, T, tx - 1, T).max(axis=(1, 3))
    print(f"   tiles exact everywhere {100*np.mean(tex == 0):.1f} %, within 1 mm {100*np.mean(tex < 1e-3):.1f} %, "
          f"within 0.1 m {100*np.mean(tex < 0.1):.1f} %", flush=True)
    assert bad == 0
    if use_gpu:
        for rep in range(2):
            ud = B.DeviceRaster.from_host(u, ctx=ctx)
            ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=ud, flags=B.FILL_WARM); ctx.synchronize()
            print(f"   fill WARM from the hub start: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} "
                  f"({st['tile_visits']/st['tiles']:.2f} per tile) unchanged {st['visits_unchanged']} flat {st['visits_flat']}", flush=True)
            assert np.array_equal(ud.to_host(), want)
            ud.free()
