"""What start values from directional monotone-path sweeps are worth (exploration).

D_N(y, x) = max(z(y, x), min(D_N(y-1, x-1), D_N(y-1, x), D_N(y-1, x+1))), D_N = z on the raster
ring: the minimax cost of the best path to the border that only ever steps north (N, NW, NE)
-- a restricted family of paths, so an upper bound of the fill; likewise S, W, E.  U = min of
the four.  The script makes U on the host (NumPy), checks U >= fill, reports how tight it is,
and runs the GPU fill WARM from U against the ordinary call.
usage: python tools/directional_start.py [n] [strip]   (strip: confine paths to column / row
strips of that width, 0 = unconfined)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
import hdem_synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
strip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
z = hdem_synth.synth_dem(n, n)


def sweep_down(z, strip):
    """north -> south sweep over rows; returns D (float32)."""
    h, w = z.shape
    d = np.empty_like(z)
    d[0] = z[0]
    inf = np.float32(np.inf)
    wall = None
    if strip:
        wall = np.zeros(w, dtype=bool)
        wall[::strip] = True                      # first column of every strip: no step across
    for y in range(1, h):
        p = d[y - 1]
        left = np.concatenate(([inf], p[:-1]))
        right = np.concatenate((p[1:], [inf]))
        if strip:
            left = np.where(wall, inf, left)                       # from x-1 into a strip start
            right = np.where(np.roll(wall, -1), inf, right)        # from x+1 across a strip start
        m = np.minimum(np.minimum(left, p), right)
        row = np.maximum(z[y], m)
        row[0] = z[y, 0]
        row[-1] = z[y, -1]
        d[y] = row
    d[-1] = z[-1]
    return d


t = time.time()
dn = sweep_down(z, strip)
ds = sweep_down(z[::-1], strip)[::-1]
dw = sweep_down(np.ascontiguousarray(z.T), strip).T
de = sweep_down(np.ascontiguousarray(z.T[::-1]), strip)[::-1].T
u = np.minimum(np.minimum(dn, ds), np.minimum(dw, de)).astype(np.float32)
print(f"host sweeps {time.time()-t:.1f} s")
ctx = B.context()
zd = B.DeviceRaster.from_host(z)
wd = B.DeviceRaster.empty(z.shape, np.float32)
for rep in range(2):
    ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=wd); ctx.synchronize()
    print(f"ordinary fill: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} unchanged {st['visits_unchanged']}")
want = wd.to_host()
assert (u >= want).all(), "directional bound is not an upper bound"
ex = u - want
print(f"bound: exact on {100*(ex == 0).mean():.1f} % of the cells, mean excess {ex.mean():.4f} m, "
      f"p99 {np.quantile(ex, 0.99):.3f} m, max {ex.max():.2f} m")
for name, d in (("N", dn), ("S", ds), ("W", dw), ("E", de)):
    print(f"  {name}: exact on {100*(d == want).mean():.1f} %")
for rep in range(3):
    ud = B.DeviceRaster.from_host(u, ctx=ctx)
    ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=ud, flags=B.FILL_WARM); ctx.synchronize()
    print(f"fill WARM from U: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} unchanged {st['visits_unchanged']} flat {st['visits_flat']}")
    assert np.array_equal(ud.to_host(), want)
    ud.free()
