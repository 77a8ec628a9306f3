"""Prototype (host NumPy + GPU fill): start values from an exact-ish coarse graph on a lattice
of bands along the tile seams (exploration).

Bands of K rows (columns) lie on every tile seam, so every cell of every tile's window ring is
a band cell.  Nodes: the lowest cell of every band crossing.  Edge between neighbouring nodes:
the minimax cost of the best path between the two node cells that stays in the band and only
steps forward (a row recurrence: cost(r, c) = max(z, min3 cost(r-1..r+1, c-1))) -- a restricted
family of paths, hence a valid upper bound on the true minimax distance, but tight: a band K
wide lets the path dodge the noise.  The coarse problem (nodes + edges) is an 8-connected
node-weighted raster of (2 ny + 1) x (2 nx + 1) cells and is filled exactly; a band cell's start
value is min over its segment's two nodes of max(level(node), cost(cell -> node)).  Interior
cells start at +big: the tile's first visit relaxes them against its (tight) ring.
usage: python tools/lattice_start.py [n] [K]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B
from oracle import c_oracle
import hdem_synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = 62
BIG = np.float32(3.0e38)
z = hdem_synth.synth_dem(n, n)
H = W = n
t00 = time.time()

# band j covers rows lo(j) .. hi(j) (inclusive) around the seam rows 62 j, 62 j + 1; the first
# band starts at the raster's row 0, the last one ends at row n - 1
ntiles = (n - 2 + T - 1) // T
bands = []
for j in range(ntiles):
    lo, hi = max(T * j - (K // 2 - 1), 0), min(T * j + K // 2, n - 1)
    bands.append([lo, hi])
if n - 1 - bands[-1][1] > 1:
    bands.append([max(n - 1 - K // 2, bands[-1][1] + 1), n - 1])      # a band of its own on the last rows
else:
    bands[-1][1] = n - 1
nb = len(bands)
# nodes: lowest cell of every crossing (on the raster ring for the outermost bands)
node_r = np.zeros((nb, nb), dtype=np.int64); node_c = np.zeros((nb, nb), dtype=np.int64)
for j, (rlo, rhi) in enumerate(bands):
    for i, (clo, chi) in enumerate(bands):
        sq = z[rlo:rhi + 1, clo:chi + 1].copy()
        if j == 0: sq[1:, :] = np.inf                      # outermost nodes sit on the raster ring
        if j == nb - 1: sq[:-1, :] = np.inf
        if i == 0: sq[:, 1:] = np.inf
        if i == nb - 1: sq[:, :-1] = np.inf
        a = np.argmin(sq); node_r[j, i] = rlo + a // sq.shape[1]; node_c[j, i] = clo + a % sq.shape[1]
node_z = z[node_r, node_c]

def band_dp(zz, bands, node_r, node_c):
    """Horizontal bands of raster zz: forward recurrences between consecutive nodes of every
    band, all bands at once (vectorised over bands, a Python loop over the columns).
    Returns (edge[j][i] between node i and i+1, [(costA, costB)] per band)."""
    nb = len(bands)
    k = max(hi - lo + 1 for lo, hi in bands)
    wd = zz.shape[1]
    strips = np.full((nb, k, wd), np.inf, dtype=np.float32)
    for j, (lo, hi) in enumerate(bands):
        strips[j, :hi - lo + 1] = zz[lo:hi + 1]
    rr = node_r - np.array([lo for lo, _ in bands])[:, None]            # node rows inside the band
    is_node = np.zeros((nb, wd), dtype=np.int64) - 1                     # node index at (band, col)
    for j in range(nb):
        is_node[j, node_c[j]] = np.arange(node_c.shape[1])
    edges = np.full((nb, node_c.shape[1] - 1), np.inf, dtype=np.float32)
    ca = np.full((nb, k, wd), np.inf, dtype=np.float32)
    cb = np.full((nb, k, wd), np.inf, dtype=np.float32)
    inf_col = np.full((nb, 1), np.inf, dtype=np.float32)
    ar = np.arange(nb)

    def sweep(cols, out, record):
        prev = np.full((nb, k), np.inf, dtype=np.float32)
        for c in cols:
            m = np.minimum(np.minimum(np.concatenate((inf_col, prev[:, :-1]), axis=1), prev),
                           np.concatenate((prev[:, 1:], inf_col), axis=1))
            cur = np.maximum(strips[:, :, c], m)
            hit = is_node[:, c] >= 0
            if hit.any():
                jj = ar[hit]; ii = is_node[hit, c]; r = rr[jj, ii]
                if record:
                    ok = ii > 0
                    edges[jj[ok], ii[ok] - 1] = cur[jj[ok], r[ok]]       # reached the next node
                cur[jj] = np.inf
                cur[jj, r] = strips[jj, r, c]                             # restart at the node
            out[:, :, c] = cur
            prev = cur

    sweep(range(wd), ca, True)
    sweep(range(wd - 1, -1, -1), cb, False)
    costs = [(ca[j, :hi - lo + 1], cb[j, :hi - lo + 1]) for j, (lo, hi) in enumerate(bands)]
    return edges, costs

eh, costs_h = band_dp(z, bands, node_r, node_c)
zt = np.ascontiguousarray(z.T)
ev_t, costs_v = band_dp(zt, bands, node_c.T.copy(), node_r.T.copy())   # vertical bands = horizontal of z^T
ev = ev_t.T                                                            # ev[j][i]: node (j,i) -> (j+1,i)
print(f"band recurrences {time.time()-t00:.1f} s; nodes {nb}x{nb}; mean edge - max(node z) = "
      f"{np.mean(eh - np.maximum(node_z[:, :-1], node_z[:, 1:])):.3f} m")

# coarse raster: nodes at (2j, 2i), horizontal edges at (2j, 2i+1), vertical at (2j+1, 2i)
cr = np.full((2 * nb - 1, 2 * nb - 1), BIG, dtype=np.float32)
cr[0::2, 0::2] = node_z
cr[0::2, 1::2] = eh
cr[1::2, 0::2] = ev
lev = c_oracle.sinkfill_pflood(cr)[0::2, 0::2]                          # level of every node

# start values
u = np.full((n, n), BIG, dtype=np.float32)
seg = lambda cc_row: np.searchsorted(cc_row, np.arange(n), side="right") - 1   # node index left of / at col
segr = lambda cc_row: np.searchsorted(cc_row, np.arange(n), side="left")       # node index right of / at col
for j, (rlo, rhi) in enumerate(bands):
    ca, cb = costs_h[j]
    left = np.clip(seg(node_c[j]), 0, nb - 1); right = np.clip(segr(node_c[j]), 0, nb - 1)
    # columns left of the first node / right of the last: only one node in reach
    ua = np.maximum(ca, lev[j][left][None, :]); ub = np.maximum(cb, lev[j][right][None, :])
    ub[:, np.arange(n) > node_c[j][-1]] = np.inf
    u[rlo:rhi + 1] = np.minimum(u[rlo:rhi + 1], np.minimum(ua, ub))
for i, (clo, chi) in enumerate(bands):
    ca, cb = costs_v[i]
    up = np.clip(seg(node_r[:, i]), 0, nb - 1); down = np.clip(segr(node_r[:, i]), 0, nb - 1)
    ua = np.maximum(ca, lev[:, i][up][None, :]); ub = np.maximum(cb, lev[:, i][down][None, :])
    ub[:, np.arange(n) > node_r[:, i][-1]] = np.inf
    u[:, clo:chi + 1] = np.minimum(u[:, clo:chi + 1], np.minimum(ua, ub).T)
u = np.maximum(u, z)
u[0] = z[0]; u[-1] = z[-1]; u[:, 0] = z[:, 0]; u[:, -1] = z[:, -1]
print(f"start values made in {time.time()-t00:.1f} s (host prototype)")

ctx = B.context()
zd = B.DeviceRaster.from_host(z)
wd = B.DeviceRaster.empty(z.shape, np.float32)
for rep in range(2):
    ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=wd); ctx.synchronize()
    print(f"ordinary fill: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} unchanged {st['visits_unchanged']}")
want = wd.to_host()
onband = u < BIG
bad = (u < want)
print(f"band cells {100*onband.mean():.1f} % of the raster; bound violated on {int(bad.sum())} cells")
ex = (u - want)[onband]
print(f"on the bands: exact on {100*(ex == 0).mean():.1f} %, mean excess {ex.mean():.4f} m, "
      f"p99 {np.quantile(ex, 0.99):.3f}, max {ex.max():.2f}; nodes: mean excess "
      f"{np.mean(lev - want[node_r, node_c]):.4f} m, exact {100*np.mean(lev == want[node_r, node_c]):.1f} %")
assert not bad.any()
for rep in range(3):
    ud = B.DeviceRaster.from_host(u, ctx=ctx)
    ctx.synchronize(); t = time.time(); _, st = B.sinkfill_dev(zd, out=ud, flags=B.FILL_WARM); ctx.synchronize()
    print(f"fill WARM from the lattice start: {1e3*(time.time()-t):.2f} ms, visits {st['tile_visits']} "
          f"({st['tile_visits']/st['tiles']:.2f} per tile) unchanged {st['visits_unchanged']} flat {st['visits_flat']}")
    assert np.array_equal(ud.to_host(), want)
    ud.free()
