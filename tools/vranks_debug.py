"""debug: validity of the partition's hub start values (u = max(d, level) >= fill) per rank."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrodem_amd import backend as B, partition as P
import hdem_synth
world, rows, cols = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
h = world * rows
z = hdem_synth.synth_dem(h, cols)
zd = B.DeviceRaster.from_host(z)
wd, _ = B.sinkfill_dev(zd)
want = wd.to_host(); wd.free(); zd.free()
ghost = P.ghost_rows(world, h)
T = 62
def body(rank, comm):
    g0, g1, top, bottom = P.local_range(rank, world, h, ghost)
    zt = torch.from_numpy(z[g0:g1]).cuda()
    w = torch.empty_like(zt)
    solver = P.HipLocalSolver(0, turn=comm.gpu_turn)
    flags = (B.FILL_GHOST_TOP if top else 0) | (B.FILL_GHOST_BOTTOM if bottom else 0)
    levels = P.hub_start(zt, w, comm, solver, flags, ghost)
    torch.cuda.synchronize()
    d = w.cpu().numpy(); lev = levels.cpu().numpy()
    hh, ww = d.shape
    ty, tx = (hh - 2 + T - 1) // T, (ww - 2 + T - 1) // T
    L = lev[1::2, 1::2]
    Lc = np.repeat(np.repeat(L, T, axis=0), T, axis=1)[:hh - 2, :ww - 2]
    Lc = np.where(Lc >= 3e38, np.inf, Lc)
    u = d.copy()
    inner = u[1:-1, 1:-1]
    u[1:-1, 1:-1] = np.where(np.isnan(Lc) | np.isnan(inner), inner, np.maximum(inner, Lc))
    wl = want[g0:g1]
    bad = u < wl
    bad[:, 0] = bad[:, -1] = False
    if not top: bad[0] = False
    if not bottom: bad[-1] = False
    msg = f"rank {rank}: local {hh} rows, {int(bad.sum())} start values below the fill"
    if bad.any():
        ys, xs = np.nonzero(bad)
        tiles = sorted(set(zip(((ys - 1) // T).tolist(), ((xs - 1) // T).tolist())))
        msg += f"; rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; tiles {tiles[:10]}"
        y, x = ys[0], xs[0]
        j, i = (y - 1) // T, (x - 1) // T
        msg += f"; first ({y},{x}) u {u[y,x]} d {d[y,x]} level {L[j,i] if 0 <= j < ty else None} want {wl[y,x]} z {z[g0+y,x]}"
        msg += f"; hi {(hh - 2 * ghost - 1) // T}; levels col {i}: rows {max(j-2,0)}..{min(j+2,ty-1)} {L[max(j-2,0):j+3, i]}; raster around node: {lev[2*j:2*j+3, 2*i:2*i+3]}"
    solver.ctx.close()
    return msg
for m in P.ThreadWorld(world).run(body):
    print(m[:1500], flush=True)
