"""The partitioned fill on virtual ranks of one GPU against the C oracle, with and without the
global hub start (exploration).  usage: python tools/vranks_check.py [world] [rows_per_rank] [cols]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrodem_amd import backend as B, partition as P
from oracle import c_oracle
import hdem_synth
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
holes = len(sys.argv) > 4
h = world * rows
z = hdem_synth.synth_dem(h, cols)
if holes:
    z[rows - 30:rows + 40, 500:700] = np.nan           # nodata across a seam
    z[h // 2 + 7, cols // 3] = np.nan
check = True
if h * cols <= 3e8:
    want = c_oracle.sinkfill_pflood(z)
    want_d8 = c_oracle.d8(want)
else:                                   # too large for the CPU oracle in a tool: the undivided GPU fill
    zd = B.DeviceRaster.from_host(z)
    wd, dd, _ = B.sinkfill_d8_dev(zd)
    want, want_d8 = wd.to_host(), dd.to_host()
    for r in (zd, wd, dd): r.free()
ghost = P.ghost_rows(world, h)
for hub in ((True,) if os.environ.get("HUB_ONLY") else (True, True, False, False)):
    def body(rank, comm):
        g0, g1, _, _ = P.local_range(rank, world, h, ghost)
        zt = torch.from_numpy(z[g0:g1]).cuda()
        codes = torch.empty(zt.shape, dtype=torch.uint8, device=zt.device)
        solver = P.HipLocalSolver(0, turn=comm.gpu_turn)
        w, info = P.sinkfill_distributed(zt, rank, world, solver, d8_out=codes, ghost=ghost, comm=comm, hub=hub)
        torch.cuda.synchronize()
        own = P.owned_slice(rank, world, ghost)
        out = w[own].cpu().numpy(), codes[own].cpu().numpy(), info, list(solver.timings)
        solver.ctx.close()
        return out
    t0 = time.time()
    got = P.ThreadWorld(world).run(body)
    dt = time.time() - t0
    bad = 0
    for rank, (w_own, d_own, info, _) in enumerate(got):
        r0, r1 = P.row_range(rank, world, h)
        if check:
            m = ~((w_own == want[r0:r1]) | (np.isnan(w_own) & np.isnan(want[r0:r1])))
            bad += int(m.sum()) + int((d_own != want_d8[r0:r1]).sum())
            if m.any():
                ys, xs = np.nonzero(m)
                print(f"   rank {rank}: {int(m.sum())} cells differ, local rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}, "
                      f"got<want {int((w_own[m] < want[r0:r1][m]).sum())} got>want {int((w_own[m] > want[r0:r1][m]).sum())}; first "
                      f"{[(int(y), int(x), float(w_own[y, x]), float(want[r0 + y, x])) for y, x in list(zip(ys, xs))[:4]]}", flush=True)
    print(f"hub={hub}: start {got[0][2]['start_values']}, mismatches {bad}, exchanges {got[0][2]['exchanges']}, "
          f"visits/rank {[g[2]['tile_visits'] for g in got]}, solves of rank 1 {got[min(1, world - 1)][2]['solves']}, wall {dt:.2f} s", flush=True)
    # critical path: the ranks' calls line up phase by phase (same sequence on every rank)
    n = min(len(g[3]) for g in got)
    crit = [(got[0][3][k][0], max(g[3][k][1] for g in got)) for k in range(n)]
    print("   critical path %.2f ms (GPU calls only, each with the GPU to itself): " % sum(c[1] for c in crit)
          + "  ".join(f"{l} {t:.2f}" for l, t in crit), flush=True)
