"""Does a coarse-solve start value for the WHOLE raster (not just ghost rows) speed the
single-GPU fill up?  W0 = max(z, fill(blockmax(z, b)) expanded), ring = z, then a WARM
solve.  Exploration only.  usage: python tools/coarse_start.py [n]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HDEM_FILL_TRACE", "1")
from hydrodem_amd import backend as B, partition as P
import hdem_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variant = sys.argv[2] if len(sys.argv) > 2 else "rough"
solver = P.HipLocalSolver(0)
z = torch.from_numpy(hdem_synth.synth_dem(n, n, variant=variant)).cuda()
ref = torch.empty_like(z)
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); v = solver.fill(z, ref, 0.0, B.FILL_INIT); torch.cuda.synchronize()
    print(f"plain fill: {1e3*(time.perf_counter()-t):.2f} ms, {v[0]} visits")
for b in (32, 16):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        coarse = solver.blockmax(z, b); filled = torch.empty_like(coarse)
        solver.fill(coarse, filled, 0.0, B.FILL_INIT | B.FILL_NO_VERIFY)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        w = filled.repeat_interleave(b, 0).repeat_interleave(b, 1)[:n, :n].contiguous()
        w = torch.maximum(w, z)
        w[0], w[-1], w[:, 0], w[:, -1] = z[0], z[-1], z[:, 0], z[:, -1]
        torch.cuda.synchronize(); t2 = time.perf_counter()
        v = solver.fill(z, w, 0.0, B.FILL_WARM)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"b={b}: coarse {1e3*(t1-t0):.2f} ms, expand(torch) {1e3*(t2-t1):.2f} ms, warm solve {1e3*(t3-t2):.2f} ms, {v[0]} visits, exact={torch.equal(w, ref)}")
