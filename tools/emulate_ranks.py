"""Emulate the N-rank row-block sink fill on ONE GPU, rank by rank, to predict the
multi-GPU critical path: per phase (time slice or full local solve, then an exchange)
the slowest rank counts, since ranks run in parallel on a real node.  Exploration only.
usage: python tools/emulate_ranks.py N [rows_per_rank] [cols] [slice_us (0 = unsliced)] [coarse block (0 = +inf ghosts)] [first_only (1: only the first solve is sliced)]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B, partition as P
import hdem_synth

N = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
W = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
SLICE = int(sys.argv[4]) if len(sys.argv) > 4 else P.DEFAULT_SLICE_US
COARSE = int(sys.argv[5]) if len(sys.argv) > 5 else P.COARSE_BLOCK
FIRST_ONLY = len(sys.argv) > 6 and sys.argv[6] == "1"
H = N * S
blocks = []
for r in range(N):
    g0, g1, top, bot = P.local_range(r, N, H)
    zt = torch.from_numpy(hdem_synth.synth_dem(H, W, row0=g0, rows=g1 - g0)).cuda()
    blocks.append({"z": zt, "w": torch.empty_like(zt), "top": top, "bot": bot, "pending": 0,
                   "solver": P.HipLocalSolver(0, slice_us=SLICE, own_context=True)})
def timed_fill(b, flags):
    if not (flags & B.FILL_WARM) and COARSE:
        b["solver"].set_coarse_start(filled, COARSE, b["row_map"])
    torch.cuda.synchronize(); t = time.perf_counter()
    sliced = SLICE > 0 and not (FIRST_ONLY and (flags & B.FILL_WARM))
    v, lowered, b["pending"] = b["solver"].fill(b["z"], b["w"], 0.0, flags, sliced)
    torch.cuda.synchronize(); return time.perf_counter() - t, v, lowered
crit = 0.0; rounds = []; visits = 0
given = 0
if COARSE:
    # what partition.coarse_ghost_guess does: every rank coarsens its rows (parallel), then
    # solves the stacked coarse raster (redundantly: counts once on the critical path)
    for warm in range(2):      # time the second pass: the first pays module loads and mallocs
        torch.cuda.synchronize(); t0 = time.perf_counter()
        parts = [b["solver"].blockmax(b["z"][P.owned_slice(r, N)].contiguous(), COARSE) for r, b in enumerate(blocks)]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        coarse = torch.cat(parts).contiguous(); filled = torch.empty_like(coarse)
        cv = blocks[0]["solver"].fill(coarse, filled, 0.0, B.FILL_INIT)[0]
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"coarse {tuple(coarse.shape)}: blockmax {1e3*(t1-t0)/N:.3f} ms/rank, fill {1e3*(t2-t1):.3f} ms, {cv} visits")
    crit += (t1 - t0) / N + (t2 - t1); rounds.append(("coarse", (t1 - t0) / N + (t2 - t1), (t1 - t0) + N * (t2 - t1)))
    first = 0
    for r, b in enumerate(blocks):
        n = parts[r].shape[0]
        owned = b["z"][P.owned_slice(r, N)].shape[0]
        own = first + torch.arange(owned, dtype=torch.int32) // COARSE
        rm = torch.cat(([torch.tensor([first - 1], dtype=torch.int32)] if b["top"] else []) + [own] +
                       ([torch.tensor([first + n], dtype=torch.int32)] if b["bot"] else []))
        b["row_map"] = rm.to(torch.int32).cuda()
        first += n
    given = 0
ts = []
for b in blocks:
    f = given | B.FILL_INIT | B.FILL_NO_VERIFY | (B.FILL_GHOST_TOP if b["top"] else 0) | (B.FILL_GHOST_BOTTOM if b["bot"] else 0)
    t = timed_fill(b, f); ts.append(t[0]); visits += t[1]
crit += max(ts); rounds.append(("init", max(ts), sum(ts)))
while True:
    sends = [(b["w"][1].clone(), b["w"][-2].clone()) for b in blocks]
    ts = []
    for r, b in enumerate(blocks):
        flags = B.FILL_WARM | B.FILL_NO_VERIFY | B.FILL_RESUME
        ch = b["pending"] > 0
        if b["top"] and not torch.equal(sends[r - 1][1], b["w"][0]):
            b["w"][0].copy_(sends[r - 1][1]); flags |= B.FILL_ACT_TOP; ch = True
        if b["bot"] and not torch.equal(sends[r + 1][0], b["w"][-1]):
            b["w"][-1].copy_(sends[r + 1][0]); flags |= B.FILL_ACT_BOTTOM; ch = True
        if ch:
            t = timed_fill(b, flags); ts.append(t[0]); visits += t[1]
    if not ts:
        ts = [timed_fill(b, B.FILL_WARM | B.FILL_SYNC_ONLY) for b in blocks]
        visits += sum(t[1] for t in ts)
        crit += max(t[0] for t in ts); rounds.append(("verify", max(t[0] for t in ts), sum(t[0] for t in ts)))
        if not any(t[2] for t in ts):
            break
        continue
    crit += max(ts); rounds.append(("slice", max(ts), sum(ts)))
print(f"N={N} rows/rank={S} cols={W} slice={SLICE}us coarse={COARSE}: {len(rounds)} phases, critical path {crit*1e3:.2f} ms "
      f"(+ ~{0.2*len(rounds):.1f} ms exchange latency), {visits} tile visits = {visits/(N*((S+61)//62)*((W+59)//62)):.1f}/tile")
for k, (name, mx, sm) in enumerate(rounds):
    print(f"  {k:2d} {name:8s} slowest rank {mx*1e3:7.3f} ms  (sum over ranks {sm*1e3:7.3f} ms)")
