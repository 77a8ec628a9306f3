"""Would a thicker ghost zone (g rows solved redundantly by both neighbours, only the outermost
row pinned) shorten the tail of halo-exchange rounds of the row-block sink fill?  Virtual
ranks on one GPU, as tools/emulate_ranks.py.  Exploration only.
usage: python tools/emulate_overlap.py N rows_per_rank cols g"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydrodem_amd import backend as B, partition as P
import hdem_synth

N, S, W, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
H, BLK = N * S, 16
blocks = []
for r in range(N):
    r0, r1 = r * S, (r + 1) * S
    lo, hi = max(r0 - G, 0), min(r1 + G, H)
    zt = torch.from_numpy(hdem_synth.synth_dem(H, W, row0=lo, rows=hi - lo)).cuda()
    blocks.append({"z": zt, "w": torch.empty_like(zt), "lo": lo, "hi": hi, "r0": r0, "r1": r1,
                   "top": r > 0, "bot": r < N - 1, "solver": P.HipLocalSolver(0, own_context=True)})
# global coarse raster (every rank's owned rows, stacked) and per-block row maps
parts = [b["solver"].blockmax(b["z"][b["r0"] - b["lo"]: b["r1"] - b["lo"]].contiguous(), BLK) for b in blocks]
coarse = torch.cat(parts).contiguous(); filled = torch.empty_like(coarse)
blocks[0]["solver"].fill(coarse, filled, 0.0, B.FILL_INIT | B.FILL_NO_VERIFY)
rows_per = parts[0].shape[0]
for b in blocks:
    y = torch.arange(b["lo"], b["hi"])
    b["row_map"] = ((y // S) * rows_per + (y % S) // BLK).to(torch.int32).cuda()
def timed(b, flags, coarse_start=False):
    if coarse_start:
        b["solver"].set_coarse_start(filled, BLK, b["row_map"])
    torch.cuda.synchronize(); t = time.perf_counter()
    v, lowered, _ = b["solver"].fill(b["z"], b["w"], 0.0, flags)
    torch.cuda.synchronize(); return time.perf_counter() - t, v, lowered
crit, phases = 0.0, []
ts = [timed(b, B.FILL_INIT | B.FILL_NO_VERIFY | (B.FILL_GHOST_TOP if b["top"] else 0) | (B.FILL_GHOST_BOTTOM if b["bot"] else 0), True)[0] for b in blocks]
crit += max(ts); phases.append(("init", max(ts)))
while True:
    # what each block's pinned rows should be now: the neighbour's value of that global row
    new = []
    for r, b in enumerate(blocks):
        t_ = blocks[r - 1]["w"][b["lo"] - blocks[r - 1]["lo"]].clone() if b["top"] else None
        b_ = blocks[r + 1]["w"][b["hi"] - 1 - blocks[r + 1]["lo"]].clone() if b["bot"] else None
        new.append((t_, b_))
    ts = []
    for (t_, b_), b in zip(new, blocks):
        flags = B.FILL_WARM | B.FILL_NO_VERIFY | B.FILL_RESUME
        if t_ is not None and not torch.equal(t_, b["w"][0]):
            b["w"][0].copy_(t_); flags |= B.FILL_ACT_TOP
        if b_ is not None and not torch.equal(b_, b["w"][-1]):
            b["w"][-1].copy_(b_); flags |= B.FILL_ACT_BOTTOM
        if flags & (B.FILL_ACT_TOP | B.FILL_ACT_BOTTOM):
            ts.append(timed(b, flags)[0])
    if not ts:
        break
    crit += max(ts); phases.append(("round", max(ts)))
ts = [timed(b, B.FILL_WARM | B.FILL_SYNC_ONLY) for b in blocks]
crit += max(t[0] for t in ts); phases.append(("verify", max(t[0] for t in ts)))
assert not any(t[2] for t in ts)
# the overlap rows agree between neighbours
for r in range(N - 1):
    a, b = blocks[r], blocks[r + 1]
    lo, hi = b["lo"], a["hi"]
    assert torch.equal(a["w"][lo - a["lo"]: hi - a["lo"]], b["w"][: hi - lo])
print(f"N={N} g={G}: {len(phases)} phases, critical path {crit*1e3:.2f} ms, exchange rounds {len(phases)-2}")
print("  " + "  ".join(f"{n} {t*1e3:.2f}" for n, t in phases))
