"""What one round of the row-block loop costs on the host side when nothing is left to do
(exploration): the seam words + vote read back and a correcting solve that waits, against the
deferred form (words stay on the device, the solve is enqueued, the vote is looked at behind it).
No transport in either: the swap and the all-reduce are the same in both forms.
usage: python tools/defer_cost.py [rows] [cols]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hdem_synth
from hydrodem_amd import backend as B, partition as P

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
z = torch.from_numpy(hdem_synth.synth_dem(rows, cols)).cuda()
w = torch.empty_like(z)
solver = P.HipLocalSolver()
flags = B.FILL_INIT | B.FILL_GHOST_BOTTOM
solver.fill(z, w, 0.0, flags | B.FILL_NO_VERIFY)
seam = P._Seam(torch, w, False, True)
seam.recv_bot.copy_(w[-1])
act = B.FILL_WARM | B.FILL_RESUME | B.FILL_NO_VERIFY
N = 50


def words(pending):
    word = seam.word[:3]
    if pending >= 0:
        word[0] = int(pending > 0)
    word[1:].zero_()
    word[2] = (seam.recv_bot.view(torch.int32) != w[-1].view(torch.int32)).any()
    w[-1].copy_(seam.recv_bot)
    return word.max().clamp(max=1).reshape(1)


for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N):
        busy = words(0)
        out = torch.cat([busy, seam.word[1:]]).cpu()
        # (the old loop skips the solve when nothing changed; a round where a ghost row did
        # change pays the waiting call -- timed here with nothing to do in it)
        solver.fill(z, w, 0.0, act | B.FILL_ACT_BOTTOM)
    torch.cuda.synchronize(); t_old = (time.perf_counter() - t) / N
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N):
        busy = words(0)
        out = torch.cat([busy, seam.word[1:]]).cpu()
    torch.cuda.synchronize(); t_vote = (time.perf_counter() - t) / N
    pending = 0
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N):
        solver.seam_apply(w, None, seam.recv_bot, pending, seam.word)
        seam.vote_host.copy_(seam.word[3:], non_blocking=True)
        seam.event.record()
        solver.fill_deferred(z, w, 0.0, act | B.FILL_ACT_BOTTOM, seam.word)
        pending = -1
        seam.event.synchronize()
    torch.cuda.synchronize(); t_new = (time.perf_counter() - t) / N
    print(f"per round, nothing to do: read-back + waiting solve {t_old*1e6:.0f} us (read-back alone "
          f"{t_vote*1e6:.0f} us), deferred {t_new*1e6:.0f} us", flush=True)
