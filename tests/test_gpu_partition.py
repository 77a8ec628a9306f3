"""
N > 1 rehearsal on a one-GPU box: two ranks share cuda:0 and exchange over gloo
(device rows staged through the host), the local solver is the HIP library.
Result must equal the unpartitioned oracle bit for bit.  (RCCL itself needs one
GPU per rank; the exchange logic is identical.)
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, H, W, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from hydrodem_amd import partition as P
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g0, g1, _, _ = P.local_range(rank, world, H)
        z = oracle.synth_dem(H, W, row0=g0, rows=g1 - g0)
        zt = torch.from_numpy(z).cuda()
        solver = P.HipLocalSolver(0)
        w, info = P.sinkfill_distributed(zt, rank, world, solver)
        d = P.d8_distributed(w, solver)
        torch.cuda.synchronize()
        own = P.owned_slice(rank, world)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), w=w.cpu().numpy()[own],
                 d=d.cpu().numpy()[own], exchanges=info["exchanges"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 1400, 900), (3, 1000, 700)])
def test_two_rank_rehearsal_on_one_gpu(tmp_path, built, world, H, W):
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    z = oracle.synth_dem(H, W)
    want_w = c_oracle.sinkfill_pflood(z)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert np.array_equal(np.concatenate([p["w"] for p in parts]), want_w)
    assert np.array_equal(np.concatenate([p["d"] for p in parts]), c_oracle.d8(want_w))
    assert all(int(p["exchanges"]) >= 2 for p in parts)
