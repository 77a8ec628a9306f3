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
        ghost = P.ghost_rows(world, H)                     # one tile row of overlap
        g0, g1, _, _ = P.local_range(rank, world, H, ghost)
        z = oracle.synth_dem(H, W, row0=g0, rows=g1 - g0)
        zt = torch.from_numpy(z).cuda()
        solver = P.HipLocalSolver(0)
        d = torch.empty(zt.shape, dtype=torch.uint8, device=zt.device)
        w, info = P.sinkfill_distributed(zt, rank, world, solver, d8_out=d, ghost=ghost)
        torch.cuda.synchronize()
        own_ = P.owned_slice(rank, world, ghost)
        assert torch.equal(d[own_], P.d8_distributed(w, solver)[own_])
        torch.cuda.synchronize()
        own = own_
        # the one-exchange stencils: groves x3 (21-row halo), box mean (1 row)
        r0, r1 = P.row_range(rank, world, H)
        img = torch.from_numpy(oracle.synth_dem(H, W, pits=False)[r0:r1].copy()).cuda()
        mask = torch.from_numpy(oracle.synth_groves(H, W)[r0:r1].copy()).cuda()
        gr = P.groves_distributed(img, mask, rank, world, solver)
        bm = P.boxmean_distributed(gr, rank, world, solver)
        torch.cuda.synchronize()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), w=w.cpu().numpy()[own],
                 d=d.cpu().numpy()[own], exchanges=info["exchanges"],
                 groves=gr.cpu().numpy(), boxmean=bm.cpu().numpy())
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
    # groves: the kernel sums float32 offsets from a per-strip reference level, so a
    # different block origin may round the last bit differently: <= 1e-4 m (the bar of
    # SURVEY 8d), not bit-equal.  Box mean + round of the same input: bit for bit.
    from hydrodem_amd import backend
    img, mask = oracle.synth_dem(H, W, pits=False), oracle.synth_groves(H, W)
    whole = backend.groves_dev(backend.DeviceRaster.from_host(img),
                               backend.DeviceRaster.from_host(mask)).to_host()
    got_g = np.concatenate([p["groves"] for p in parts])
    assert got_g.shape == whole.shape and np.abs(got_g - whole).max() <= 1e-4
    assert np.array_equal(np.concatenate([p["boxmean"] for p in parts]),
                          backend.boxmean3_dev(backend.DeviceRaster.from_host(got_g)).to_host())


def test_bench_starts_its_own_ranks(built):
    """`python bench.py --gpus 2` with no launcher around it -- the shape of command the
    driver issues -- spawns its ranks itself (here both on cuda:0 over gloo:
    HDEM_REHEARSE=1) and prints rank 0's one JSON line with the per-rank detail."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HDEM_REHEARSE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size",
                          "2048", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["metric"].endswith("4096x2048 float32 DEM")
    assert rec["config"]["halo_exchanges"] >= 2 and rec["config"]["tiles"] > 0
    assert rec["config"]["visits_unchanged"] is not None
    assert len(rec["per_rank"]["fill_async_ms_per_step"]) == 2
    assert all(v > 0 for v in rec["per_rank"]["tile_visits_per_step"])
    # a rank that dies takes the whole run down with a non-zero status
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size",
                          "-5"], env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def _rccl_worker(rank, world, port, H, W, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from hydrodem_amd import partition as P
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        z = oracle.synth_dem(H, W)
        zt = torch.from_numpy(z).cuda()
        solver = P.HipLocalSolver(0)
        comm = P.DistComm()
        assert not comm.gloo
        d = torch.empty(zt.shape, dtype=torch.uint8, device=zt.device)
        w, info = P.sinkfill_distributed(zt, rank, world, solver, d8_out=d, comm=comm)
        # the two collectives of the schedule on device tensors, through RCCL itself
        word = torch.tensor([rank + 3], dtype=torch.int32, device=zt.device)
        assert int(comm.all_reduce_max(word).item()) == world + 2
        parts = comm.all_gather(torch.arange(4, device=zt.device) + rank)
        assert len(parts) == world and parts[0].tolist() == [0, 1, 2, 3]
        torch.cuda.synchronize()
        np.savez(os.path.join(outdir, "r0.npz"), w=w.cpu().numpy(), d=d.cpu().numpy(),
                 solves=info["solves"])
    finally:
        dist.destroy_process_group()


def test_rccl_backend_world_of_one(tmp_path, built):
    """The `nccl` (= RCCL) backend on the one GPU of the box: a world of one rank has no
    neighbour to swap with, but the process group, the device MAX-reduce of the vote and
    the all-gather of the coarse start run through RCCL, on the library's stream order."""
    H, W = 900, 1100
    mp.spawn(_rccl_worker, args=(1, _free_port(), H, W, str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "r0.npz")
    want = c_oracle.sinkfill_pflood(oracle.synth_dem(H, W))
    assert np.array_equal(got["w"], want) and np.array_equal(got["d"], c_oracle.d8(want))


@pytest.mark.parametrize("world,rows,cols,holes", [(4, 1024, 2100, False), (3, 700, 1000, True),
                                                   (8, 512, 1300, False)])
def test_partition_hub_start_bounds_the_fill(built, world, rows, cols, holes):
    """The start values of the partitioned fill -- ONE hub graph over all ranks
    (partition.hub_start): every rank's d, the seam rows of d swapped into the ghost rows, the
    ranks' hub rasters stacked and filled, levels handed back -- are upper bounds of the fill,
    cell by cell, on every rank: u = max(d, level of the cell's tile) >= the C oracle, ghost
    rows included.  Widths that leave a partial last tile column, nodata across a seam."""
    from hydrodem_amd import backend, partition as P
    h, t = world * rows, 62
    z = oracle.synth_dem(h, cols)
    if holes:
        z[rows - 20:rows + 30, 300:420] = np.nan
        z[2 * rows + 3, 77] = np.nan
    want = c_oracle.sinkfill_pflood(z)
    ghost = P.ghost_rows(world, h)

    def body(rank, comm):
        g0, g1, top, bottom = P.local_range(rank, world, h, ghost)
        zt = torch.from_numpy(z[g0:g1]).cuda()
        w = torch.empty_like(zt)
        solver = P.HipLocalSolver(0, turn=comm.gpu_turn)
        flags = (backend.FILL_GHOST_TOP if top else 0) | (backend.FILL_GHOST_BOTTOM if bottom else 0)
        levels = P.hub_start(zt, w, comm, solver, flags, ghost)
        torch.cuda.synchronize()
        d, lev = w.cpu().numpy(), levels.cpu().numpy()
        solver.ctx.close()
        hh, ww = d.shape
        per_cell = np.repeat(np.repeat(lev[1::2, 1::2], t, axis=0), t, axis=1)[:hh - 2, :ww - 2]
        per_cell = np.where(per_cell >= 3e38, np.inf, per_cell)
        u = d.copy()
        inner = u[1:-1, 1:-1]
        u[1:-1, 1:-1] = np.where(np.isnan(per_cell) | np.isnan(inner), inner,
                                 np.maximum(inner, per_cell))
        low = u < want[g0:g1]                       # (NaN compares false: nodata stays out)
        low[:, 0] = low[:, -1] = False              # raster ring columns: written by the fill
        if not top:
            low[0] = False
        if not bottom:
            low[-1] = False
        return int(low.sum()), float(np.mean(u[1:-1, 1:-1] == want[g0:g1][1:-1, 1:-1]))

    got = P.ThreadWorld(world).run(body)
    assert [g[0] for g in got] == [0] * world, f"start values below the fill: {got}"
    assert min(g[1] for g in got) > 0.3             # and tight: a third of the cells exact


@pytest.mark.parametrize("world,rows,cols,holes,hub,eps", [(4, 640, 1500, False, True, 0.0),
                                                           (3, 500, 900, True, False, 0.0),
                                                           (3, 400, 700, True, True, 1e-3)])
def test_partition_deferred_loop_matches_the_waiting_one(built, monkeypatch, world, rows, cols,
                                                         holes, hub, eps):
    """The exchange loop that keeps its decisions on the device (HDEM_FILL_DEFER: seam words,
    solves enqueued without a host wait, vote looked at behind them) against the loop that
    reads every vote back first: same bits as the C oracle from both, on every rank, and the
    counters of the deferred solves arrive with the verifying call.  Starts far from the truth
    (hub = False: +inf ghost rows) so that several exchanges really correct something."""
    from hydrodem_amd import partition as P
    h = world * rows
    z = oracle.synth_dem(h, cols, variant="rough")
    if holes:
        z[rows - 9:rows + 12, 200:260] = np.nan
    want = c_oracle.sinkfill_pflood(z, eps)
    ghost = P.ghost_rows(world, h)

    def run(defer):
        monkeypatch.setenv("HDEM_PARTITION_DEFER", "1" if defer else "0")

        def body(rank, comm):
            g0, g1, _, _ = P.local_range(rank, world, h, ghost)
            zt = torch.from_numpy(z[g0:g1]).cuda()
            solver = P.HipLocalSolver(0, turn=comm.gpu_turn)
            w, info = P.sinkfill_distributed(zt, rank, world, solver, eps=eps, ghost=ghost,
                                             comm=comm, hub=hub, coarse_block=0)
            out = w.cpu().numpy()
            solver.ctx.close()
            return out, info, (g0, g1)

        return P.ThreadWorld(world).run(body)

    for defer in (True, False):
        got = run(defer)
        for w, info, (g0, g1) in got:
            assert np.array_equal(w, want[g0:g1], equal_nan=True), (defer, info)
            assert (info["deferred_solves"] > 0) == defer
            assert info["deferred_solves"] in (0, info["exchanges"])
        if defer:
            # every rank's tally holds the visits of its deferred solves (reported by the
            # verifying call): more than the first solve and the verifying passes alone
            for _, info, _ in got:
                first = info["solves"][0][1]
                assert info["tile_visits"] >= first
