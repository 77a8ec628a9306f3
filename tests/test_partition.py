"""
CPU suite, part 4: the N > 1 path.  The row-block partition + halo exchange of
`hydrodem_amd/partition.py` runs here under `gloo` with world sizes 2 and 3;
the local block solver is injected, so on CPU it is the NumPy oracle (tests
may use the oracle; the product path injects the HIP solver).  The result
must equal the unpartitioned oracle bit for bit -- the sink-fill fixed point
does not depend on the update order.
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
from hydrodem_amd import backend, partition as P


class NumpyLocalSolver:
    """Oracle-backed stand-in for HipLocalSolver (same two methods).  A "time slice"
    is ``slice_sweeps`` Jacobi sweeps; an unfinished slice reports one pending tile."""

    def __init__(self, slice_sweeps=3):
        self.slice_sweeps = slice_sweeps
        self.start = None

    def set_coarse_start(self, filled, block, row_map):
        self.start = (filled.numpy(), block, row_map.numpy())

    def fill(self, z, w, eps, flags, sliced=False, d8=None):
        zn, wn = z.numpy(), w.numpy()
        if not flags & backend.FILL_WARM:
            w0 = oracle.sinkfill_init(zn)
            level = np.inf
            if self.start is not None and eps == 0:
                filled, block, row_map = self.start
                level = np.repeat(filled[row_map], block, axis=1)[:, :zn.shape[1]]
                level = np.where(level >= 3e38, np.inf, level)
                free = np.isinf(w0)
                w0[free] = np.maximum(level, zn)[free]
                level = level[[0, -1]][:, 1:-1]
            self.start = None
            given = bool(flags & backend.FILL_GHOST_GIVEN)
            lv = level if np.ndim(level) else np.full((2, zn.shape[1] - 2), np.inf)
            if flags & backend.FILL_GHOST_TOP:
                start = np.maximum(wn[0, 1:-1], zn[0, 1:-1]) if given else \
                    np.maximum(lv[0], zn[0, 1:-1])
                w0[0, 1:-1] = np.where(np.isnan(zn[0, 1:-1]), zn[0, 1:-1], start)
            if flags & backend.FILL_GHOST_BOTTOM:
                start = np.maximum(wn[-1, 1:-1], zn[-1, 1:-1]) if given else \
                    np.maximum(lv[1], zn[-1, 1:-1])
                w0[-1, 1:-1] = np.where(np.isnan(zn[-1, 1:-1]), zn[-1, 1:-1], start)
            wn[:] = w0
        sweeps = 0
        while True:
            new, changed = oracle.sinkfill_sweep(zn, wn, eps)
            wn[:] = new
            sweeps += 1
            if changed == 0:
                if d8 is not None:
                    d8.numpy()[:] = oracle.d8_flow_direction(wn)
                return sweeps, sweeps > 1, 0
            if sliced and flags & backend.FILL_NO_VERIFY and sweeps >= self.slice_sweeps:
                return sweeps, True, 1

    def d8(self, w, out):
        out.numpy()[:] = oracle.d8_flow_direction(w.numpy())

    def groves(self, img, mask, window_size, threshold, iterations):
        return torch.from_numpy(oracle.groves_exact64(
            img.numpy(), mask.numpy(), iterations, window_size, threshold)[0].astype(np.float32))

    def boxmean(self, x, do_round):
        f = oracle.boxmean3_round if do_round else oracle.boxmean3
        return torch.from_numpy(f(x.numpy()))

    def blockmax(self, z, block):
        zn = np.where(np.isnan(z.numpy()), np.finfo(np.float32).max, z.numpy())
        ch, cw = -(-zn.shape[0] // block), -(-zn.shape[1] // block)
        pad = np.full((ch * block, cw * block), -np.inf, dtype=np.float32)
        pad[:zn.shape[0], :zn.shape[1]] = zn
        return torch.from_numpy(pad.reshape(ch, block, cw, block).max(axis=(1, 3)))


class DeferringNumpySolver(NumpyLocalSolver):
    """The same stand-in with the two calls of the deferred exchange loop
    (``HipLocalSolver.seam_apply`` / ``fill_deferred``): seam words in a host tensor, the
    correcting solve "enqueued" = run at once, nothing returned."""

    can_defer = True

    def __init__(self, slice_sweeps=3):
        super().__init__(slice_sweeps)
        self.deferred_calls = self.deferred_idle = 0

    def seam_apply(self, w, recv_top, recv_bot, pending, words):
        wn, k = w.numpy(), words.numpy()
        if pending >= 0:
            k[0] = int(pending > 0)
        k[1] = k[2] = 0
        for row, recv, slot in ((0, recv_top, 1), (wn.shape[0] - 1, recv_bot, 2)):
            if recv is None:
                continue
            r = recv.numpy()
            k[slot] = int(not np.array_equal(r.view(np.int32), wn[row].view(np.int32)))
            wn[row] = r
        k[3] = int(k[:3].max() > 0)

    def fill_deferred(self, z, w, eps, flags, words):
        assert flags & backend.FILL_WARM and flags & backend.FILL_RESUME
        k = words.numpy()
        self.deferred_calls += 1
        if not k[:3].any():                            # (the library's launches leave at once)
            self.deferred_idle += 1
            return
        _, _, pending = self.fill(z, w, eps, flags, sliced=True)
        k[0] = int(pending > 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, H, W, eps, variant, nodata, outdir, coarse_block=4, ghost=1,
            deferring=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ghost = P.ghost_rows(world, H, ghost)
        g0, g1, _, _ = P.local_range(rank, world, H, ghost)
        z = oracle.synth_dem(H, W, row0=g0, rows=g1 - g0, variant=variant)
        if nodata:
            full = oracle.synth_dem(H, W, variant=variant)
            full[H // 2 - 3:H // 2 + 3, 10:20] = np.nan        # straddles a seam for world=2
            z = full[g0:g1].copy()
        zt = torch.from_numpy(z)
        solver = DeferringNumpySolver() if deferring else NumpyLocalSolver()
        d = torch.empty(zt.shape, dtype=torch.uint8)
        w, info = P.sinkfill_distributed(zt, rank, world, solver, eps=eps,
                                         coarse_block=coarse_block, d8_out=d, ghost=ghost)
        assert (info["deferred_solves"] > 0) == deferring
        if deferring:
            # one deferred solve per exchange, and the last of them -- behind the vote that
            # said "all at rest" -- found nothing to do
            assert solver.deferred_calls == info["exchanges"] and solver.deferred_idle >= 1
        assert torch.equal(d[P.owned_slice(rank, world, ghost)],
                           P.d8_distributed(w, solver)[P.owned_slice(rank, world, ghost)])
        own = P.owned_slice(rank, world, ghost)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), w=w.numpy()[own], d=d.numpy()[own],
                 exchanges=info["exchanges"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("block", [4, 8, 32])
@pytest.mark.parametrize("nodata", [False, True])
def test_coarse_fill_bounds_the_fine_fill_from_above(block, nodata):
    """The property the ghost-row start values rest on: the fill of the block-maximum
    raster, expanded back, is >= the fill of the raster, cell by cell (walls where a
    block holds nodata)."""
    z = oracle.synth_dem(150, 203)
    if nodata:
        z[40:44, 100:180] = np.nan
        z[120, 7] = np.nan
    fine = c_oracle.sinkfill_pflood(z)
    coarse = NumpyLocalSolver().blockmax(torch.from_numpy(z), block).numpy()
    assert coarse.shape == (-(-150 // block), -(-203 // block))
    bound = c_oracle.sinkfill_pflood(coarse)
    up = np.repeat(np.repeat(bound, block, axis=0), block, axis=1)[:150, :203]
    ok = ~np.isnan(fine)
    assert np.all(up[ok] >= fine[ok])
    assert np.isfinite(up).all()


@pytest.mark.parametrize("world,H,W,eps,variant,nodata,coarse_block,ghost", [
    (2, 96, 80, 0.0, "rough", False, 4, 7),        # 7 rows of overlap, one of them pinned
    (3, 99, 70, 0.0, "rough", True, 4, 62),        # asks for 62, gets what the blocks allow (32)
    (2, 64, 48, 1e-3, "rough", False, 4, 5),
    (3, 75, 70, 0.0, "srtm", False, 8, 3),
])
def test_partitioned_fill_with_overlap_rows(tmp_path, world, H, W, eps, variant, nodata,
                                            coarse_block, ghost):
    test_partitioned_fill_and_d8_equal_unpartitioned(tmp_path, world, H, W, eps, variant, nodata,
                                                     coarse_block, ghost)


@pytest.mark.parametrize("world,H,W,eps,variant,nodata,coarse_block", [
    (2, 96, 80, 0.0, "rough", False, 4),
    (2, 101, 64, 0.0, "srtm", False, 4),
    (3, 90, 70, 0.0, "rough", False, 4),
    (2, 64, 48, 1e-3, "rough", False, 4),         # gradient: no coarse solve either
    (2, 80, 60, 0.0, "rough", True, 4),
    (3, 100, 90, 0.0, "rough", True, 4),
    (2, 96, 80, 0.0, "rough", False, 0),          # ghost rows start at +inf (no coarse solve)
    (3, 75, 70, 0.0, "srtm", False, 8),
])
def test_partitioned_fill_and_d8_equal_unpartitioned(tmp_path, world, H, W, eps, variant, nodata,
                                                     coarse_block, ghost=1, deferring=False):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, eps, variant, nodata, str(tmp_path),
                            coarse_block, ghost, deferring), nprocs=world, join=True)
    z = oracle.synth_dem(H, W, variant=variant)
    if nodata:
        z[H // 2 - 3:H // 2 + 3, 10:20] = np.nan
    want_w = c_oracle.sinkfill_pflood(z, eps=eps)
    want_d = c_oracle.d8(want_w)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    got_w = np.concatenate([p["w"] for p in parts])
    got_d = np.concatenate([p["d"] for p in parts])
    assert got_w.shape == z.shape
    assert np.array_equal(np.nan_to_num(got_w, nan=-1), np.nan_to_num(want_w, nan=-1))
    assert np.array_equal(got_d, want_d)
    assert all(int(p["exchanges"]) >= 2 for p in parts)     # information did cross the seam


@pytest.mark.parametrize("world,H,W,eps,variant,nodata,coarse_block,ghost", [
    (2, 96, 80, 0.0, "rough", False, 4, 1),
    (3, 100, 90, 0.0, "rough", True, 0, 1),       # +inf ghost rows: several rounds of corrections
    (2, 64, 48, 1e-3, "rough", False, 4, 1),
    (3, 130, 70, 0.0, "srtm", False, 4, 8),       # overlap rows
])
def test_partitioned_fill_with_the_deferred_exchange_loop(tmp_path, world, H, W, eps, variant,
                                                          nodata, coarse_block, ghost):
    """The exchange loop that does not come back to the host between an exchange and the solve
    behind it (seam words, deferred solves, the vote looked at afterwards), over gloo between
    real processes: a solver with the two extra calls takes it, and the result is the oracle's."""
    test_partitioned_fill_and_d8_equal_unpartitioned(tmp_path, world, H, W, eps, variant, nodata,
                                                     coarse_block, ghost, deferring=True)


def _stencil_worker(rank, world, port, H, W, iters, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r0, r1 = P.row_range(rank, world, H)
        img = torch.from_numpy(oracle.synth_dem(H, W, pits=False)[r0:r1].copy())
        mask = torch.from_numpy(oracle.synth_groves(H, W)[r0:r1].copy())
        solver = NumpyLocalSolver()
        gr = P.groves_distributed(img, mask, rank, world, solver, iterations=iters)
        bm = P.boxmean_distributed(gr, rank, world, solver)
        np.savez(os.path.join(outdir, f"s{rank}.npz"), groves=gr.numpy(), boxmean=bm.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,iters", [(2, 120, 64, 3), (3, 150, 50, 2)])
def test_partitioned_groves_and_boxmean_equal_unpartitioned(tmp_path, world, H, W, iters):
    """One halo exchange (iters * 7 rows, then 1 row), overlap recomputed: the owned
    rows equal the unpartitioned result of the same solver."""
    mp.spawn(_stencil_worker, args=(world, _free_port(), H, W, iters, str(tmp_path)),
             nprocs=world, join=True)
    solver = NumpyLocalSolver()
    img = torch.from_numpy(oracle.synth_dem(H, W, pits=False))
    mask = torch.from_numpy(oracle.synth_groves(H, W))
    want_g = solver.groves(img, mask, 15, 1.5, iters)
    want_b = solver.boxmean(want_g, True)
    parts = [np.load(tmp_path / f"s{r}.npz") for r in range(world)]
    assert np.array_equal(np.concatenate([p["groves"] for p in parts]), want_g.numpy())
    assert np.array_equal(np.concatenate([p["boxmean"] for p in parts]), want_b.numpy())


def test_halo_exchange_rejects_blocks_thinner_than_the_halo():
    with pytest.raises(ValueError, match="fewer than the halo"):
        P.halo_exchange(torch.zeros(5, 8), 21, 0, 1)


def test_row_ranges_tile_the_raster():
    for world in (1, 2, 3, 8):
        for H in (8, 17, 16384, 65536 + 5):
            rows = [P.row_range(r, world, H) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            for r in range(world):
                g0, g1, top, bot = P.local_range(r, world, H)
                assert (top, bot) == (r > 0, r < world - 1)
                assert g0 == rows[r][0] - top and g1 == rows[r][1] + bot
                g = P.ghost_rows(world, H)
                assert 1 <= g <= max(1, H // world - 1) and (world == 1 or g <= P.GHOST_ROWS)
                g0, g1, _, _ = P.local_range(r, world, H, g)
                assert g0 >= 0 and g1 <= H and g0 == rows[r][0] - g * top


# --------------------------------------------------------------------------
# communicators: a sub-group of the world, and virtual ranks in one process
# --------------------------------------------------------------------------
def _subgroup_worker(rank, world, port, H, W, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        members = [0, 2]                           # global rank 1 sits this one out
        group = dist.new_group(ranks=members)
        if rank in members:
            r, n = members.index(rank), len(members)
            g0, g1, _, _ = P.local_range(r, n, H)
            zt = torch.from_numpy(oracle.synth_dem(H, W, row0=g0, rows=g1 - g0))
            w, info = P.sinkfill_distributed(zt, r, n, NumpyLocalSolver(), group=group,
                                             coarse_block=4)
            bm = P.boxmean_distributed(w[P.owned_slice(r, n)].contiguous(), r, n,
                                       NumpyLocalSolver(), group=group)
            np.savez(os.path.join(outdir, f"g{r}.npz"), w=w.numpy()[P.owned_slice(r, n)],
                     bm=bm.numpy(), exchanges=info["exchanges"])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_partition_over_a_sub_group_addresses_the_right_peers(tmp_path):
    """The neighbours of a rank are its neighbours *in the group*: with the group
    {0, 2} of a 3-process world, group rank 1 is global rank 2."""
    H, W = 90, 64
    mp.spawn(_subgroup_worker, args=(3, _free_port(), H, W, str(tmp_path)), nprocs=3, join=True)
    want = c_oracle.sinkfill_pflood(oracle.synth_dem(H, W))
    parts = [np.load(tmp_path / f"g{r}.npz") for r in range(2)]
    assert np.array_equal(np.concatenate([p["w"] for p in parts]), want)
    assert np.array_equal(np.concatenate([p["bm"] for p in parts]), oracle.boxmean3_round(want))
    with pytest.raises(ValueError, match="do not match the communicator"):
        class Fake:                                 # pylint: disable=too-few-public-methods
            rank, world = 0, 3
        P.sinkfill_distributed(torch.zeros(8, 8), 1, 3, NumpyLocalSolver(), comm=Fake())


@pytest.mark.parametrize("world,H,W,variant,ghost", [(2, 96, 80, "rough", 1),
                                                    (4, 130, 70, "srtm", 9),
                                                    (3, 99, 64, "rough", 62)])
def test_virtual_ranks_in_one_process_run_the_same_schedule(world, H, W, variant, ghost):
    """ThreadWorld: the distributed schedule with threads for ranks and row copies for
    the transport (what the one-GPU tests of BASELINE configs 4 and 5 run on)."""
    z = oracle.synth_dem(H, W, variant=variant)
    ghost = P.ghost_rows(world, H, ghost)

    def rank_body(rank, comm):
        g0, g1, _, _ = P.local_range(rank, world, H, ghost)
        zt = torch.from_numpy(z[g0:g1].copy())
        d = torch.empty(zt.shape, dtype=torch.uint8)
        w, info = P.sinkfill_distributed(zt, rank, world, NumpyLocalSolver(), coarse_block=4,
                                         d8_out=d, ghost=ghost, comm=comm)
        own = P.owned_slice(rank, world, ghost)
        r0, r1 = P.row_range(rank, world, H)
        gr = P.groves_distributed(torch.from_numpy(z[r0:r1].copy()),
                                  torch.from_numpy(oracle.synth_groves(H, W)[r0:r1].copy()),
                                  rank, world, NumpyLocalSolver(), iterations=1, comm=comm)
        return w.numpy()[own], d.numpy()[own], info, gr.numpy()

    got = P.ThreadWorld(world).run(rank_body)
    want = c_oracle.sinkfill_pflood(z)
    assert np.array_equal(np.concatenate([g[0] for g in got]), want)
    assert np.array_equal(np.concatenate([g[1] for g in got]), c_oracle.d8(want))
    assert all(g[2]["exchanges"] >= 2 and g[2]["verifications"] >= 1 for g in got)
    assert all(g[2]["solves"][0][0] == "first" and g[2]["solves"][-1][0] == "verify" for g in got)
    want_g = oracle.groves_exact64(z, oracle.synth_groves(H, W), 1, 15, 1.5)[0].astype(np.float32)
    assert np.array_equal(np.concatenate([g[3] for g in got]), want_g)


def test_a_failing_virtual_rank_releases_the_others():
    def rank_body(rank, comm):
        if rank == 1:
            raise KeyError("rank 1 gives up")
        comm.all_reduce_max(torch.zeros(1, dtype=torch.int32))
    with pytest.raises(KeyError, match="rank 1 gives up"):
        P.ThreadWorld(3).run(rank_body)
