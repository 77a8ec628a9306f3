"""
Row-block decomposition of the raster path over the GPUs of one node
(one process per GPU, ``torch.distributed``; backend ``nccl`` is RCCL over
xGMI on ROCm, ``gloo`` in the CPU tests).  New work: the reference is single
process.

Layout.  Rank r owns rows [r*H/P, (r+1)*H/P) of the H x W raster and stores
them with one ghost row above (r > 0) and below (r < P-1); columns are not
split, so row loads stay coalesced and a rank has at most two neighbours.
The first and last row of every local array are therefore the Dirichlet ring
of the local problem -- the raster border on the outer ranks, a ghost row
elsewhere -- which is exactly what the single-GPU solver pins.

Sink fill.  repeat { relax the local block to its fixed point with the ghost
rows frozen ; swap boundary rows with rank+-1 (point-to-point, W*4 bytes each
way) ; all-reduce one "any ghost row changed" flag } until the flag is clear;
then every rank runs one verifying pass over its whole block (the intermediate
solves skip it) and the loop resumes only if that pass lowered something.
Legal for any interleaving because the relaxation is monotone from above
(stale ghost rows are upper bounds: they delay, never corrupt), and the state
at exit is a fixed point of the global operator, hence the same bits as the
single-GPU result.  D8 needs the ghost rows of the filled surface, which the
last exchange leaves in place.

The local solver is injected (``solver=``): the HIP backend on GPUs; the CPU
tests pass a NumPy solver so that this exchange logic runs under ``gloo``.
"""

import numpy as np

from . import backend


def row_range(rank, world, total_rows):
    """Rows owned by ``rank`` (balanced split, remainder to the low ranks)."""
    base, rem = divmod(total_rows, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def local_range(rank, world, total_rows):
    """(first, last+1) global rows of the local array incl. ghost rows, and
    the (has_top_ghost, has_bottom_ghost) pair."""
    r0, r1 = row_range(rank, world, total_rows)
    top, bottom = rank > 0, rank < world - 1
    return r0 - int(top), r1 + int(bottom), top, bottom


class HipLocalSolver:
    """Local block solver on the HIP backend; tensors are CUDA torch tensors
    whose memory the kernels use in place (no copies)."""

    def __init__(self, device_index=None):
        import torch
        self.torch = torch
        self.ctx = backend.context(torch.cuda.current_device()
                                   if device_index is None else device_index)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def _wrap(self, t, dtype):
        return backend.DeviceRaster.wrap(t.data_ptr(), tuple(t.shape), dtype,
                                         ctx=self.ctx, keepalive=t)

    def fill(self, z, w, eps, flags):
        """Returns (tile visits, whether any cell was lowered)."""
        _, st = backend.sinkfill_dev(self._wrap(z, np.float32), eps=eps,
                                     out=self._wrap(w, np.float32), flags=flags)
        return st["tile_visits"], st["tile_visits"] > st["visits_unchanged"]

    def d8(self, w, out):
        backend.d8_dev(self._wrap(w, np.float32), out=self._wrap(out, np.uint8))


def _exchange(dist, torch, w, top, bottom, rank):
    """Swap boundary rows with the neighbours; returns (top_changed,
    bottom_changed) as Python bools.  One batched isend/irecv group.  Under the
    gloo backend device tensors are staged through the host (gloo has no device
    point-to-point); that is the rehearsal path, RCCL moves device memory."""
    ops, recv_top, recv_bot = [], None, None
    h = w.shape[0]
    stage = w.is_cuda and dist.get_backend() == "gloo"
    buf_dev = torch.device("cpu") if stage else w.device
    if top:
        recv_top = torch.empty(w.shape[1], dtype=w.dtype, device=buf_dev)
        ops.append(dist.P2POp(dist.isend, w[1].to(buf_dev).contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, recv_top, rank - 1))
    if bottom:
        recv_bot = torch.empty(w.shape[1], dtype=w.dtype, device=buf_dev)
        ops.append(dist.P2POp(dist.isend, w[h - 2].to(buf_dev).contiguous(), rank + 1))
        ops.append(dist.P2POp(dist.irecv, recv_bot, rank + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if stage:
        recv_top = recv_top.to(w.device) if top else None
        recv_bot = recv_bot.to(w.device) if bottom else None
    # NaN (nodata) never compares equal: compare bit patterns
    flags = torch.zeros(2, dtype=torch.int32, device=w.device)
    if top:
        flags[0] = (recv_top.view(torch.int32) != w[0].view(torch.int32)).any()
        w[0].copy_(recv_top)
    if bottom:
        flags[1] = (recv_bot.view(torch.int32) != w[h - 1].view(torch.int32)).any()
        w[h - 1].copy_(recv_bot)
    f = flags.cpu()
    return bool(f[0]), bool(f[1])


def sinkfill_distributed(z_local, rank, world, solver, eps=0.0, w_out=None,
                         max_exchanges=100000, group=None):
    """Sink fill of a row-block partitioned raster.

    ``z_local``: torch tensor, local rows incl. ghost rows (see
    :func:`local_range`), float32.  Returns (w_local, info): ``w_local`` has
    the same shape, ghost rows holding the neighbours' final values."""
    import torch
    import torch.distributed as dist

    top, bottom = rank > 0, rank < world - 1
    w = torch.empty_like(z_local) if w_out is None else w_out
    flags = backend.FILL_INIT
    if top:
        flags |= backend.FILL_GHOST_TOP
    if bottom:
        flags |= backend.FILL_GHOST_BOTTOM
    flag_dev = None
    visits, _ = solver.fill(z_local, w, eps, flags | backend.FILL_NO_VERIFY)
    exchanges = verifications = 0
    while world > 1:
        ch_top, ch_bot = _exchange(dist, torch, w, top, bottom, rank)
        if flag_dev is None:
            flag_dev = "cpu" if dist.get_backend() == "gloo" else w.device
        any_changed = torch.tensor([int(ch_top or ch_bot)], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(any_changed, op=dist.ReduceOp.MAX, group=group)
        exchanges += 1
        if exchanges >= max_exchanges:
            raise RuntimeError("distributed sink fill did not converge")
        if int(any_changed.item()) == 0:
            # every rank is at rest: certify the whole block (round driver, all tiles
            # due); resume only if some rank still found something to lower
            v, lowered = solver.fill(z_local, w, eps, backend.FILL_WARM | backend.FILL_SYNC_ONLY)
            visits += v
            verifications += 1
            again = torch.tensor([int(lowered)], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(again, op=dist.ReduceOp.MAX, group=group)
            if int(again.item()) == 0:
                break
            continue
        if ch_top or ch_bot:
            act = backend.FILL_WARM | backend.FILL_NO_VERIFY
            act |= backend.FILL_ACT_TOP if ch_top else 0
            act |= backend.FILL_ACT_BOTTOM if ch_bot else 0
            v, _ = solver.fill(z_local, w, eps, act)
            visits += v
    if world == 1:
        v, _ = solver.fill(z_local, w, eps, backend.FILL_WARM | backend.FILL_SYNC_ONLY)
        visits += v
    return w, {"tile_visits": int(visits), "exchanges": exchanges}


def d8_distributed(w_local, solver, out=None):
    """D8 on the local block; ghost rows of ``w_local`` must hold the
    neighbours' filled values (they do after :func:`sinkfill_distributed`).
    Codes of ghost rows are meaningless and should be dropped by the caller
    (``owned_slice``)."""
    import torch
    out = torch.empty(w_local.shape, dtype=torch.uint8, device=w_local.device) \
        if out is None else out
    solver.d8(w_local, out)
    return out


def owned_slice(rank, world):
    """Slice of the local array that holds the owned rows."""
    return slice(1 if rank > 0 else 0, -1 if rank < world - 1 else None)
