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

Sink fill.  repeat { relax the local block with the ghost rows frozen, for one
time slice or to its fixed point, whichever comes first ; swap boundary rows
with rank+-1 (point-to-point, W*4 bytes each way) ; all-reduce one "a ghost row
changed or tiles are still queued somewhere" flag } until the flag is clear;
then every rank runs one verifying pass over its whole block (the intermediate
solves skip it) and the loop resumes only if that pass lowered something.
(The time slice is optional, ``DEFAULT_SLICE_US``: with good start values for the
ghost rows -- next paragraph -- solving to the local fixed point between exchanges
is the faster schedule.)
Legal for any interleaving because the relaxation is monotone from above
(stale ghost rows are upper bounds: they delay, never corrupt), and the state
at exit is a fixed point of the global operator, hence the same bits as the
single-GPU result.  D8 needs the ghost rows of the filled surface, which the
last exchange leaves in place.

Start values.  A rank that starts with its ghost rows at +inf first fills
against two walls and redoes most of that when the neighbours' real rows arrive
(measured: 2.3x the tile visits of an undivided raster).  So the ranks first
solve the *whole* raster on a coarse grid: each takes block maxima of its owned
rows (``COARSE_BLOCK`` x ``COARSE_BLOCK`` cells, NaN -> wall), all-gathers them
(W*H/b^2 floats in total) and fills the stacked coarse raster -- redundantly, it
is tiny.  A fine path that stays inside a chain of adjacent blocks never exceeds
the chain's maxima, so the coarse fill bounds the fine fill from above, which is
all a start value has to satisfy; every free cell of the block -- ghost rows
included -- starts at the coarse level of its block instead of +inf
(``hdem_set_fill_coarse_start``; the single-GPU solver does the same with its own
coarse raster).  Only for epsilon = 0 (with a gradient the bound would need the
path length).

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


# Rows of overlap on each side of a block.  Only the outermost one is a ghost row in the
# solver's sense (pinned, replaced by the exchange); the others are relaxed by both
# neighbours.  Cells next to a seam settle their back-and-forth inside one rank instead of
# one exchange at a time: with one tile row of overlap 4 x 16384^2 needs 3 correcting
# rounds instead of 8 (tools/emulate_overlap.py), for 0.4 % more cells per rank.
GHOST_ROWS = 62


def ghost_rows(world, total_rows, want=GHOST_ROWS):
    """Overlap every rank uses: ``want``, cut down so that the row a neighbour pins is
    always one the rank owns (blocks of at least that many rows)."""
    if world <= 1:
        return 1
    return max(1, min(int(want), total_rows // world - 1))


def local_range(rank, world, total_rows, ghost=1):
    """(first, last+1) global rows of the local array incl. ``ghost`` rows of overlap
    on each inner side, and the (has_top_ghost, has_bottom_ghost) pair."""
    r0, r1 = row_range(rank, world, total_rows)
    top, bottom = rank > 0, rank < world - 1
    return r0 - ghost * int(top), r1 + ghost * int(bottom), top, bottom


# Time slice of the intermediate solves in microseconds; 0 = every solve runs to its local
# fixed point.  Slicing lets blocks trade rows while both still relax, which pays when
# the ghost rows start far from the truth (+inf: 26 -> 23 ms predicted at 4 x 16384^2);
# behind the coarse pre-solve the start values are good and the extra exchanges cost
# more than they save (17.3 ms unsliced, 18.9 ms with 2.5 ms slices), so it is off.
DEFAULT_SLICE_US = 0
# Edge of the blocks of the coarse pre-solve (power of two, 4..256).
COARSE_BLOCK = 16


class HipLocalSolver:
    """Local block solver on the HIP backend; tensors are CUDA torch tensors
    whose memory the kernels use in place (no copies)."""

    def __init__(self, device_index=None, slice_us=DEFAULT_SLICE_US, own_context=True):
        import torch
        self.torch = torch
        device_index = torch.cuda.current_device() if device_index is None else device_index
        # its own context by default: the stream below is bound to it, and the worklist of a
        # time-sliced solve lives in it (one context per block solved in a process)
        self.ctx = backend.Context(device_index) if own_context else backend.context(device_index)
        # The kernels run on a torch stream of their own, ordered against torch's current
        # stream with events on the way in and out of every call.  (Handing over torch's
        # current stream directly does not work for the default stream: its handle is 0,
        # the legacy NULL stream, which does not order against non-blocking streams.)
        with torch.cuda.device(device_index):
            self.stream = torch.cuda.Stream()
        self.ctx.set_stream(self.stream.cuda_stream)
        self.slice_us = slice_us

    def _enter(self):
        self.stream.wait_stream(self.torch.cuda.current_stream())

    def _exit(self):
        self.torch.cuda.current_stream().wait_stream(self.stream)

    def _wrap(self, t, dtype):
        return backend.DeviceRaster.wrap(t.data_ptr(), tuple(t.shape), dtype,
                                         ctx=self.ctx, keepalive=t)

    def fill(self, z, w, eps, flags, sliced=False, d8=None):
        """Returns (tile visits, whether any cell was lowered, tiles still queued).
        ``sliced``: stop after ``slice_us`` even if tiles are still queued (only
        honoured together with FILL_NO_VERIFY).  ``d8``: a uint8 tensor that receives
        the flow directions of the filled block (the certifying pass writes them)."""
        sliced = bool(sliced and self.slice_us > 0)
        self.ctx.set_fill_slice_us(self.slice_us if sliced else 0)
        self._enter()
        try:
            if d8 is not None:
                _, _, st = backend.sinkfill_d8_dev(self._wrap(z, np.float32), eps=eps,
                                                   out=self._wrap(w, np.float32),
                                                   codes=self._wrap(d8, np.uint8), flags=flags)
            else:
                _, st = backend.sinkfill_dev(self._wrap(z, np.float32), eps=eps,
                                             out=self._wrap(w, np.float32), flags=flags)
        finally:
            self.ctx.set_fill_slice_us(0)
            self._exit()
        return st["tile_visits"], st["tile_visits"] > st["visits_unchanged"], st["pending"]

    def set_coarse_start(self, filled, block, row_map):
        """Start values of the next INIT ``fill``: device tensors, see
        ``hdem_set_fill_coarse_start``."""
        self.ctx.set_fill_coarse_start(filled.data_ptr(), filled.shape[0], filled.shape[1],
                                       block, row_map.data_ptr())

    def d8(self, w, out):
        self._enter()
        backend.d8_dev(self._wrap(w, np.float32), out=self._wrap(out, np.uint8))
        self._exit()

    def groves(self, img, mask, window_size, threshold, iterations):
        out = self.torch.empty_like(img)
        scratch = self.torch.empty_like(img)
        self._enter()
        backend.groves_dev(self._wrap(img, np.float32), self._wrap(mask, np.uint8),
                           window_size, threshold, iterations,
                           out=self._wrap(out, np.float32),
                           scratch=self._wrap(scratch, np.float32))
        self._exit()
        return out

    def boxmean(self, x, do_round):
        dt = np.float64 if x.dtype == self.torch.float64 else np.float32
        out = self.torch.empty_like(x)
        self._enter()
        backend.boxmean3_dev(self._wrap(x, dt), do_round, out=self._wrap(out, dt))
        self._exit()
        return out

    def blockmax(self, z, block):
        """Block maxima of ``z`` (contiguous rows), NaN -> FLT_MAX wall."""
        out = self.torch.empty((-(-z.shape[0] // block), -(-z.shape[1] // block)),
                               dtype=z.dtype, device=z.device)
        self._enter()
        backend.blockmax_dev(self._wrap(z, np.float32), block,
                             out=self._wrap(out, np.float32))
        self._exit()
        return out


def _exchange(dist, torch, w, top, bottom, rank, ghost=1):
    """Swap boundary rows with the neighbours; returns a 2-element int32 tensor on
    ``w``'s device, (top_changed, bottom_changed) -- not read back here, so that the
    caller pays one host synchronisation per exchange for flags and all-reduce
    together.  One batched isend/irecv group.  Under the gloo backend device tensors
    are staged through the host (gloo has no device point-to-point); that is the
    rehearsal path, RCCL moves device memory."""
    ops, recv_top, recv_bot = [], None, None
    h = w.shape[0]
    stage = w.is_cuda and dist.get_backend() == "gloo"
    buf_dev = torch.device("cpu") if stage else w.device
    if top:
        recv_top = torch.empty(w.shape[1], dtype=w.dtype, device=buf_dev)
        # the row rank-1 pins is ghost rows into my block: index 2 * ghost - 1 here
        ops.append(dist.P2POp(dist.isend, w[2 * ghost - 1].to(buf_dev).contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, recv_top, rank - 1))
    if bottom:
        recv_bot = torch.empty(w.shape[1], dtype=w.dtype, device=buf_dev)
        ops.append(dist.P2POp(dist.isend, w[h - 2 * ghost].to(buf_dev).contiguous(), rank + 1))
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
    return flags


def _exchange_and_vote(dist, torch, w, top, bottom, rank, pending, group, ghost=1):
    """One halo exchange plus the global "is anybody still busy" vote.  Returns
    (any rank busy, top ghost changed, bottom ghost changed)."""
    flags = _exchange(dist, torch, w, top, bottom, rank, ghost)
    busy = (flags.max() + int(pending > 0)).clamp(max=1).reshape(1)
    if dist.get_backend() == "gloo":
        busy = busy.cpu()                       # gloo reduces host tensors
    dist.all_reduce(busy, op=dist.ReduceOp.MAX, group=group)
    out = torch.cat([busy.to(flags.device), flags]).cpu()          # the one read-back
    return bool(out[0]), bool(out[1]), bool(out[2])


def _all_gather(dist, torch, t, world, group):
    """all_gather of equally shaped tensors; device tensors are staged through the
    host under gloo (the CPU rehearsal path)."""
    stage = t.is_cuda and dist.get_backend() == "gloo"
    src = t.cpu() if stage else t
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return [p.to(t.device) for p in parts] if stage else parts


def coarse_start(z_local, rank, world, solver, block=COARSE_BLOCK, group=None, ghost=1):
    """Start values from a fill of the whole raster coarsened to block maxima (see the
    module docstring).  Returns (filled, row_map): the filled stacked coarse raster
    (every rank holds the same one) and, per row of ``z_local`` (ghost rows included),
    the coarse row that bounds it -- what ``hdem_set_fill_coarse_start`` takes."""
    import torch
    import torch.distributed as dist

    top, bottom = rank > 0, rank < world - 1
    owned = z_local[owned_slice(rank, world, ghost)]
    mine = solver.blockmax(owned.contiguous(), block)
    # ranks own floor or ceil(H/world) rows: pad to a common shape for the all_gather
    # (each rank tells its coarse and its fine row count)
    counts = torch.tensor([mine.shape[0] + (owned.shape[0] << 32)], dtype=torch.int64)
    all_counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    if dist.get_backend() == "gloo":
        dist.all_gather(all_counts, counts, group=group)
    else:
        dev_counts = _all_gather(dist, torch, counts.to(z_local.device), world, group)
        all_counts = [c.cpu() for c in dev_counts]
    fine = [int(c.item()) >> 32 for c in all_counts]
    rows = [int(c.item()) & 0xffffffff for c in all_counts]
    padded = torch.full((max(rows), mine.shape[1]), float("inf"), dtype=mine.dtype,
                        device=mine.device)
    padded[:mine.shape[0]] = mine
    parts = _all_gather(dist, torch, padded, world, group)
    coarse = torch.cat([p[:n] for p, n in zip(parts, rows)]).contiguous()
    filled = torch.empty_like(coarse)
    solver.fill(coarse, filled, 0.0, backend.FILL_INIT | backend.FILL_NO_VERIFY)
    # coarse row of every local row: owned rows in my part of the stack, overlap rows in
    # the neighbours' (a rank's coarse rows start at its first owned row)
    def part(r, lo, hi):                           # rows [lo, hi) of rank r, counted from its first
        return sum(rows[:r]) + torch.arange(lo, hi, dtype=torch.int32) // block
    pieces = []
    if top:
        pieces.append(part(rank - 1, fine[rank - 1] - ghost, fine[rank - 1]))
    pieces.append(part(rank, 0, owned.shape[0]))
    if bottom:
        pieces.append(part(rank + 1, 0, ghost))
    row_map = torch.cat(pieces)
    return filled, row_map.to(torch.int32).to(z_local.device)


def sinkfill_distributed(z_local, rank, world, solver, eps=0.0, w_out=None,
                         max_exchanges=100000, group=None, coarse_block=None, d8_out=None,
                         ghost=1):
    """Sink fill of a row-block partitioned raster.

    ``z_local``: torch tensor, local rows incl. ghost rows (see
    :func:`local_range`), float32.  Returns (w_local, info): ``w_local`` has
    the same shape, ghost rows holding the neighbours' final values.  ``d8_out``
    (uint8, same shape): also receives the D8 codes of the filled block, written by
    the last verifying pass (rows of ghost rows are meaningless, as in
    :func:`d8_distributed`).  ``ghost``: rows of overlap the local array was cut with
    (:func:`local_range`, :func:`ghost_rows`); only the outermost is pinned."""
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
    sliced = world > 1
    if coarse_block is None:
        # (every rank fills the whole stacked coarse raster; at 8 x 16384^2 that is
        # 8192 x 1024 cells and 1.6 ms, and 32 x 32 blocks would cost more in the fine
        # solve than they save here: 16.1 against 15.4 ms predicted)
        coarse_block = COARSE_BLOCK
    keep = None
    if world > 1 and eps == 0.0 and coarse_block:
        keep = coarse_start(z_local, rank, world, solver, coarse_block, group, ghost)
        solver.set_coarse_start(keep[0], coarse_block, keep[1])
    visits, _, pending = solver.fill(z_local, w, eps, flags | backend.FILL_NO_VERIFY, sliced)
    del keep                                       # (alive until the solve has consumed them)
    exchanges = verifications = 0
    while world > 1:
        any_busy, ch_top, ch_bot = _exchange_and_vote(dist, torch, w, top, bottom, rank,
                                                      pending, group, ghost)
        if flag_dev is None:
            flag_dev = "cpu" if dist.get_backend() == "gloo" else w.device
        exchanges += 1
        if exchanges >= max_exchanges:
            raise RuntimeError("distributed sink fill did not converge")
        if not any_busy:
            # every rank is at rest: certify the whole block (round driver, all tiles
            # due); resume only if some rank still found something to lower
            v, lowered, pending = solver.fill(z_local, w, eps,
                                              backend.FILL_WARM | backend.FILL_SYNC_ONLY,
                                              d8=d8_out)
            visits += v
            verifications += 1
            again = torch.tensor([int(lowered)], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(again, op=dist.ReduceOp.MAX, group=group)
            if int(again.item()) == 0:
                break
            continue
        if ch_top or ch_bot or pending > 0:
            # next slice: the tiles left queued plus those next to a replaced ghost row
            act = backend.FILL_WARM | backend.FILL_RESUME | backend.FILL_NO_VERIFY
            act |= backend.FILL_ACT_TOP if ch_top else 0
            act |= backend.FILL_ACT_BOTTOM if ch_bot else 0
            v, _, pending = solver.fill(z_local, w, eps, act, sliced)
            visits += v
    if world == 1:
        v, _, _ = solver.fill(z_local, w, eps, backend.FILL_WARM | backend.FILL_SYNC_ONLY,
                              d8=d8_out)
        visits += v
    return w, {"tile_visits": int(visits), "exchanges": exchanges,
               "verifications": verifications}


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


def halo_exchange(owned, halo, rank, world, group=None):
    """Owned rows plus ``halo`` rows of each neighbour (one batched isend/irecv
    group; SURVEY 8e: D8 / box mean 1 row, quadratic 7 rows per pass).  Returns
    (extended tensor, rows on top that belong to rank-1, rows at the bottom that
    belong to rank+1).  Ranks must own at least ``halo`` rows."""
    import torch
    import torch.distributed as dist

    top, bottom = rank > 0, rank < world - 1
    if owned.shape[0] < halo:
        raise ValueError(f"rank {rank} owns {owned.shape[0]} rows, fewer than the halo of {halo}")
    stage = owned.is_cuda and dist.get_backend() == "gloo"
    dev = torch.device("cpu") if stage else owned.device
    ops, r_top, r_bot = [], None, None
    if top:
        r_top = torch.empty((halo,) + tuple(owned.shape[1:]), dtype=owned.dtype, device=dev)
        ops.append(dist.P2POp(dist.isend, owned[:halo].to(dev).contiguous(), rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, r_top, rank - 1, group))
    if bottom:
        r_bot = torch.empty((halo,) + tuple(owned.shape[1:]), dtype=owned.dtype, device=dev)
        ops.append(dist.P2POp(dist.isend, owned[-halo:].to(dev).contiguous(), rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, r_bot, rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    parts = ([r_top.to(owned.device)] if top else []) + [owned] + \
            ([r_bot.to(owned.device)] if bottom else [])
    return torch.cat(parts).contiguous(), halo if top else 0, halo if bottom else 0


def groves_distributed(img_owned, groves_owned, rank, world, solver, iterations=3,
                       window_size=15, threshold=1.5, group=None):
    """``GrovesCorrectionsIter`` on a row-block partitioned raster: one exchange of
    ``iterations * (window_size // 2)`` rows each way, then the fused passes run on the
    extended block and the overlap is recomputed instead of exchanged again.  Pass k
    is right from row k * (window_size // 2) of the extended block on, so after the
    last pass the owned rows are; the raster's own first and last rows keep the
    reference's untouched border ring because they are the block's."""
    halo = iterations * (window_size // 2)
    img, t, b = halo_exchange(img_owned, halo, rank, world, group)
    mask, _, _ = halo_exchange(groves_owned, halo, rank, world, group)
    out = solver.groves(img, mask, window_size, threshold, iterations)
    return out[t:out.shape[0] - b]


def boxmean_distributed(x_owned, rank, world, solver, do_round=True, group=None):
    """``PostProcessingFinal`` (3 x 3 mean, reflect at the raster border, optional
    rounding) on a row-block partitioned raster: one ghost row each way."""
    x, t, b = halo_exchange(x_owned, 1, rank, world, group)
    out = solver.boxmean(x, do_round)
    return out[t:out.shape[0] - b]


def owned_slice(rank, world, ghost=1):
    """Slice of the local array that holds the owned rows."""
    return slice(ghost if rank > 0 else 0, -ghost if rank < world - 1 else None)
