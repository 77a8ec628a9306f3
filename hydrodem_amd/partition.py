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

import contextlib
import os
import time

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

    def __init__(self, device_index=None, slice_us=DEFAULT_SLICE_US, own_context=True,
                 turn=None):
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
        # virtual ranks sharing one GPU (ThreadWorld) solve one at a time
        self.turn = turn
        self.timings = []

    @contextlib.contextmanager
    def _call(self, label="call"):
        """Order the library's stream behind torch's current stream on the way in and
        torch's behind the library's on the way out -- also when the call raises.
        Virtual ranks (``turn``): the call has the GPU to itself and is waited for, so its
        wall time is what it would take on a GPU of its own -- kept in ``timings``."""
        if self.turn is not None:
            self.turn.acquire()
            self.torch.cuda.current_stream().synchronize()
            t0 = time.perf_counter()
        try:
            self.stream.wait_stream(self.torch.cuda.current_stream())
            try:
                yield
            finally:
                self.torch.cuda.current_stream().wait_stream(self.stream)
        finally:
            if self.turn is not None:
                self.stream.synchronize()
                self.timings.append((label, (time.perf_counter() - t0) * 1e3))
                self.turn.release()

    def _wrap(self, t, dtype):
        return backend.DeviceRaster.wrap(t.data_ptr(), tuple(t.shape), dtype,
                                         ctx=self.ctx, keepalive=t)

    def fill(self, z, w, eps, flags, sliced=False, d8=None):
        """Returns (tile visits, whether any cell was lowered, tiles still queued).
        ``sliced``: stop after ``slice_us`` even if tiles are still queued (only
        honoured together with FILL_NO_VERIFY).  ``d8``: a uint8 tensor that receives
        the flow directions of the filled block (the certifying pass writes them).
        ``last_stats`` keeps the library's counters of the call."""
        sliced = bool(sliced and self.slice_us > 0)
        with self._call("fill"):
            self.ctx.set_fill_slice_us(self.slice_us if sliced else 0)
            try:
                if d8 is not None:
                    _, _, st = backend.sinkfill_d8_dev(self._wrap(z, np.float32), eps=eps,
                                                       out=self._wrap(w, np.float32),
                                                       codes=self._wrap(d8, np.uint8),
                                                       flags=flags)
                else:
                    _, st = backend.sinkfill_dev(self._wrap(z, np.float32), eps=eps,
                                                 out=self._wrap(w, np.float32), flags=flags)
            finally:
                self.ctx.set_fill_slice_us(0)
        self.last_stats = st
        # (counters of deferred solves before this call ride along: not this call's lowering)
        own = st["tile_visits"] - st.get("deferred_visits", 0)
        own_same = st["visits_unchanged"] - st.get("deferred_unchanged", 0)
        return st["tile_visits"], own > own_same, st["pending"]

    # the correcting solves of the exchange loop can be enqueued without a host wait
    can_defer = True

    def fill_deferred(self, z, w, eps, flags, words):
        """Enqueue a correcting solve (``FILL_WARM | FILL_RESUME`` + ACT bits) and return
        without waiting: ``words`` (int32[3], device) carries what the solve needs to know and
        what it has to tell -- ``hdem_set_fill_seam_words``.  Its counters arrive with the next
        :meth:`fill` (``deferred_visits``)."""
        with self._call("fill_deferred"):
            self.ctx.set_fill_slice_us(self.slice_us)
            self.ctx.set_fill_seam_words(words.data_ptr())
            try:
                backend.sinkfill_dev(self._wrap(z, np.float32), eps=eps,
                                     out=self._wrap(w, np.float32),
                                     flags=flags | backend.FILL_DEFER)
            finally:
                self.ctx.set_fill_slice_us(0)
                self.ctx.set_fill_seam_words(0)

    def seam_apply(self, w, recv_top, recv_bot, pending, words):
        """Received rows into the ghost rows of ``w``, seam words set -- one launch."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        with self._call("seam"):
            self.ctx.fill_seam_apply(w.data_ptr(), w.shape[0], w.shape[1], ptr(recv_top),
                                     ptr(recv_bot), pending, words.data_ptr())

    def set_coarse_start(self, filled, block, row_map):
        """Start values of the next INIT ``fill``: device tensors, see
        ``hdem_set_fill_coarse_start``."""
        self.ctx.set_fill_coarse_start(filled.data_ptr(), filled.shape[0], filled.shape[1],
                                       block, row_map.data_ptr())

    # -- hub start of the partition (hub_start below) ---------------------------------------
    TILE = 62                                          # tile edge of the fill (hdem_fill_stats.tile_h)

    def hub_prepare(self, z, w, flags):
        """Path costs d of every cell to its tile's hub into the interior of ``w``."""
        with self._call("hub_prepare"):
            self.ctx.fill_hub_prepare(z.data_ptr(), z.shape[0], z.shape[1], flags, w.data_ptr())

    def hub_raster(self, z):
        """The prepared block's hub raster (ghost rows of ``w`` hold the neighbours' d)."""
        t = self.TILE
        out = self.torch.empty((2 * (-(-(z.shape[0] - 2) // t)) + 1, 2 * (-(-(z.shape[1] - 2) // t)) + 1),
                               dtype=self.torch.float32, device=z.device)
        with self._call("hub_raster"):
            self.ctx.fill_hub_raster(out.data_ptr())
        return out

    def set_hub_levels(self, levels):
        self.ctx.set_fill_hub_levels(levels.data_ptr())

    def d8(self, w, out):
        with self._call():
            backend.d8_dev(self._wrap(w, np.float32), out=self._wrap(out, np.uint8))

    def groves(self, img, mask, window_size, threshold, iterations):
        out = self.torch.empty_like(img)
        scratch = self.torch.empty_like(img)
        with self._call():
            backend.groves_dev(self._wrap(img, np.float32), self._wrap(mask, np.uint8),
                               window_size, threshold, iterations,
                               out=self._wrap(out, np.float32),
                               scratch=self._wrap(scratch, np.float32))
        return out

    def boxmean(self, x, do_round):
        dt = np.float64 if x.dtype == self.torch.float64 else np.float32
        out = self.torch.empty_like(x)
        with self._call():
            backend.boxmean3_dev(self._wrap(x, dt), do_round, out=self._wrap(out, dt))
        return out

    def blockmax(self, z, block):
        """Block maxima of ``z`` (contiguous rows), NaN -> FLT_MAX wall."""
        out = self.torch.empty((-(-z.shape[0] // block), -(-z.shape[1] // block)),
                               dtype=z.dtype, device=z.device)
        with self._call():
            backend.blockmax_dev(self._wrap(z, np.float32), block,
                                 out=self._wrap(out, np.float32))
        return out


class DistComm:
    """The ranks of one ``torch.distributed`` process group (``nccl`` = RCCL over xGMI
    on the GPUs, ``gloo`` in the CPU tests and in the one-GPU rehearsal, where device
    rows are staged through the host: gloo has no device point-to-point).  Bulk data
    only moves between rank r and r +- 1; the collectives are a 4-byte MAX and one small
    all-gather."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.gloo = dist.get_backend(group) == "gloo"
        self._host = {}                                    # staging rows under gloo, by role

    def _peer(self, group_rank):
        # P2POp addresses global ranks; under a sub-group the neighbour is not rank +- 1
        return group_rank if self.group is None else \
            self.dist.get_global_rank(self.group, group_rank)

    def _staged(self, role, like):
        """Host twin of a device buffer (gloo only), allocated once per role and shape."""
        buf = self._host.get(role)
        if buf is None or buf.shape != like.shape or buf.dtype != like.dtype:
            buf = self.torch.empty(like.shape, dtype=like.dtype, device="cpu",
                                   pin_memory=like.is_cuda)
            self._host[role] = buf
        return buf

    def swap(self, send_up, send_down, recv_up, recv_down):
        """Send ``send_up`` to rank-1 and ``send_down`` to rank+1 (``None`` = no such
        neighbour) while receiving their counterparts into ``recv_up`` / ``recv_down``:
        one batched isend/irecv group."""
        dist, ops, staged = self.dist, [], []
        for send, recv, peer, role in ((send_up, recv_up, self.rank - 1, "up"),
                                       (send_down, recv_down, self.rank + 1, "down")):
            if send is None:
                continue
            if self.gloo and send.is_cuda:
                out = self._staged("s" + role, send)
                out.copy_(send)
                inp = self._staged("r" + role, recv)
                staged.append((recv, inp))
            else:
                out, inp = send.contiguous(), recv
            ops.append(dist.P2POp(dist.isend, out, self._peer(peer), self.group))
            ops.append(dist.P2POp(dist.irecv, inp, self._peer(peer), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for recv, inp in staged:
            recv.copy_(inp)

    def all_reduce_max(self, value):
        """MAX over the ranks of a small integer tensor (returned where it came from)."""
        if self.gloo and value.is_cuda:
            host = value.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.MAX, group=self.group)
            return host.to(value.device)
        self.dist.all_reduce(value, op=self.dist.ReduceOp.MAX, group=self.group)
        return value

    def all_gather(self, t):
        """Equally shaped tensors of all ranks, in rank order, on ``t``'s device."""
        src = t.cpu() if self.gloo and t.is_cuda else t.contiguous()
        parts = [self.torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(parts, src, group=self.group)
        return [p.to(t.device) for p in parts]


class ThreadWorld:
    """``world`` virtual ranks inside one process, one thread each, all on the same GPU:
    the schedule of :func:`sinkfill_distributed` -- coarse start, local solves, seam
    exchanges, votes, certification -- runs unchanged, only the transport differs
    (device-to-device row copies instead of RCCL).  For the default-on tests of the
    multi-GPU configurations on a one-GPU box (BASELINE configs 4 and 5) and for
    ``tools/emulate_ranks.py``.  Local solves take turns (``gpu_turn``): the persistent
    fill launch wants the whole device, as it has on a node with one GPU per rank."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.gpu_turn = threading.Lock()
        self.slots = [None] * world

    def comm(self, rank):
        return _ThreadComm(self, rank)

    def run(self, fn):
        """``fn(rank, comm)`` on every virtual rank; returns the results in rank order
        and re-raises the first failure (the others are released from their barriers)."""
        import threading
        results, errors = [None] * self.world, []

        def body(rank):
            try:
                results[rank] = fn(rank, self.comm(rank))
            except BaseException as exc:  # pylint: disable=broad-except
                if not isinstance(exc, threading.BrokenBarrierError):
                    errors.append(exc)
                self.barrier.abort()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        if self.barrier.broken:
            raise RuntimeError("a virtual rank left its barrier")
        return results


class _ThreadComm:
    def __init__(self, world, rank):
        self.shared, self.rank, self.world = world, rank, world.world
        self.gpu_turn = world.gpu_turn

    def _publish(self, item):
        self.shared.slots[self.rank] = item
        self.shared.barrier.wait()                         # everybody has published

    def _done(self):
        self.shared.barrier.wait()                         # everybody has read

    def swap(self, send_up, send_down, recv_up, recv_down):
        self._publish((send_up, send_down))
        if send_up is not None:
            recv_up.copy_(self.shared.slots[self.rank - 1][1])
        if send_down is not None:
            recv_down.copy_(self.shared.slots[self.rank + 1][0])
        if recv_up is not None and recv_up.is_cuda:
            import torch
            torch.cuda.current_stream().synchronize()      # the senders may overwrite now
        self._done()

    def all_reduce_max(self, value):
        self._publish(value)
        out = self.shared.slots[0].clone()
        for other in self.shared.slots[1:]:
            out = out.maximum(other.to(out.device))
        self._done()
        value.copy_(out)
        return value

    def all_gather(self, t):
        self._publish(t)
        parts = [p.clone() for p in self.shared.slots]
        self._done()
        return parts


class _Seam:
    """Receive rows and flag words of one block's seam exchanges, allocated once per
    distributed fill instead of once per exchange."""

    def __init__(self, torch, w, top, bottom):
        row = lambda: torch.empty(w.shape[1], dtype=w.dtype, device=w.device)
        self.recv_top = row() if top else None
        self.recv_bot = row() if bottom else None
        # busy, top, bottom (+ deferred loop: any of the three, the word that is voted on)
        self.word = torch.zeros(4, dtype=torch.int32, device=w.device)
        # deferred loop: the vote's copy on the host and the event that says it has arrived
        # (host tensors -- the CPU tests' solver -- need neither pinning nor an event)
        self.vote_host = torch.zeros(1, dtype=torch.int32, device="cpu", pin_memory=w.is_cuda)
        self.event = torch.cuda.Event() if w.is_cuda else None


def _exchange_and_vote(comm, torch, w, top, bottom, pending, seam, ghost=1):
    """One halo exchange plus the global "is anybody still busy" vote: swap the seam
    rows with rank +- 1, note which ghost row changed (bit patterns: NaN never compares
    equal), MAX-reduce one word.  One host read-back for all three answers.  Returns
    (any rank busy, top ghost changed, bottom ghost changed)."""
    h = w.shape[0]
    # the row rank-1 pins is `ghost` rows into my block: index 2 * ghost - 1 here
    comm.swap(w[2 * ghost - 1] if top else None, w[h - 2 * ghost] if bottom else None,
              seam.recv_top, seam.recv_bot)
    word = seam.word[:3]
    word.zero_()
    if top:
        word[1] = (seam.recv_top.view(torch.int32) != w[0].view(torch.int32)).any()
        w[0].copy_(seam.recv_top)
    if bottom:
        word[2] = (seam.recv_bot.view(torch.int32) != w[h - 1].view(torch.int32)).any()
        w[h - 1].copy_(seam.recv_bot)
    word[0] = (word[1:].max() + int(pending > 0)).clamp(max=1)
    busy = comm.all_reduce_max(word[:1].clone())
    out = torch.cat([busy, word[1:]]).cpu()                         # the one read-back
    return bool(out[0]), bool(out[1]), bool(out[2])


def _exchange_and_vote_deferred(comm, solver, w, top, bottom, pending, seam, ghost=1):
    """The same exchange and vote with every answer left on the device: the ghost-changed
    words stay in ``seam.word[1:]`` for the deferred solve that follows (its seeding reads them),
    the vote is copied to pinned host memory behind the all-reduce and ``seam.event`` marks its
    arrival -- the caller enqueues the next solve first and looks at the vote afterwards.
    ``pending`` < 0: ``seam.word[0]`` already holds what the last deferred solve left queued."""
    h = w.shape[0]
    comm.swap(w[2 * ghost - 1] if top else None, w[h - 2 * ghost] if bottom else None,
              seam.recv_top, seam.recv_bot)
    solver.seam_apply(w, seam.recv_top, seam.recv_bot, pending, seam.word)
    busy = comm.all_reduce_max(seam.word[3:])
    seam.vote_host.copy_(busy, non_blocking=True)
    if seam.event is not None:
        seam.event.record()


def coarse_start(z_local, comm, solver, block=COARSE_BLOCK, ghost=1):
    """Start values from a fill of the whole raster coarsened to block maxima (see the
    module docstring).  Returns (filled, row_map): the filled stacked coarse raster
    (every rank holds the same one) and, per row of ``z_local`` (ghost rows included),
    the coarse row that bounds it -- what ``hdem_set_fill_coarse_start`` takes."""
    import torch

    rank, world = comm.rank, comm.world
    top, bottom = rank > 0, rank < world - 1
    owned = z_local[owned_slice(rank, world, ghost)]
    mine = solver.blockmax(owned.contiguous(), block)
    # ranks own floor or ceil(H/world) rows: pad to a common shape for the all_gather
    # (each rank tells its coarse and its fine row count)
    counts = torch.tensor([mine.shape[0], owned.shape[0]], dtype=torch.int64,
                          device=z_local.device)
    all_counts = torch.stack(comm.all_gather(counts)).cpu()
    rows, fine = all_counts[:, 0].tolist(), all_counts[:, 1].tolist()
    padded = torch.full((max(rows), mine.shape[1]), float("inf"), dtype=mine.dtype,
                        device=mine.device)
    padded[:mine.shape[0]] = mine
    parts = comm.all_gather(padded)
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


HUB_BIG = 3.0e38                  # a wall of the hub raster (hdem_sinkfill.hip)


def hub_start(z_local, w, comm, solver, flags, ghost):
    """Start values of a partitioned fill from ONE hub graph over all blocks (epsilon = 0).

    The single-GPU fill starts from a graph of tile hubs (hdem_sinkfill.hip: a hub per 62 x 62
    tile, the minimax path cost d of every cell to its hub inside the tile, the cheapest
    crossings between hubs of neighbouring tiles; the graph is filled exactly as a small
    raster and a cell starts at max(d, level of its hub)).  Here every rank makes d and the
    hub raster of its own block; the seam rows of d are swapped into the ghost rows first, so
    that the crossings over a ghost row join a rank's top hubs to the hubs of the neighbour's
    tiles underneath that row.  The ranks' rasters are stacked -- rank r gives its rows down to
    the tile row that holds the row rank r+1 pins, rank r+1 continues with its crossing row --
    all-gathered and filled by every rank (one node per tile: tiny); each rank takes its part
    of the levels, and its ghost rows start at max(d, level) of the neighbour's tile they lie
    in.  Returns the levels raster of the block (kept alive by the caller until the solve)."""
    import torch

    rank, world = comm.rank, comm.world
    top, bottom = rank > 0, rank < world - 1
    h, cols = z_local.shape
    t = solver.TILE
    solver.hub_prepare(z_local, w, flags)
    # the row a neighbour pins is `ghost` rows into my block; what it gets is my d there
    comm.swap(w[2 * ghost - 1] if top else None, w[h - 2 * ghost] if bottom else None,
              w[0] if top else None, w[h - 1] if bottom else None)
    d_top = w[0].clone() if top else None
    d_bot = w[h - 1].clone() if bottom else None
    mine = solver.hub_raster(z_local)
    ch, cw = mine.shape
    # my part of the stack: down to the node row of the tile that holds the row rank+1 pins
    hi = (h - 2 * ghost - 1) // t if bottom else (ch - 1) // 2 - 1
    keep = 2 * hi + 2 if bottom else ch
    counts = torch.tensor([keep], dtype=torch.int64, device=z_local.device)
    rows = torch.stack(comm.all_gather(counts)).cpu()[:, 0].tolist()
    padded = torch.full((max(rows), cw), HUB_BIG, dtype=mine.dtype, device=mine.device)
    padded[:keep] = mine[:keep]
    parts = comm.all_gather(padded)
    stack = torch.cat([p[:n] for p, n in zip(parts, rows)]).contiguous()
    filled = torch.empty_like(stack)
    # (the library fills a raster of this kind from a hub start of its own when it is large)
    solver.fill(stack, filled, 0.0, backend.FILL_INIT | backend.FILL_NO_VERIFY)
    off = [sum(rows[:r]) for r in range(world)]
    levels = torch.full_like(mine, HUB_BIG)
    levels[:keep] = filled[off[rank]:off[rank] + keep]
    # the tile rows underneath my part (overlap rows that the next rank's tiles stand for in the
    # stack): one step down the block's own crossings from the last row that has a level
    for j in range(hi + 1, (ch - 1) // 2):
        up = levels[2 * j - 1, 1::2]
        up = torch.where(torch.isnan(up), torch.full_like(up, float("-inf")), up)   # an outlet above
        node = mine[2 * j + 1, 1::2]
        lev = torch.maximum(node, torch.maximum(up, mine[2 * j, 1::2]))
        levels[2 * j + 1, 1::2] = torch.where(torch.isnan(node), node, lev)

    def ghost_bound(d_row, lev_row, z_row):
        # start value of a ghost row: max(d, level of the neighbour's tile the cell lies in);
        # an outlet tile (NaN level) needs no level, nodata stays nodata, the two cells on the
        # raster's first and last column are the raster ring
        lev = lev_row[1::2].repeat_interleave(t)[:cols - 2]
        lev = torch.where(lev >= HUB_BIG, torch.full_like(lev, float("inf")), lev)
        out = d_row.clone()
        inner = out[1:-1]
        raised = torch.where(torch.isnan(lev) | torch.isnan(inner), inner, torch.maximum(inner, lev))
        out[1:-1] = torch.where(raised >= HUB_BIG, torch.full_like(raised, float("inf")), raised)
        out[0], out[-1] = z_row[0], z_row[-1]
        return out

    if top:        # my row 0 = the upper neighbour's row h' - 2 ghost, in its tile row `hi` there
        w[0] = ghost_bound(d_top, filled[off[rank] - 1], z_local[0])
    if bottom:     # my last row = the lower neighbour's row 2 ghost - 1
        tj = (2 * ghost - 2) // t
        w[h - 1] = ghost_bound(d_bot, filled[off[rank + 1] + 2 * tj + 1], z_local[h - 1])
    solver.set_hub_levels(levels)
    return levels


def _block_rows(world, z_local, comm, ghost):
    """Rows of every rank's local array (the hub start wants blocks that hold the rows their
    neighbours pin well inside a tile row of their own)."""
    import torch
    mine = torch.tensor([z_local.shape[0]], dtype=torch.int64, device=z_local.device)
    return torch.stack(comm.all_gather(mine)).cpu()[:, 0].tolist()


def _comm_for(rank, world, group, comm):
    if comm is None:
        comm = DistComm(group)
    if (comm.rank, comm.world) != (rank, world):
        raise ValueError(f"rank/world ({rank}/{world}) do not match the communicator "
                         f"({comm.rank}/{comm.world})")
    return comm


def sinkfill_distributed(z_local, rank, world, solver, eps=0.0, w_out=None,
                         max_exchanges=100000, group=None, coarse_block=None, d8_out=None,
                         ghost=1, comm=None, hub=True):
    """Sink fill of a row-block partitioned raster.

    ``z_local``: torch tensor, local rows incl. ghost rows (see
    :func:`local_range`), float32.  Returns (w_local, info): ``w_local`` has
    the same shape, ghost rows holding the neighbours' final values.  ``d8_out``
    (uint8, same shape): also receives the D8 codes of the filled block, written by
    the last verifying pass (rows of ghost rows are meaningless, as in
    :func:`d8_distributed`).  ``ghost``: rows of overlap the local array was cut with
    (:func:`local_range`, :func:`ghost_rows`); only the outermost is pinned.
    ``group``: a ``torch.distributed`` process group (default: the world); ``comm``:
    a ready communicator instead (:class:`DistComm`, or a :class:`ThreadWorld` rank).
    ``info``: tile visits (``unchanged`` of them found nothing to lower), exchanges,
    verifying passes, ``solves`` -- (phase, tile visits) of every local solve, in order --
    ``async_fallbacks``: local solves whose persistent launch ran out of its wall-clock
    budget and were finished by the round driver, and ``shared_gpu_solves``: solves whose
    launch did not get all its workgroups resident at once (another process on the GPU) and
    ran with those it got."""
    import torch

    top, bottom = rank > 0, rank < world - 1
    w = torch.empty_like(z_local) if w_out is None else w_out
    flags = backend.FILL_INIT
    if top:
        flags |= backend.FILL_GHOST_TOP
    if bottom:
        flags |= backend.FILL_GHOST_BOTTOM
    sliced = world > 1
    if coarse_block is None:
        # (every rank fills the whole stacked coarse raster; at 8 x 16384^2 that is
        # 8192 x 1024 cells and 1.6 ms, and 32 x 32 blocks would cost more in the fine
        # solve than they save here: 16.1 against 15.4 ms predicted)
        coarse_block = COARSE_BLOCK
    tally = {"tile_visits": 0, "unchanged": 0, "solves": [], "fallbacks": 0, "shared": 0,
             "start": "inf"}

    def solve(phase, fill_flags, **kw):
        v, lowered, pending = solver.fill(z_local, w, eps, fill_flags, **kw)
        st = getattr(solver, "last_stats", None) or {}
        tally["tile_visits"] += int(v)
        tally["unchanged"] += int(st.get("visits_unchanged", 0))
        tally["solves"].append((phase, int(v)))
        tally["fallbacks"] += int(bool(st.get("async_timed_out", 0)))
        tally["shared"] += int(bool(st.get("partial_residency", 0)))
        return lowered, pending

    keep = None
    if world > 1:
        comm = _comm_for(rank, world, group, comm)
        if eps == 0.0 and hub and hasattr(solver, "hub_prepare") and \
                all(n >= 4 * ghost + 2 * 62 for n in _block_rows(world, z_local, comm, ghost)):
            keep = hub_start(z_local, w, comm, solver, flags, ghost)
            flags |= backend.FILL_GHOST_GIVEN
            tally["start"] = "hub"
        elif eps == 0.0 and coarse_block:
            keep = coarse_start(z_local, comm, solver, coarse_block, ghost)
            solver.set_coarse_start(keep[0], coarse_block, keep[1])
            tally["start"] = "blockmax"
    _, pending = solve("first", flags | backend.FILL_NO_VERIFY, sliced=sliced)
    del keep                                       # (alive until the solve has consumed them)
    exchanges = verifications = 0
    seam = _Seam(torch, w, top, bottom) if world > 1 else None
    # Deferred loop (device solvers): no host decision stands between a seam exchange and the
    # correcting solve behind it.  The solve is enqueued at once -- seeded on the device from the
    # ghost-changed words, a no-op when they are clear and nothing is queued -- and the vote is
    # looked at while it runs; a vote of "all at rest" means that solve found nothing to do.
    defer = world > 1 and getattr(solver, "can_defer", False) and \
        os.environ.get("HDEM_PARTITION_DEFER", "1") != "0" and \
        "HDEM_FILL_SYNC" not in os.environ             # (a deferred call is an asynchronous launch)
    tally["deferred"] = 0
    while world > 1:
        if defer:
            _exchange_and_vote_deferred(comm, solver, w, top, bottom, pending, seam, ghost)
            act = backend.FILL_WARM | backend.FILL_RESUME | backend.FILL_NO_VERIFY
            act |= backend.FILL_ACT_TOP if top else 0
            act |= backend.FILL_ACT_BOTTOM if bottom else 0
            solver.fill_deferred(z_local, w, eps, act, seam.word)
            tally["deferred"] += 1
            pending = -1                               # (on the device: seam.word[0])
            if seam.event is not None:
                seam.event.synchronize()
            any_busy, ch_top, ch_bot = bool(seam.vote_host[0]), False, False
        else:
            any_busy, ch_top, ch_bot = _exchange_and_vote(comm, torch, w, top, bottom, pending,
                                                          seam, ghost)
        exchanges += 1
        if exchanges >= max_exchanges:
            raise RuntimeError("distributed sink fill did not converge")
        if not any_busy:
            # every rank is at rest: certify the whole block (round driver, all tiles
            # due); resume only if some rank still found something to lower
            lowered, pending = solve("verify", backend.FILL_WARM | backend.FILL_SYNC_ONLY,
                                     d8=d8_out)
            verifications += 1
            seam.word[0] = int(lowered)
            if int(comm.all_reduce_max(seam.word[:1].clone()).item()) == 0:
                break
            pending = 0                                # (the round driver leaves nothing queued)
            continue
        if not defer and (ch_top or ch_bot or pending > 0):
            # next slice: the tiles left queued plus those next to a replaced ghost row
            act = backend.FILL_WARM | backend.FILL_RESUME | backend.FILL_NO_VERIFY
            act |= backend.FILL_ACT_TOP if ch_top else 0
            act |= backend.FILL_ACT_BOTTOM if ch_bot else 0
            _, pending = solve("correct", act, sliced=sliced)
    if world == 1:
        solve("verify", backend.FILL_WARM | backend.FILL_SYNC_ONLY, d8=d8_out)
    return w, {"tile_visits": tally["tile_visits"], "visits_unchanged": tally["unchanged"],
               "exchanges": exchanges, "verifications": verifications,
               "solves": tally["solves"], "async_fallbacks": tally["fallbacks"],
               "shared_gpu_solves": tally["shared"], "start_values": tally["start"],
               "deferred_solves": tally["deferred"]}


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


def halo_exchange(owned, halo, rank, world, group=None, comm=None):
    """Owned rows plus ``halo`` rows of each neighbour (one batched isend/irecv
    group; SURVEY 8e: D8 / box mean 1 row, quadratic 7 rows per pass).  Returns
    (extended tensor, rows on top that belong to rank-1, rows at the bottom that
    belong to rank+1).  Ranks must own at least ``halo`` rows."""
    import torch

    top, bottom = rank > 0, rank < world - 1
    if owned.shape[0] < halo:
        raise ValueError(f"rank {rank} owns {owned.shape[0]} rows, fewer than the halo of {halo}")
    if world == 1:
        return owned, 0, 0
    comm = _comm_for(rank, world, group, comm)
    # received straight into the extended block: no separate receive rows, no torch.cat
    rows = owned.shape[0] + halo * (int(top) + int(bottom))
    ext = torch.empty((rows,) + tuple(owned.shape[1:]), dtype=owned.dtype, device=owned.device)
    first = halo if top else 0
    ext[first:first + owned.shape[0]].copy_(owned)
    comm.swap(owned[:halo] if top else None, owned[-halo:] if bottom else None,
              ext[:halo] if top else None, ext[rows - halo:] if bottom else None)
    return ext, halo if top else 0, halo if bottom else 0


def groves_distributed(img_owned, groves_owned, rank, world, solver, iterations=3,
                       window_size=15, threshold=1.5, group=None, comm=None):
    """``GrovesCorrectionsIter`` on a row-block partitioned raster: one exchange of
    ``iterations * (window_size // 2)`` rows each way, then the fused passes run on the
    extended block and the overlap is recomputed instead of exchanged again.  Pass k
    is right from row k * (window_size // 2) of the extended block on, so after the
    last pass the owned rows are; the raster's own first and last rows keep the
    reference's untouched border ring because they are the block's."""
    halo = iterations * (window_size // 2)
    img, t, b = halo_exchange(img_owned, halo, rank, world, group, comm)
    mask, _, _ = halo_exchange(groves_owned, halo, rank, world, group, comm)
    out = solver.groves(img, mask, window_size, threshold, iterations)
    return out[t:out.shape[0] - b]


def boxmean_distributed(x_owned, rank, world, solver, do_round=True, group=None, comm=None):
    """``PostProcessingFinal`` (3 x 3 mean, reflect at the raster border, optional
    rounding) on a row-block partitioned raster: one ghost row each way."""
    x, t, b = halo_exchange(x_owned, 1, rank, world, group, comm)
    out = solver.boxmean(x, do_round)
    return out[t:out.shape[0] - b]


def owned_slice(rank, world, ghost=1):
    """Slice of the local array that holds the owned rows."""
    return slice(ghost if rank > 0 else 0, -ghost if rank < world - 1 else None)
