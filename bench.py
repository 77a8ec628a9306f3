#!/usr/bin/env python3
"""
bench.py -- the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--config 4|5]

Metric (BASELINE.json): Mcells/s of SinkFill (to convergence) + D8FlowDirection
on a 16384^2 float32 DEM, inputs resident in HBM when the timed region starts,
outputs left in HBM.  One "step" = one full sink fill + D8 of the raster.

N > 1: the raster is N*S rows x S columns, row-block partitioned, S rows per rank --
weak scaling -- with seam exchanges between local solves (hydrodem_amd/partition.py),
one process per GPU over RCCL.  --config 4 / --config 5 run BASELINE's named multi-GPU
configurations instead: the 32768^2 raster on 4 GPUs (8192 x 32768 per rank) and the
65536^2 mosaic on 8 (8192 x 65536 per rank); with another --gpus N the same raster is cut
into N row blocks (then it is a strong-scaling point and the line says so).  Under torch.distributed.run (RANK / WORLD_SIZE in the
environment) this process is one rank; started plainly (`python bench.py --gpus N`) it
starts its N ranks itself, as fresh child processes, before anything touches a GPU.
HDEM_REHEARSE=1 puts every rank on cuda:0 over gloo: the N > 1 code path on a one-GPU
box, never a performance number.

One JSON line on stdout (rank 0) with, besides the contract keys:
  roofline      dominant kernel = fill_async_kernel (the sink-fill tile relaxation):
                  achieved     visit bytes / its HIP-event time over the timed steps: 12 B per
                               cell of a visit that writes its tile back (Z in, W in, W out),
                               8 B per cell of one that finds nothing to lower
                  frac         achieved / 8 TB/s;  frac_of_copy: / the copy rate measured here
                  useful_frac  the END-TO-END floor of a fill, 8 B per raster cell (Z in once,
                               W out once), over the whole step time, / 8 TB/s -- what the
                               schedule's revisits cost shows up here, not in `frac`
                  traffic      HBM bytes per launch from rocprofv3 --pmc passes of this
                               command (profiles/), null when they were taken from another
                               build of the kernel
  kernels       the certifying pass (also writes the D8 codes), the init kernel, the coarse
                pre-solve, each against both roofs
  filters       the other operators of the scope table on the same raster (outside the timed
                region), with the NumPy / SciPy oracle timed beside them on a crop (CPU-B)
  config2       BASELINE configs[1]: the same step at 4096^2
  cpu_baseline  the C priority-flood oracle + C D8 on the FULL workload raster, one core;
                the NumPy Jacobi oracle on a crop beside it (size-dependent, context only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
VISIT_BYTES_WRITING = 12    # per tile cell: Z in + W in + W out
VISIT_BYTES_UNCHANGED = 8   # a visit that lowers nothing skips the write-back
FLOOR_BYTES_PER_CELL = 8    # a whole fill: read Z once, write W once
FILL_KERNEL = "fill_async_kernel<false, 0>"
TRAFFIC_RECORD = os.path.join(ROOT, "profiles", "r03_fill_traffic.json")
FILL_SOURCES = ("hydrodem_amd/csrc/hdem_sinkfill.hip", "hydrodem_amd/csrc/hdem_internal.h")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--size", type=int, default=16384)
    p.add_argument("--config", type=int, default=0, choices=(0, 4, 5),
                   help="BASELINE configs[3] / configs[4]: 32768^2 on 4 GPUs, 65536^2 on 8 "
                        "(sets the raster; --gpus defaults to the configuration's)")
    p.add_argument("--cpu-sample", type=int, default=1024,
                   help="edge of the crop the NumPy Jacobi oracle is timed on (0 = no CPU lines)")
    p.add_argument("--no-filters", action="store_true",
                   help="skip the untimed extras (filters, config2)")
    return p.parse_args()


# --------------------------------------------------------------------------
# N > 1 without an external launcher
# --------------------------------------------------------------------------
def launch_ranks(n):
    """Start the N ranks as children of this (GPU-free) process and relay rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # as torch.distributed.run does: N ranks with a full OpenMP team each oversubscribe
        # the host (measured: 246 against 8.8 ms per step in the two-rank rehearsal)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    line, failed = b"", None
    try:
        pending = set(range(n))
        while pending and failed is None:
            for rank in sorted(pending):
                code = procs[rank].poll()
                if code is None:
                    continue
                pending.discard(rank)
                if code != 0:
                    failed = (rank, code)
            if pending and failed is None:
                time.sleep(0.2)
        line = procs[0].stdout.read() if failed is None else b""
    finally:
        for p in procs:                        # a failed rank leaves the others in a collective
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    if failed is not None:
        raise SystemExit(f"bench.py: rank {failed[0]} exited with status {failed[1]}")
    # (only the record: gloo announces its connections on stdout in the rehearsal)
    for text in line.decode().splitlines():
        if text.startswith('{"metric"'):
            sys.stdout.write(text + "\n")
    sys.stdout.flush()


# --------------------------------------------------------------------------
# measurement helpers
# --------------------------------------------------------------------------
def kernel_source_hash():
    h = hashlib.sha256()
    for rel in FILL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profiled_traffic(size):
    """HBM bytes per launch of the fill kernel from the committed rocprofv3 --pmc passes of
    this command (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be collected from
    inside this process; tools/record_traffic.py condenses them into TRAFFIC_RECORD together
    with the hash of the kernel's source).  Null -- with the reason -- when the record is
    missing, was taken at another size or from another build of the kernel."""
    if not os.path.exists(TRAFFIC_RECORD):
        return None, {"traffic_note": "no PMC record committed"}
    rec = json.load(open(TRAFFIC_RECORD))
    meta = {"traffic_profile_head": rec.get("head"), "traffic_kernel_hash": rec.get("kernel_hash"),
            "traffic_source": rec.get("source")}
    if rec.get("kernel_hash") != kernel_source_hash():
        meta["traffic_note"] = "PMC record is of another build of the kernel: not reported"
        return None, meta
    if rec.get("size") != size:
        meta["traffic_note"] = f"PMC record is of size {rec.get('size')}"
        return None, meta
    return rec["bytes_per_launch"], meta


def roofs(gbs, copy_gbs):
    return {"achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "frac_of_copy": gbs / copy_gbs if copy_gbs else None}


def timed(ctx, fn, reps=3):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cpu_baseline(z, jacobi_edge):
    """The oracle timed as a baseline (checker code, never the product): C priority flood +
    C D8 on the whole workload raster; the NumPy Jacobi oracle -- the literal "NumPy CPU
    reference" of north_star -- only on a crop, because its cost grows with the raster's
    diameter (sweeps to convergence), so its Mcells/s cannot be carried to another size."""
    import oracle
    from oracle import c_oracle
    t = time.perf_counter()
    w = c_oracle.sinkfill_pflood(z)
    c_oracle.d8(w)
    dt = time.perf_counter() - t
    out = {"value": z.size / dt / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
           "sample": f"the whole {z.shape[0]}x{z.shape[1]} workload raster: C priority-flood sink "
                     f"fill + C D8 (oracle/hdem_oracle.c), single thread, {dt:.1f} s",
           "host_cores": os.cpu_count()}
    del w
    c = min(jacobi_edge, *z.shape)
    zc = np.ascontiguousarray(z[:c, :c])
    t = time.perf_counter()
    wj, sweeps = oracle.sinkfill_jacobi(zc)
    oracle.d8_flow_direction(wj)
    dtj = time.perf_counter() - t
    assert np.array_equal(wj, c_oracle.sinkfill_pflood(zc))
    out["numpy_jacobi"] = {"value": zc.size / dtj / 1e6, "unit": "Mcells/s", "cores": 1,
                           "sample": f"{c}x{c} crop, NumPy Jacobi sink fill to convergence "
                                     f"({sweeps} sweeps) + NumPy D8, {dtj:.1f} s",
                           "note": "size-dependent: sweeps grow with the raster's diameter; "
                                   "not comparable with the 16384^2 metric"}
    return out


def cpu_b(z, groves):
    """BASELINE.md section 3, CPU-B: the NumPy / SciPy oracle of groves x3 and of the final
    3x3 mean + rounding, one thread, on crops of the workload (both are local operators with
    a size-independent cost per cell)."""
    import oracle
    g = 2048
    zc, gc = np.ascontiguousarray(z[:g, :g]), np.ascontiguousarray(groves[:g, :g])
    t = time.perf_counter()
    oracle.groves_exact64(zc, gc, 3, 15, 1.5)
    dt_g = time.perf_counter() - t
    b = 4096
    zb = np.ascontiguousarray(z[:b, :b])
    t = time.perf_counter()
    oracle.boxmean3_round(zb)
    dt_b = time.perf_counter() - t
    return ({"Mcells_per_s": zc.size / dt_g / 1e6, "cores": 1,
             "sample": f"{g}x{g} crop, NumPy separable quadratic + groves algebra x3, {dt_g:.1f} s"},
            {"Mcells_per_s": zb.size / dt_b / 1e6, "cores": 1,
             "sample": f"{b}x{b} crop, scipy.ndimage.convolve + np.around, {dt_b:.1f} s"})


def filter_paths(B, ctx, zd, scratch, S, copy_gbs, with_cpu, reps=8):
    """The other operators of the scope table on the same raster, outside the timed region
    (they are not part of the headline metric): warm call, then ``reps`` timed."""
    import hdem_synth
    res = {}

    def line(ms, bytes_per_cell=None):
        out = {"ms": ms, "Mcells_per_s": S * S / ms / 1e3}
        if bytes_per_cell:
            out["algorithmic_bytes_per_cell"] = bytes_per_cell
            out.update(roofs(bytes_per_cell * S * S / ms / 1e6, copy_gbs))
        return out

    groves_host = hdem_synth.synth_groves(S, S)
    mask = B.DeviceRaster.from_host(groves_host, ctx=ctx)
    pong = B.DeviceRaster.empty(zd.shape, np.float32, ctx)
    res["groves_x3"] = line(timed(ctx, lambda: B.groves_dev(zd, mask, iterations=3, out=scratch,
                                                            scratch=pong), reps), 27)
    # BASELINE configs[2]: the full chain -- groves x3, sink fill, D8 -- device resident
    filled = B.DeviceRaster.empty(zd.shape, np.float32, ctx)
    codes = B.DeviceRaster.empty(zd.shape, np.uint8, ctx)

    def chain():
        B.groves_dev(zd, mask, iterations=3, out=scratch, scratch=pong)
        B.sinkfill_d8_dev(scratch, out=filled, codes=codes)
    res["full_chain_groves_fill_d8"] = line(timed(ctx, chain, reps))
    # the chain's end-to-end floor: groves x3 27 B/cell, fill 8 (Z in, W out), D8 codes 1
    res["full_chain_groves_fill_d8"].update(
        floor_bytes_per_cell=36,
        useful_frac=36 * S * S / res["full_chain_groves_fill_d8"]["ms"] / 1e6 / HBM_PEAK_GBS)
    res["d8_alone"] = line(timed(ctx, lambda: B.d8_dev(filled, out=codes), reps), 5)
    for r in (mask, pong, filled, codes):
        r.free()
    res["boxmean3_round"] = line(timed(ctx, lambda: B.boxmean3_dev(zd, out=scratch), reps), 8)
    # the raster I/O seam: groves x3 with host arrays at both ends, PCIe and
    # host copies included -- band stream, 2048-row bands, one stream and one host thread
    # per slot
    from hydrodem_amd import streaming as St
    z_host = zd.to_host()
    streamed = np.empty_like(z_host)
    with St.BandStream(z_host.shape, band_rows=2048, depth=3, **St.groves_op(3)) as bs:
        bs.run([z_host, groves_host], streamed)
        t0 = time.perf_counter()
        bs.run([z_host, groves_host], streamed)
        ms = (time.perf_counter() - t0) * 1e3
    res["groves_x3_host_to_host_stream"] = {
        "ms": ms, "Mcells_per_s": S * S / ms / 1e3,
        "note": "host ndarray in, host ndarray out: pinned band buffers, H2D, 3 fused passes, "
                "D2H; bounded by the host copies (9 B/cell in, 4 B/cell out through PCIe)"}
    del streamed
    if with_cpu:
        res["groves_x3"]["cpu_numpy_oracle"], res["boxmean3_round"]["cpu_scipy_oracle"] = \
            cpu_b(z_host, groves_host)
    del z_host, groves_host
    ctx.profile(True)
    ctx.profile_reset()
    # algorithmic bytes per raster cell, stage by stage: reference level 8; real
    # forward transform 4 in + 8 out, its Hermitian completion 8; two quadrant magnitudes
    # (a quarter of the cells each) 12 / 2; hollow mean 9 per quadrant cell x 2 passes x 2
    # quadrants / 4; masks ~1; inverse: two row passes 16 + 16, transpose 16, transpose + abs 12
    res["fourier_destripe"] = line(timed(ctx, lambda: B.fourier_destripe_dev(zd, out=scratch), reps),
                                   8 + 12 + 8 + 6 + 9 + 1 + 60)
    n_calls = reps + 1
    for name, kid in (("rocfft_c2c", B.K_FFT), ("detect", B.K_FOURIER_DETECT),
                      ("mask", B.K_FOURIER_MASK), ("pointwise", B.K_FOURIER_POINT)):
        res["fourier_destripe"][name + "_ms"] = ctx.profile_get(kid)["ms"] / n_calls
    ctx.profile(False)
    # SURVEY 8d's second input variant: the same DEM in integer metres (large flats, ties)
    hs = np.round(zd.to_host())
    hd_ = B.DeviceRaster.from_host(hs, ctx=ctx)
    codes = B.DeviceRaster.empty(zd.shape, np.uint8, ctx)
    info = {}
    res["sinkfill_d8_srtm_variant"] = line(timed(
        ctx, lambda: info.update(B.sinkfill_d8_dev(hd_, out=scratch, codes=codes)[2]), reps))
    res["sinkfill_d8_srtm_variant"]["tile_visits"] = info.get("tile_visits")
    codes.free()
    hd_.free()
    # lagoon branch on the integer-metre variant of the raster with voids
    hs[::97, ::89] = -32768.0
    hd_ = B.DeviceRaster.from_host(hs, ctx=ctx)
    del hs
    ctx.profile(True)
    ctx.profile_reset()
    l_fixed = B.DeviceRaster.empty(zd.shape, np.float32, ctx)
    l_mask = B.DeviceRaster.empty(zd.shape, np.uint8, ctx)

    def lagoons():
        ctx.check(ctx.lib.hdem_lagoons_detection_f32_dev(ctx.handle, hd_.ptr, S, S, l_fixed.ptr,
                                                         scratch.ptr, l_mask.ptr))
    # void repair 8 + majority 8 + fused tidy 9 (with the positive mask) B per cell
    res["lagoons_detection"] = line(timed(ctx, lagoons, reps), 25)
    res["lagoons_detection"]["majority_ms"] = ctx.profile_get(B.K_MAJORITY)["ms"] / n_calls
    res["lagoons_detection"]["other_kernels_ms"] = ctx.profile_get(B.K_LAGOON)["ms"] / n_calls
    ctx.profile(False)
    for r in (hd_, l_fixed, l_mask):
        r.free()
    return res


def config2(B, ctx, steps=10):
    """BASELINE configs[1]: 4096 x 4096, sink fill + D8 on one GPU."""
    import hdem_synth
    s = 4096
    return fill_line(B, ctx, hdem_synth.synth_dem(s, s),
                     f"BASELINE configs[1]: {s}x{s} float32 synthetic DEM, SinkFill eps=0 + D8", steps)


def fill_line(B, ctx, z, workload, steps=10):
    """Fill + D8 of one raster, resident: time, visits, and the fine launch against the roof
    the way the headline line states it."""
    zd = B.DeviceRaster.from_host(z, ctx=ctx)
    wd = B.DeviceRaster.empty(zd.shape, np.float32, ctx)
    dd = B.DeviceRaster.empty(zd.shape, np.uint8, ctx)
    stats = []
    call = lambda: stats.append(B.sinkfill_d8_dev(zd, out=wd, codes=dd)[2])
    call()
    ctx.synchronize()
    del stats[:]
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    kt = ctx.profile_get(B.K_FILL_TILE)
    ctx.profile(False)
    for r in (zd, wd, dd):
        r.free()
    cells = z.size
    info = stats[-1]
    unchanged = sum(s_["visits_unchanged"] - s_.get("flat_unchanged", 0) for s_ in stats) * 62 * 62
    visit_bytes = VISIT_BYTES_WRITING * kt["units"] - \
        (VISIT_BYTES_WRITING - VISIT_BYTES_UNCHANGED) * min(unchanged, kt["units"])
    gbs = visit_bytes / max(kt["ms"], 1e-9) / 1e6
    return {"workload": workload, "steps": steps,
            "ms_per_step": ms, "Mcells_per_s": cells / ms / 1e3,
            "tile_visits_per_step": info.get("tile_visits"), "tiles": info.get("tiles"),
            "visits_per_tile": (info.get("tile_visits") or 0) / max(info.get("tiles") or 1, 1),
            "fill_async_ms_per_step": kt["ms"] / steps,
            "algorithmic_bytes_per_cell_visit": VISIT_BYTES_WRITING,
            "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "useful_frac": FLOOR_BYTES_PER_CELL * cells / ms / 1e6 / HBM_PEAK_GBS}


def batch_line(B, ctx, n=16, edge=1201, reps=5):
    """Sixteen SRTM-sized tiles (1201^2), resident: one fill + D8 call each, against all of them
    in ONE call on a canvas with nodata gutters (what HydroConditioning.apply_batch launches;
    same bits: nodata's neighbours are pinned like a raster ring)."""
    import hdem_synth
    tiles = [hdem_synth.synth_dem(edge, edge) + np.float32(k) for k in range(n)]
    zs = [B.DeviceRaster.from_host(t, ctx=ctx) for t in tiles]
    ws = [B.DeviceRaster.empty(t.shape, np.float32, ctx) for t in tiles]
    ds = [B.DeviceRaster.empty(t.shape, np.uint8, ctx) for t in tiles]

    def one_by_one():
        for z, w, d in zip(zs, ws, ds):
            B.sinkfill_d8_dev(z, out=w, codes=d)
    ms_each = timed(ctx, one_by_one, reps)
    for r in zs + ws + ds:
        r.free()
    canvas = np.full((n * (edge + 1) - 1, edge), np.nan, np.float32)
    for k, t in enumerate(tiles):
        canvas[k * (edge + 1):k * (edge + 1) + edge] = t
    zc = B.DeviceRaster.from_host(canvas, ctx=ctx)
    wc = B.DeviceRaster.empty(canvas.shape, np.float32, ctx)
    dc = B.DeviceRaster.empty(canvas.shape, np.uint8, ctx)
    ms_batch = timed(ctx, lambda: B.sinkfill_d8_dev(zc, out=wc, codes=dc), reps)
    for r in (zc, wc, dc):
        r.free()
    cells = n * edge * edge
    return {"workload": f"{n} rasters of {edge}x{edge}, SinkFill eps=0 + D8, resident",
            "one_call_per_raster_ms": ms_each, "one_canvas_ms": ms_batch,
            "Mcells_per_s_one_by_one": cells / ms_each / 1e3,
            "Mcells_per_s_canvas": cells / ms_batch / 1e3}


def config1(B, ctx):
    """BASELINE configs[0]'s raster on the GPU path: the reference's own 519 x 508 study-area
    DEM (cguerrero/resources/images/final_dem.tif, committed as tests/golden/ref_rasters.npz
    by tests/golden/make_golden.py), integer metres with ties everywhere."""
    path = os.path.join(ROOT, "tests", "golden", "ref_rasters.npz")
    if not os.path.exists(path):
        return None
    z = np.ascontiguousarray(np.load(path)["final_dem"], dtype=np.float32)
    return fill_line(B, ctx, z, "BASELINE configs[0]'s raster: the reference's 519x508 final_dem.tif, "
                                "SinkFill eps=0 + D8 (latency-bound: 81 tiles)", 20)


# --------------------------------------------------------------------------
def main():
    a = parse()
    S, N = a.size, a.gpus
    if a.config:
        if N == 1 and "WORLD_SIZE" not in os.environ and "--gpus" not in " ".join(sys.argv):
            N = a.gpus = {4: 4, 5: 8}[a.config]
            sys.argv += ["--gpus", str(N)]
    if N > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(N)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != N:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")

    import hdem_synth               # inputs; oracle/ is only touched by the cpu_* legs
    from hydrodem_amd import backend as B

    step_stats = []                 # per timed step: visits / unchanged / ...
    H, Wd = S, S
    if N == 1:
        ctx = B.context(0)
        z = hdem_synth.synth_dem(S, S)
        zd = B.DeviceRaster.from_host(z, ctx=ctx)
        wd = B.DeviceRaster.empty(z.shape, np.float32, ctx)
        dd = B.DeviceRaster.empty(z.shape, np.uint8, ctx)

        def step():
            # fill + D8 in one call: the certifying pass of the fill writes the codes
            _, _, st = B.sinkfill_d8_dev(zd, out=wd, codes=dd)
            step_stats.append(st)

        def sync():
            ctx.synchronize()

        def reduce_max(x):
            return x
    else:
        import torch
        import torch.distributed as dist
        from hydrodem_amd import partition as P
        rehearse = os.environ.get("HDEM_REHEARSE") == "1"
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if rehearse else "nccl")
        if a.config:
            H = Wd = {4: 32768, 5: 65536}[a.config]     # the named raster, cut into N blocks
        else:
            H, Wd = N * S, S                            # weak scaling: S rows per rank
        ghost = P.ghost_rows(world, H)                  # one tile row of overlap per seam
        g0, g1, _, _ = P.local_range(rank, world, H, ghost)
        z = hdem_synth.synth_dem(H, Wd, row0=g0, rows=g1 - g0)
        dev = torch.device("cuda", local_rank)
        zt = torch.from_numpy(z).to(dev)
        wt = torch.empty_like(zt)
        dt_ = torch.empty(zt.shape, dtype=torch.uint8, device=dev)
        solver = P.HipLocalSolver(local_rank)
        comm = P.DistComm()                             # its staging rows live across steps
        ctx = solver.ctx

        def step():
            # (the last verifying pass of the fill writes the D8 codes of the block)
            _, st = P.sinkfill_distributed(zt, rank, world, solver, w_out=wt, d8_out=dt_,
                                           ghost=ghost, comm=comm)
            st["tiles"] = (solver.last_stats or {}).get("tiles")
            step_stats.append(st)

        def sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

    for _ in range(a.warmup):
        step()
    del step_stats[:]
    ctx.profile(True)
    ctx.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = reduce_max(time.perf_counter() - t0)

    kt = ctx.profile_get(B.K_FILL_TILE)
    kc = ctx.profile_get(B.K_FILL_COARSE)
    kb = ctx.profile_get(B.K_BLOCKMAX)
    kr = ctx.profile_get(B.K_FILL_ROUND)
    ki = ctx.profile_get(B.K_FILL_INIT)
    kh = ctx.profile_get(B.K_FILL_HUB)
    ctx.profile(False)
    last = step_stats[-1]
    per_rank = None
    if N > 1:
        # every rank's share of the step, gathered after the timed region
        mine = torch.tensor([kt["ms"], kc["ms"] + kb["ms"] + kh["ms"], kr["ms"], ki["ms"],
                             float(sum(s["tile_visits"] for s in step_stats)),
                             float(sum(s["visits_unchanged"] for s in step_stats)),
                             float(last["tiles"] or 0), float(last["exchanges"]),
                             float(sum(s_["async_fallbacks"] for s_ in step_stats)),
                             float(sum(s_["shared_gpu_solves"] for s_ in step_stats))],
                            dtype=torch.float64)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        if rehearse:
            dist.all_gather(parts, mine)
        else:
            dev_parts = [p.to(dev) for p in parts]
            dist.all_gather(dev_parts, mine.to(dev))
            parts = [p.cpu() for p in dev_parts]
        k = float(a.steps)
        per_rank = {"fill_async_ms_per_step": [float(p[0]) / k for p in parts],
                    "coarse_start_ms_per_step": [float(p[1]) / k for p in parts],
                    "certify_ms_per_step": [float(p[2]) / k for p in parts],
                    "init_ms_per_step": [float(p[3]) / k for p in parts],
                    "tile_visits_per_step": [float(p[4]) / k for p in parts],
                    "visits_unchanged_per_step": [float(p[5]) / k for p in parts],
                    "tiles": [int(p[6]) for p in parts],
                    "exchanges": [int(p[7]) for p in parts],
                    # local solves finished by the round driver (budget), and solves whose
                    # persistent launch shared the GPU with another process (0 / 0 on a node
                    # with one rank per GPU)
                    "async_fallbacks": [int(p[8]) for p in parts],
                    "shared_gpu_solves": [int(p[9]) for p in parts]}

    if rank == 0:
        copy_gbs = B.copy_rate(ctx)
        cells_total = H * Wd
        ms_per_step = elapsed / a.steps * 1e3
        # rank 0's fill launches: cells of all window visits (profile units; visits of flat
        # tiles read ~500 cells and are not in them), of which the unchanged ones skipped the
        # write-back
        ft2 = 62 * 62
        unchanged_cells = sum(s["visits_unchanged"] - s.get("flat_unchanged", 0)
                              for s in step_stats) * ft2
        visit_bytes = VISIT_BYTES_WRITING * kt["units"] - \
            (VISIT_BYTES_WRITING - VISIT_BYTES_UNCHANGED) * min(unchanged_cells, kt["units"])
        fill_gbs = visit_bytes / max(kt["ms"], 1e-9) / 1e6
        launches = max(kt["launches"], 1)
        traffic, traffic_meta = profiled_traffic(S) if N == 1 else (None, {})
        useful_gbs = FLOOR_BYTES_PER_CELL * cells_total / N / ms_per_step / 1e6   # per GPU
        roofline = {"bound": "hbm", "kernel": FILL_KERNEL, "peak": HBM_PEAK_GBS}
        roofline.update(roofs(fill_gbs, copy_gbs))
        roofline.update({
            "copy_rate_measured": copy_gbs,
            "useful_frac": useful_gbs / HBM_PEAK_GBS,
            "useful_frac_of_copy": useful_gbs / copy_gbs,
            "end_to_end_bytes_over_floor": (visit_bytes / a.steps + 17.0 * cells_total / N)
                                           / (FLOOR_BYTES_PER_CELL * cells_total / N),
            "traffic": traffic,
            "counter_to_algorithmic": traffic / (visit_bytes / launches) if traffic else None,
            "launches": kt["launches"], "ms_total": kt["ms"],
            "bytes_per_launch": visit_bytes / launches,
            "avg_launch_ms": kt["ms"] / launches,
            "note": "rank 0.  achieved/frac: algorithmic bytes of the tile visits (12 B per cell "
                    "of a writing visit, 8 B of an unchanged one) over the kernel's HIP-event "
                    "time; useful_frac: 8 B per raster cell over the whole step.  "
                    "end_to_end_bytes_over_floor adds the start values (hub start: Z in, d out, "
                    "8 B/cell) and the certifying pass (9 B/cell)"})
        roofline.update(traffic_meta)
        certify_gbs = 9 * kr["units"] / max(kr["ms"], 1e-9) / 1e6
        init_gbs = 8 * ki["units"] / max(ki["ms"], 1e-9) / 1e6
        size_name = f"{S}^2" if N == 1 else f"{H}x{Wd}"
        hub_gbs = 8 * kh["units"] / max(kh["ms"], 1e-9) / 1e6
        named = {4: "BASELINE configs[3]: 32768x32768 DEM, row-block decomposed, 4 GPUs",
                 5: "BASELINE configs[4]: 65536x65536 mosaic, 8 GPUs"}.get(a.config)
        out = {
            "metric": f"Mcells/s sink-fill+D8 on {size_name} float32 DEM",
            "value": cells_total * a.steps / elapsed / 1e6,
            "unit": "Mcells/s",
            "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if a.config and N != {4: 4, 5: 8}[a.config] else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (named + ": " if named else "") +
                                   f"{H}x{Wd} float32 synthetic DEM (SURVEY 8d 'rough': "
                                   f"plane + 4 sinusoids + 0.5 m noise + 0.1% pits), "
                                   f"SinkFill eps=0 to exact convergence + D8; "
                                   f"{H // N} rows per GPU, row-block partition",
                       "rows_per_gpu": H // N, "cols": Wd,
                       "tile_visits_per_step": last.get("tile_visits"),
                       "tiles": last.get("tiles"),
                       "visits_per_tile": (last.get("tile_visits") or 0) / max(last.get("tiles") or 1, 1),
                       "visits_unchanged": last.get("visits_unchanged"),
                       "visits_flat": last.get("visits_flat"),
                       "certifying_rounds": last.get("rounds", 0),
                       "halo_exchanges": last.get("exchanges", 0),
                       "verifications": last.get("verifications")},
            "roofline": roofline,
            "kernels": {"certify_d8_kernel": dict(
                            roofs(certify_gbs, copy_gbs), launches=kr["launches"],
                            ms_total=kr["ms"], algorithmic_bytes_per_cell=9,
                            note="certifying pass behind the asynchronous launch, one stream "
                                 "over the raster: reads Z and W, writes the D8 codes; rounds of "
                                 "tile visits (fill_round_kernel) only if it finds a cell to "
                                 "lower -- then they are counted here too"),
                        "fill_init_kernel": dict(
                            roofs(init_gbs, copy_gbs), algorithmic_bytes_per_cell=8,
                            ms_per_step=ki["ms"] / a.steps,
                            note="start values of the fine raster (and of the coarse one: two "
                                 "launches per step)"),
                        "hub_dist_kernel": dict(
                            roofs(hub_gbs, copy_gbs), algorithmic_bytes_per_cell=8,
                            ms_per_step=kh["ms"] / a.steps,
                            note="hub start: per tile the minimax path cost of every cell to "
                                 "the tile's hub (Z in, d out) + the crossings between hubs; "
                                 "three rounds of directional scans per tile; waits on its loads and stores (its time does not depend on the number of rounds), tiles in XCD-aware order"),
                        "start_raster_solve": {
                            "blockmax_avg_launch_ms": kb["ms"] / max(kb["launches"], 1),
                            "fill_async_kernel<false, 1>_avg_launch_ms":
                                kc["ms"] / max(kc["launches"], 1),
                            "note": "fill of the hub raster (one node per 62x62 tile and the "
                                    "crossings between them, 1/930 of the cells; N > 1: of the "
                                    "16x16 block maxima): latency-bound"}},
        }
        if per_rank:
            out["per_rank"] = per_rank
        if N == 1 and not a.no_filters:
            out["filters"] = filter_paths(B, ctx, zd, wd, S, copy_gbs, bool(a.cpu_sample))
            out["config2"] = config2(B, ctx)
            out["config1_raster"] = config1(B, ctx)
            out["batch_of_tiles"] = batch_line(B, ctx)
        if a.cpu_sample and N == 1:
            out["cpu_baseline"] = cpu_baseline(z, a.cpu_sample)
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
