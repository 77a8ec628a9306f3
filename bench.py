#!/usr/bin/env python3
"""
bench.py -- the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S]

Metric (BASELINE.json): Mcells/s of SinkFill (to convergence) + D8FlowDirection
on a 16384^2 float32 DEM, inputs resident in HBM when the timed region starts,
outputs left in HBM.  One "step" = one full sink fill + D8 of the raster.
N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the raster
is N*S rows x S columns, row-block partitioned, S rows per rank -- weak
scaling -- with halo exchange between local solves (hydrodem_amd/partition.py).

One JSON line on stdout (rank 0) with, besides the contract keys:
  roofline      dominant kernel = fill_async_kernel (the sink-fill tile
                relaxation): algorithmic bytes (12 B per cell of every tile visit:
                Z in, W in, W out) / its HIP-event time over the timed steps, vs
                8 TB/s HBM peak;
  kernels       the certifying pass of the fill (which also writes the D8 codes: 9 B per
                cell), the init kernel, the coarse pre-solve;
  filters       the other operators of the scope table on the same raster (outside the
                timed region);
  cpu_baseline  the NumPy oracle (sink fill Jacobi to convergence + D8,
                1 thread) on a bounded crop of the same DEM, same host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FILL_BYTES_PER_CELL = 12    # per tile visit: Z in + W in + W out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--size", type=int, default=16384)
    p.add_argument("--cpu-sample", type=int, default=2048,
                   help="edge of the crop the CPU oracle is timed on (0 = skip)")
    return p.parse_args()


def profiled_traffic(kernel_substring):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes of this
    same command (profiles/r01_bench_pmc_*.csv; FETCH_SIZE and WRITE_SIZE need separate
    passes and cannot be collected from inside this process).  FETCH_SIZE under-reports
    on gfx950: x1.605 is the factor measured on a kernel with the same 4-byte-per-lane
    row loads that reads a known 1.0737 GB.  None when no profile exists."""
    import csv
    out = {}
    for key, name in (("fetch", "r01_bench_pmc_fetch_size.csv"), ("write", "r01_bench_pmc_write_size.csv")):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None
        for row in csv.DictReader(open(path)):
            if kernel_substring in row["Kernel_Name"]:
                out[key] = float(row["Per_Dispatch"]) * 1024.0
    if "fetch" not in out or "write" not in out:
        return None
    return out["fetch"] * 1.605 + out["write"]


def cpu_baseline(z_crop):
    """NumPy oracle (the 'NumPy CPU reference' of north_star) on a crop of the
    workload; checker code timed as a baseline, never used as product."""
    import oracle
    from oracle import c_oracle
    n = z_crop.size
    t = time.perf_counter()
    w, sweeps = oracle.sinkfill_jacobi(z_crop)
    oracle.d8_flow_direction(w)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    w2 = c_oracle.sinkfill_pflood(z_crop)
    c_oracle.d8(w2)
    dt_c = time.perf_counter() - t
    assert np.array_equal(w, w2)
    return {"value": n / dt / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": f"{z_crop.shape[0]}x{z_crop.shape[1]} crop (rows/cols 0..) of the "
                      f"workload DEM; NumPy Jacobi sink fill to convergence "
                      f"({sweeps} sweeps) + NumPy D8, single thread, {dt:.1f} s",
            "host_cores": os.cpu_count(),
            "c_priority_flood": {"value": n / dt_c / 1e6, "unit": "Mcells/s", "cores": 1,
                                 "seconds": dt_c,
                                 "note": "same crop, C priority-flood oracle + C D8 "
                                         "(a better CPU algorithm than the NumPy path)"}}


def filter_paths(B, ctx, zd, scratch, S, reps=3):
    """The other operators of the scope table on the same raster, outside the timed
    region (they are not part of the headline metric): warm call, then ``reps`` timed."""
    import hdem_synth
    res = {}

    def timed(fn):
        fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        return {"ms": ms, "Mcells_per_s": S * S / ms / 1e3}

    mask = B.DeviceRaster.from_host(hdem_synth.synth_groves(S, S), ctx=ctx)
    pong = B.DeviceRaster.empty(zd.shape, np.float32, ctx)
    res["groves_x3"] = dict(timed(lambda: B.groves_dev(zd, mask, iterations=3, out=scratch,
                                                       scratch=pong)),
                            algorithmic_bytes_per_cell=27)
    # BASELINE configs[2]: the full chain -- groves x3, sink fill, D8 -- device resident
    filled = B.DeviceRaster.empty(zd.shape, np.float32, ctx)
    codes = B.DeviceRaster.empty(zd.shape, np.uint8, ctx)

    def chain():
        B.groves_dev(zd, mask, iterations=3, out=scratch, scratch=pong)
        B.sinkfill_d8_dev(scratch, out=filled, codes=codes)
    res["full_chain_groves_fill_d8"] = timed(chain)
    res["d8_alone"] = dict(timed(lambda: B.d8_dev(filled, out=codes)),
                           algorithmic_bytes_per_cell=5)
    for r in (mask, pong, filled, codes):
        r.free()
    res["boxmean3_round"] = dict(timed(lambda: B.boxmean3_dev(zd, out=scratch)),
                                 algorithmic_bytes_per_cell=8)
    ctx.profile(True)
    ctx.profile_reset()
    res["fourier_destripe"] = timed(lambda: B.fourier_destripe_dev(zd, out=scratch))
    n_calls = reps + 1
    for name, kid in (("rocfft_c2c", B.K_FFT), ("detect", B.K_FOURIER_DETECT),
                      ("mask", B.K_FOURIER_MASK),
                      ("pointwise", B.K_FOURIER_POINT)):
        res["fourier_destripe"][name + "_ms"] = ctx.profile_get(kid)["ms"] / n_calls
    ctx.profile(False)
    # SURVEY 8d's second input variant: the same DEM in integer metres (large flats, ties)
    hs = np.round(zd.to_host())
    hd_ = B.DeviceRaster.from_host(hs, ctx=ctx)
    codes = B.DeviceRaster.empty(zd.shape, np.uint8, ctx)
    info = {}
    res["sinkfill_d8_srtm_variant"] = timed(
        lambda: info.update(B.sinkfill_d8_dev(hd_, out=scratch, codes=codes)[2]))
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
    res["lagoons_detection"] = timed(lagoons)
    res["lagoons_detection"]["majority_ms"] = ctx.profile_get(B.K_MAJORITY)["ms"] / n_calls
    res["lagoons_detection"]["other_kernels_ms"] = ctx.profile_get(B.K_LAGOON)["ms"] / n_calls
    ctx.profile(False)
    for r in (hd_, l_fixed, l_mask):
        r.free()
    return res


def main():
    a = parse()
    S, N = a.size, a.gpus
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if N > 1 and world != N:
        raise SystemExit(f"--gpus {N} needs torch.distributed.run with {N} ranks "
                         f"(WORLD_SIZE={world})")

    import hdem_synth               # inputs; oracle/ is only touched by cpu_baseline()
    from hydrodem_amd import backend as B

    if N == 1:
        ctx = B.context(0)
        z = hdem_synth.synth_dem(S, S)
        zd = B.DeviceRaster.from_host(z, ctx=ctx)
        wd = B.DeviceRaster.empty(z.shape, np.float32, ctx)
        dd = B.DeviceRaster.empty(z.shape, np.uint8, ctx)
        info = {}

        def step():
            # fill + D8 in one call: the certifying pass of the fill writes the codes
            _, _, st = B.sinkfill_d8_dev(zd, out=wd, codes=dd)
            info.update(st)

        def sync():
            ctx.synchronize()

        def reduce_max(x):
            return x
    else:
        import torch
        import torch.distributed as dist
        from hydrodem_amd import partition as P
        # HDEM_REHEARSE=1: every rank on cuda:0 over gloo -- lets the N > 1 code path
        # be exercised on a one-GPU box; never a performance number
        rehearse = os.environ.get("HDEM_REHEARSE") == "1"
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if rehearse else "nccl")
        H = N * S
        ghost = P.ghost_rows(world, H)                  # one tile row of overlap per seam
        g0, g1, top, bottom = P.local_range(rank, world, H, ghost)
        z = hdem_synth.synth_dem(H, S, row0=g0, rows=g1 - g0)
        dev = torch.device("cuda", local_rank)
        zt = torch.from_numpy(z).to(dev)
        wt = torch.empty_like(zt)
        dt_ = torch.empty(zt.shape, dtype=torch.uint8, device=dev)
        solver = P.HipLocalSolver(local_rank)
        ctx = solver.ctx
        info = {}

        def step():
            # (the last verifying pass of the fill writes the D8 codes of the block)
            _, st = P.sinkfill_distributed(zt, rank, world, solver, w_out=wt, d8_out=dt_,
                                           ghost=ghost)
            info.update(st)

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
    ctx.profile(False)

    if rank == 0:
        cells_total = N * S * S
        ms_per_step = elapsed / a.steps * 1e3
        fill_gbs = FILL_BYTES_PER_CELL * kt["units"] / max(kt["ms"], 1e-9) / 1e6
        out = {
            "metric": "Mcells/s sink-fill+D8 on 16384^2 float32 DEM",
            "value": cells_total * a.steps / elapsed / 1e6,
            "unit": "Mcells/s",
            "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{N * S}x{S} float32 synthetic DEM (SURVEY 8d 'rough': "
                                   f"plane + 4 sinusoids + 0.5 m noise + 0.1% pits), "
                                   f"SinkFill eps=0 to exact convergence + D8; "
                                   f"{S} rows per GPU, row-block partition",
                       "rows_per_gpu": S, "cols": S,
                       "tile_visits_per_step": info.get("tile_visits"),
                       "tiles": info.get("tiles"),
                       "visits_unchanged": info.get("visits_unchanged"),
                       "certifying_rounds": info.get("rounds"),
                       "halo_exchanges": info.get("exchanges", 0)},
            "roofline": {"bound": "hbm", "kernel": "fill_async_kernel<false, 0>",
                         "achieved": fill_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": fill_gbs / HBM_PEAK_GBS,
                         "traffic": profiled_traffic("fill_async_kernel<false, 0>")
                         if (N == 1 and S == 16384) else None,
                         "traffic_source": "profiles/r01_bench_pmc_{fetch,write}_size.csv: separate "
                                           "rocprofv3 --pmc passes of this command, bytes per launch, "
                                           "FETCH_SIZE x1.605 (gfx950 calibration)",
                         "launches": kt["launches"], "ms_total": kt["ms"],
                         "bytes_per_launch": FILL_BYTES_PER_CELL * kt["units"]
                         / max(kt["launches"], 1),
                         "avg_launch_ms": kt["ms"] / max(kt["launches"], 1),
                         "note": "rank 0; algorithmic 12 B per cell of every tile visit"},
            "kernels": {"certify_d8_kernel": {
                            "achieved": (FILL_BYTES_PER_CELL - 4 + 1) * kr["units"]
                            / max(kr["ms"], 1e-9) / 1e6,
                            "unit": "GB/s", "launches": kr["launches"], "ms_total": kr["ms"],
                            "note": "certifying pass behind the asynchronous launch, one stream "
                                    "over the raster: reads Z and W, writes the D8 codes (1 B per "
                                    "cell); rounds of tile visits (fill_round_kernel) only if it "
                                    "finds a cell to lower -- then they are counted here too"},
                        "fill_init_kernel": {"avg_launch_ms": ki["ms"] / max(ki["launches"], 1)},
                        "coarse_pre_solve": {
                            "blockmax_avg_launch_ms": kb["ms"] / max(kb["launches"], 1),
                            "fill_async_kernel<false, 1>_avg_launch_ms":
                                kc["ms"] / max(kc["launches"], 1),
                            "note": "fill of the 16x16 block maxima (1/256 of the cells): start "
                                    "values of the fine solve; latency-bound"}},
        }
        if N == 1:
            out["filters"] = filter_paths(B, ctx, zd, wd, S)
        if a.cpu_sample and N == 1:
            c = min(a.cpu_sample, S)
            zc = np.ascontiguousarray(z[:c, :c])
            out["cpu_baseline"] = cpu_baseline(zc)
        print(json.dumps(out), flush=True)
    if N > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
