"""Per-phase wall time of the distributed fill in the two-rank rehearsal (exploration)."""
import os, sys, time, socket
import numpy as np
import torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.getcwd())
import hdem_synth

def worker(rank, world, port, S):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    from hydrodem_amd import partition as P, backend as B
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    H = world * S
    ghost = P.ghost_rows(world, H)
    g0, g1, _, _ = P.local_range(rank, world, H, ghost)
    zt = torch.from_numpy(hdem_synth.synth_dem(H, S, row0=g0, rows=g1 - g0)).cuda()
    wt = torch.empty_like(zt); dt = torch.empty(zt.shape, dtype=torch.uint8, device=zt.device)
    solver = P.HipLocalSolver(0); comm = P.DistComm()
    # instrument
    marks = []
    def timed(name, fn):
        def wrap(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
            marks.append((name, (time.perf_counter() - t) * 1e3)); return r
        return wrap
    P.coarse_start = timed("coarse_start", P.coarse_start)
    P._exchange_and_vote = timed("exchange+vote", P._exchange_and_vote)
    solver.fill = timed("fill", solver.fill)
    for rep in range(4):
        del marks[:]
        torch.cuda.synchronize(); dist.barrier(); t = time.perf_counter()
        P.sinkfill_distributed(zt, rank, world, solver, w_out=wt, d8_out=dt, ghost=ghost, comm=comm)
        torch.cuda.synchronize(); tot = (time.perf_counter() - t) * 1e3
        if rank == 0 and rep >= 2:
            print(f"total {tot:.2f} ms: " + ", ".join(f"{n} {ms:.2f}" for n, ms in marks))
    dist.destroy_process_group()

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, S), nprocs=2, join=True)
