"""Soak of ONE rank's slab program on one GPU with a chosen halo transport:

    python tools/soak_transport.py <seed> <seconds> <copy|rccl> [min_solves]

`rccl` = sfl_comm_emulate_rccl: every halo message a real ncclSend / ncclRecv of the rank to itself on a one-rank
communicator, through the branch a rank of a real communicator takes (the sender count in front of the ncclGroup, the
arrival count behind it); `copy` = sfl_comm_emulate (self-copies).  Random rank / group size / grid / depth / halo /
schedule; every solve's rows out of the cuts' reach are compared bit for bit with the same solve on a whole-domain
context, and sfl_synchronize must never report a wait that gave up.  Ref: the loop being sharded, poisson.cpp:121-124."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
transport = sys.argv[3] if len(sys.argv) > 3 else "rccl"
min_solves = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(seed)
t0, solves, bad, timeouts, configs = time.time(), 0, 0, 0, 0
by_schedule = {}
while time.time() - t0 < budget or solves < min_solves:
    nranks = int(rng.choice([2, 3, 4, 8]))
    rank = int(rng.integers(0, nranks))
    dim_x = int(rng.choice([512, 1024, 2048, 3000, 4096, 8192]))
    rows = int(rng.integers(160, 700)) if dim_x <= 4096 else int(rng.choice([256, 512, 1024]))
    dim_y = rows * nranks
    fuse = int(rng.choice([0, 4, 8, 10, 12, 16]))
    halo = int(rng.choice([0, 16, 32, 64]))
    if halo and fuse and halo < fuse:
        halo = 0
    iters = int(rng.integers(3, min(60, rows // 4)))   # rows out of the cuts' reach (2 * iters from each cut) must remain
    arrival = int(rng.choice([-1, -1, 1, 0]))
    overlap = int(rng.integers(0, 8) > 0)
    # the whole-domain solve of a right-hand side that is zero away from this slab: only rows within 2 * iters of it matter
    b, e = rank * rows, (rank + 1) * rows
    lo, hi = max(b - 2 * iters - 4, 0), min(e + 2 * iters + 4, dim_y)
    d = np.zeros((dim_y, dim_x), np.float32)
    d[lo:hi] = rng.standard_normal((hi - lo, dim_x)).astype(np.float32)
    with sfl.Solver(dim_x, dim_y) as one:
        if fuse:
            one.set_option(capi.OPT_SOR_FUSE, fuse)
        one.upload(capi.FIELD_DIVERGENCE, d)
        one.poisson_solve(1.0, iters, 1.96)
        one.synchronize()
        want = one.download(capi.FIELD_PRESSURE)[b:e]
    reach = 2 * iters
    inner = slice(reach if rank > 0 else 0, rows - (reach if rank < nranks - 1 else 0))
    with sfl.Solver(dim_x, dim_y, 0, rank, nranks) as s:
        if transport == "rccl":
            with sfl.stdout_to_stderr():      # RCCL's banner
                s.comm_emulate_rccl()
        else:
            s.comm_emulate()
        s.set_option(capi.OPT_SOR_FUSE, fuse)
        s.set_option(capi.OPT_SOR_HALO, halo)
        s.set_option(capi.OPT_EXCHANGE_SCHEDULE, 1 if not overlap else (3 if arrival else 2))
        sched = s.get_option(capi.OPT_EXCHANGE_SCHEDULE)
        s.upload(capi.FIELD_DIVERGENCE, d[b:e])
        configs += 1
        for rep in range(int(rng.integers(4, 40))):   # back to back: swapped buffers, counts that only go up
            s.poisson_solve(1.0, iters, 1.96)
            if rep % 3 == 2 or rep < 2:
                try:
                    s.synchronize()
                except sfl.SflError as err:
                    timeouts += 1
                    print(f"TIME-OUT {err}", flush=True)
                got = s.download(capi.FIELD_PRESSURE)
                if not np.array_equal(got[inner].view(np.uint32), want[inner].view(np.uint32)):
                    bad += 1
                    print(f"MISMATCH rank {rank}/{nranks} {dim_x}x{dim_y} iters {iters} fuse {fuse} halo {halo} arrival {arrival} "
                          f"overlap {overlap} rep {rep}: {int(np.count_nonzero(got[inner].view(np.uint32) != want[inner].view(np.uint32)))} cells",
                          flush=True)
            solves += 1
            by_schedule[sched] = by_schedule.get(sched, 0) + 1
        try:
            s.synchronize()
        except sfl.SflError as err:
            timeouts += 1
            print(f"TIME-OUT {err}", flush=True)
print(f"transport {transport}: {solves} solves in {configs} configurations in {time.time() - t0:.0f} s "
      f"(by schedule 1 in line / 2 early by events / 3 in time: {dict(sorted(by_schedule.items()))}): "
      f"{bad} mismatches, {timeouts} waits that gave up", flush=True)
sys.exit(1 if bad or timeouts else 0)
