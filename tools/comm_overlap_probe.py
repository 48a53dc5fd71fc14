#!/usr/bin/env python3
"""Can RCCL send/recv make progress WHILE the fused SOR kernel occupies the GPU?  (All tiles of a
launch are resident at once; an RCCL kernel needs a free CU slot.)  One GPU is enough to find out:
context A runs 80-iteration solves of a slab share (8192 x 1024) on its stream, context B -- a 1-rank
RCCL communicator -- sends 64-row halos to itself (sfl_comm_loopback: real ncclSend / ncclRecv, the
same pointer / count arithmetic as a neighbour exchange) on ITS stream.  Prints the time of each
alone and of both issued together: together ~ max(...) means the exchange hides behind compute."""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sfl = importlib.import_module("esp32-fluid-simulation_amd")
cap = sfl.capi
om = np.float32(1.96)
dim_x, rows, iters, reps, halo = 8192, 1024, 80, 20, 64

a = sfl.Solver(dim_x, rows)
a.upload(cap.FIELD_DIVERGENCE, (np.random.default_rng(1).standard_normal((rows, dim_x)) * 0.1).astype(np.float32))
b = sfl.Solver(dim_x, 256)
b.comm_attach(sfl.comm_unique_id())
b.upload(cap.FIELD_DIVERGENCE, np.ones((256, dim_x), np.float32))
b.upload(cap.FIELD_PRESSURE, np.zeros((256, dim_x), np.float32))


def solves():
    for _ in range(reps):
        a.poisson_solve(1.0, iters, om)


def halos():
    for _ in range(reps * 3):       # an 8-GPU solve exchanges three times
        b.comm_loopback(halo)


def timed(*fns):
    for f in fns:
        f()
    a.synchronize(); b.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    a.synchronize(); b.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    solves()
a.synchronize()
print(f"solve alone            : {timed(solves):.4f} ms per solve")
print(f"3 x {halo}-row self send/recv alone ({halo * dim_x * 4 / 1e6:.1f} MB each): {timed(halos):.4f} ms per solve-equivalent")
print(f"both, issued together  : {timed(solves, halos):.4f} ms")


def interleaved():
    for _ in range(reps):
        a.poisson_solve(1.0, iters, om)
        for _ in range(3):
            b.comm_loopback(halo)


print(f"both, interleaved issue: {timed(interleaved):.4f} ms")
