#!/usr/bin/env python3
"""Do two half-height problems on two streams finish sooner than one full-height problem on one
stream?  (Launches of one solve are dependent and run in lock step: every wave of a launch is in its
load-heavy prologue, then in its VALU-heavy steady state, at the same time.  Two independent launch
sequences on two streams can interleave those phases.)  Usage: concurrency_probe.py [rows] [fuse]"""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sfl = importlib.import_module("esp32-fluid-simulation_amd")
cap = sfl.capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fuse = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lane = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dim_x, iters, reps = 8192, 80, 30
om = np.float32(1.96)


def make(r):
    s = sfl.Solver(dim_x, r)
    s.set_option(cap.OPT_SOR_FUSE, fuse)
    s.set_option(cap.OPT_SOR_LANE_CELLS, lane)
    s.upload(cap.FIELD_DIVERGENCE, (np.random.default_rng(r).standard_normal((r, dim_x)) * 0.1).astype(np.float32))
    return s


def timed(solvers):
    for _ in range(10):
        for s in solvers:
            s.poisson_solve(1.0, iters, om)
    for s in solvers:
        s.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in solvers:
            s.poisson_solve(1.0, iters, om)
    for s in solvers:
        s.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


one = make(rows)
print(f"one context  8192 x {rows}: {timed([one]):.4f} ms per solve")
for parts in (2, 4):
    group = [make(rows // parts) for _ in range(parts)]
    print(f"{parts} contexts of 8192 x {rows // parts} on {parts} streams: {timed(group):.4f} ms per round of solves")
    for s in group:
        s.close()
