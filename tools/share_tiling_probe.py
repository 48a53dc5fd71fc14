"""would TALLER tiles pay on the thin share of BASELINE configuration 4 (8192 x 1024 rows per GPU) if the
occupancy they cost came back from CONCURRENCY -- two launches in flight on two streams, each over one half of the columns?

An UPPER BOUND of the idea, measured with the product's own kernel and no new code: the two column halves as two independent
emulated ranks (4096 columns each, rank 3 of 8, self-copies as halo messages), their solves issued alternately so that both
streams always hold work -- nothing couples the halves at the seam between them, which a real split would have to (a
cross-stream dependency per launch).  Rows per tile 0 (automatic: 43 kept of 63 streamed at two waves per SIMD) against
taller tiles.  Prints ms per solve of the PAIR (= of the whole 8192-wide share) beside the one-context share.

    python tools/share_tiling_probe.py [solves]
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ITERS, OMEGA = 80, np.float32(1.96)


def share(dim_x, rows_per_tile, halo=64):
    s = sfl.Solver(dim_x, 8192, 0, 3, 8)
    s.comm_emulate()
    s.set_option(capi.OPT_SOR_HALO, halo)      # (a fixed depth: every variant runs the same plan)
    if rows_per_tile:
        s.set_option(capi.OPT_SOR_ROWS, rows_per_tile)
    rng = np.random.default_rng(dim_x)
    s.upload(capi.FIELD_DIVERGENCE, (rng.standard_normal((s.row_end - s.row_begin, dim_x)) * 0.1).astype(np.float32))
    return s


def timed(ctxs, n):
    for _ in range(15):
        for c in ctxs:
            c.poisson_solve(1.0, ITERS, OMEGA)
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for c in ctxs:
            c.poisson_solve(1.0, ITERS, OMEGA)
    for c in ctxs:
        c.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"# 8192 x 1024 share (rank 3 of 8, 80 iterations, halo 64, in-time exchanges as self-copies), {solves} solves each, ms per solve of the whole share")
for rep in range(2):
    for rows in (0, 64, 86, 128):
        one = share(8192, rows)
        t_one = timed([one], solves)
        launches = one.last_solve_info()["launches"]
        one.close()
        halves = [share(4096, rows), share(4096, rows)]
        t_pair = timed(halves, solves)
        t_half_alone = timed(halves[:1], solves)
        for h in halves:
            h.close()
        print(f"rep {rep}  rows per tile {rows if rows else 'auto':>4}: one context {t_one:.4f} ms ({launches} launches) | two column halves side by side "
              f"{t_pair:.4f} ms | one half alone {t_half_alone:.4f} ms", flush=True)
