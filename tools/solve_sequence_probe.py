"""Per-solve duration of consecutive poisson_solve calls (8192^2 x 80) right after set-up:
how long the GPU takes to reach its steady rate."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
sfl = importlib.import_module("esp32-fluid-simulation_amd")
import bench
n = 8192
with sfl.Solver(n, n) as s:
    s.upload(sfl.capi.FIELD_VELOCITY, bench.synthetic_velocity(n, 0, n))
    s.calculate_divergence(1.0)
    s.synchronize()
    ts = []
    for k in range(60):
        s.timer_start(); s.poisson_solve(1.0, 80, np.float32(1.96)); ts.append(s.timer_stop())
    print("per-solve ms:", " ".join(f"{t:.2f}" for t in ts))
    # back-to-back batches without a sync in between
    for batch in (5, 10, 20, 50):
        time.sleep(0.5)
        s.timer_start()
        for _ in range(batch):
            s.poisson_solve(1.0, 80, np.float32(1.96))
        print(f"after 0.5 s idle, {batch} solves back to back: {s.timer_stop()/batch:.3f} ms per solve")
