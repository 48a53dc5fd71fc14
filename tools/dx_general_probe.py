"""ms per 80-iteration solve at 8192^2 with dx == 1 (kernels without the dx * d product) and dx != 1 (the product formed once
per row as it enters the rhs ring: csrc/sor_stream_core.h iterate).   python tools/dx_general_probe.py [lib]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    capi = importlib.import_module("esp32-fluid-simulation_amd._capi")
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi
n = 8192
s = sfl.Solver(n, n)
rng = np.random.default_rng(1)
s.upload(capi.FIELD_DIVERGENCE, (rng.standard_normal((n, n)) * 0.1).astype(np.float32))
for rep in range(2):
    for dx in (1.0, 0.5):
        for _ in range(40):
            s.poisson_solve(dx, 80, np.float32(1.96))
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            s.poisson_solve(dx, 80, np.float32(1.96))
        s.synchronize()
        print(f"dx {dx}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per solve", flush=True)
