#!/bin/bash
set -u
export TMPDIR=/tmp
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "operators_vs_oracle or advection_kernels or golden or irregular" ) > gpurun_out/r02_run43_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r02_run43_pytest.log | tail -1
python - <<PY
import sys, numpy as np
sys.path.insert(0, ".")
import importlib
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from bench import synthetic_velocity
capi = sfl.capi
for n, m in ((8192, 8192), (8192, 1024), (2048, 2048)):
    v = synthetic_velocity(n, 0, m)
    with sfl.Solver(n, m) as s:
        s.upload(capi.FIELD_VELOCITY, v)
        s.upload(capi.FIELD_PRESSURE, np.zeros((m, n), np.float32))
        for rep in range(2):
            for name, fn in (("divergence", lambda: s.calculate_divergence(1.0)), ("gradient", lambda: s.subtract_gradient(1.0))):
                fn(); s.synchronize(); best = 1e9
                for _ in range(8):
                    s.timer_start(); fn(); best = min(best, s.timer_stop())
                print(n, m, name, round(best * 1e3, 1), "us", flush=True)
PY
