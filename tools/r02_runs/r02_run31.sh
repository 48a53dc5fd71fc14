#!/bin/bash
# non-temporal stores in the tiled advection kernels: step A/B
set -u
export TMPDIR=/tmp
cat > /tmp/step_nt.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, ".")
import importlib
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from bench import synthetic_velocity, synthetic_color
capi = sfl.capi
n = 8192
dt = np.float32(1/30)
v = synthetic_velocity(n, 0, n)
col = synthetic_color(n, 0, n)
with sfl.Solver(n, n) as s:
    s.upload(capi.FIELD_COLOR, col)
    s.upload(capi.FIELD_VELOCITY, v)
    for _ in range(6): s.step(dt, 1.0, 80, 1.96)
    s.synchronize()
    for rep in range(2):
        for iters in (2, 80):
            for _ in range(3): s.step(dt, 1.0, iters, 1.96)
            s.synchronize()
            s.timer_start()
            for _ in range(6): s.step(dt, 1.0, iters, 1.96)
            ms = s.timer_stop() / 6
            print(f"SFL_ADV_NT={os.environ.get('SFL_ADV_NT', '0')} iters {iters:2d}: {ms * 1e3:8.1f} us per step ({1e3 / ms:6.1f} steps/s)", flush=True)
        for name, fn in (("advect_velocity", lambda: s.advect_velocity(dt, True)), ("advect_color", lambda: s.advect_color(dt, False))):
            fn(); s.synchronize(); best = 1e9
            for _ in range(4):
                s.timer_start(); fn(); best = min(best, s.timer_stop())
            print(f"   {name}: {best * 1e3:7.1f} us")
PY
for nt in 0 1 3 7 0 1 7; do SFL_ADV_NT=$nt python /tmp/step_nt.py; done
