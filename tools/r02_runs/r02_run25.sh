#!/bin/bash
# fused velocity advection + divergence: parity and step time A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run25
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "advection_kernels or fused_advection or host_advect or golden or forces or step or long_run or randomised" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
cat > /tmp/step_ab.py <<'PY'
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
    for k, fd in ((1, 0), (2, 0), (2, 1), (1, 0), (2, 0), (2, 1)):
        s.set_option(capi.OPT_ADVECT_KERNEL, k)
        s.set_option(capi.OPT_FUSE_DIVERGENCE, fd)
        for iters in (2, 80):
            for _ in range(3): s.step(dt, 1.0, iters, 1.96)
            s.synchronize()
            s.timer_start()
            for _ in range(5): s.step(dt, 1.0, iters, 1.96)
            ms = s.timer_stop() / 5
            print(f"advect kernel {k} fused divergence {fd} iters {iters:2d}: {ms * 1e3:8.1f} us per step ({1e3 / ms:6.1f} steps/s)", flush=True)
PY
python /tmp/step_ab.py
