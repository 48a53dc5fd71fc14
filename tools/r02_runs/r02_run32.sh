#!/bin/bash
# tiled divergence / gradient (stand-alone operators): parity and A/B against the row-streaming kernels
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run32
mkdir -p $O
( SFL_STENCIL_BASELINE=3 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "operators_vs_oracle or golden or virtual_slabs_match or irregular or randomised or projection or advection_kernels_on_slabs or long_run" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
cat > /tmp/fd_ab.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, ".")
import importlib
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from bench import synthetic_velocity
capi = sfl.capi
for n, m in ((8192, 8192), (8192, 1024), (2048, 2048), (8190, 4000)):
    v = synthetic_velocity(n, 0, m)
    rng = np.random.default_rng(1)
    with sfl.Solver(n, m) as s:
        s.upload(capi.FIELD_VELOCITY, v)
        s.upload(capi.FIELD_PRESSURE, rng.standard_normal((m, n)).astype(np.float32))
        res = {}
        for name, fn in (("divergence", lambda: s.calculate_divergence(1.0)), ("gradient", lambda: s.subtract_gradient(1.0))):
            fn(); s.synchronize(); best = 1e9
            for _ in range(6):
                s.timer_start(); fn(); best = min(best, s.timer_stop())
            res[name] = best * 1e3
        print(f"SFL_STENCIL_BASELINE={os.environ.get('SFL_STENCIL_BASELINE', '0')} {n} x {m}: " + "  ".join(f"{a} {b:7.1f} us" for a, b in res.items()), flush=True)
PY
for k in 0 3 1 0 3; do SFL_STENCIL_BASELINE=$k python /tmp/fd_ab.py; done
