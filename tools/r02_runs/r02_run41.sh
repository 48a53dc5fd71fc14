#!/bin/bash
# small grids in one launch: parity (full GPU suite) and the sketch-size timings
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run41
mkdir -p $O
( python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -8
cat > /tmp/small_ab.py <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import importlib
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi
rng = np.random.default_rng(3)
for dim_x, dim_y, iters in ((61, 81, 10), (61, 81, 20), (80, 60, 20), (78, 78, 20), (64, 48, 20), (32, 32, 20)):
    v = rng.uniform(-50, 50, (dim_y, dim_x, 2)).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    with sfl.Solver(dim_x, dim_y) as s:
        s.upload(capi.FIELD_VELOCITY, v); s.upload(capi.FIELD_COLOR, c)
        out = []
        for small in (1, 0):
            s.set_option(capi.OPT_SMALL_GRID, small)
            for name, fn in (("step", lambda: s.step(np.float32(1 / 30), 1.0, iters, 1.96)), ("solve", lambda: s.poisson_solve(1.0, iters, 1.96))):
                for _ in range(20): fn()
                s.synchronize()
                s.timer_start()
                for _ in range(200): fn()
                out.append((small, name, s.timer_stop() / 200 * 1e3))
        print(f"{dim_x} x {dim_y}, {iters} iters: " + "  ".join(f"{'one launch' if a else 'general'} {b} {t:6.1f} us" for a, b, t in out), flush=True)
PY
python /tmp/small_ab.py
