#!/bin/bash
# LDS-staged advection tiles: parity + A/B against the one-thread-per-cell kernels
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run23
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "advection_kernels or host_advect or operators_vs_oracle or golden or automatic_advection or irregular or randomised" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import importlib
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from bench import synthetic_velocity, synthetic_color
capi = sfl.capi
n = 8192
dt = np.float32(1/30)
j, i = np.mgrid[0:n, 0:n].astype(np.float32)
fields = {"noise": synthetic_velocity(n, 0, n)}
sm = np.empty((n, n, 2), np.float32)
sm[..., 0] = 100 * (-(j - n/2) / n); sm[..., 1] = 100 * ((i - n/2) / n)
fields["vortex"] = sm
fields["zero"] = np.zeros((n, n, 2), np.float32)
del i, j
col = synthetic_color(n, 0, n)
with sfl.Solver(n, n) as s:
    s.upload(capi.FIELD_COLOR, col)
    s.upload(capi.FIELD_PRESSURE, np.zeros((n, n), np.float32))
    for name, v in fields.items():
        for k in (1, 2):
            s.set_option(capi.OPT_ADVECT_KERNEL, k)
            res = {}
            for op, fn in (("advect_velocity", lambda: s.advect_velocity(dt, True)), ("advect_color", lambda: s.advect_color(dt, False))):
                best = 1e9
                for rep in range(5):
                    s.upload(capi.FIELD_VELOCITY, v)
                    fn(); s.synchronize()
                    s.upload(capi.FIELD_VELOCITY, v)
                    s.timer_start(); fn(); best = min(best, s.timer_stop())
                res[op] = best * 1e3
            print(f"{name:8s} kernel {k}: advect_velocity {res['advect_velocity']:7.1f} us   advect_color {res['advect_color']:7.1f} us", flush=True)
PY
for k in 1 2; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 --sim-steps 3 --advect-kernel $k > $O/bench_k$k.json 2>> $O/bench.err
  python - <<PY
import json
d = json.load(open("$O/bench_k$k.json"))
print("kernel $k sim steps/s", d["sim_steps_per_sec"], {a: round(b["us"], 1) for a, b in d["sim_step_per_operator"].items()})
PY
done
