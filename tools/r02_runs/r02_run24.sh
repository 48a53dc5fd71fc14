#!/bin/bash
# LDS-staged advection tiles: parity + threads-per-block A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run24
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "advection_kernels or host_advect or operators_vs_oracle or golden or automatic_advection or irregular or randomised" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
cat > /tmp/adv_ab.py <<'PY'
import sys, os, numpy as np
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
del i, j
col = synthetic_color(n, 0, n)
rng = np.random.default_rng(1)
with sfl.Solver(n, n) as s:
    s.upload(capi.FIELD_COLOR, col)
    s.upload(capi.FIELD_PRESSURE, rng.standard_normal((n, n)).astype(np.float32))
    for name, v in fields.items():
        for k in [int(a) for a in sys.argv[1:]]:
            s.set_option(capi.OPT_ADVECT_KERNEL, k)
            res = {}
            ops = (("advect_velocity", lambda: s.advect_velocity(dt, True)), ("advect_color", lambda: s.advect_color(dt, False)),
                   ("step", lambda: s.step(dt, 1.0, 2, 1.96)))
            for op, fn in ops:
                best = 1e9
                for rep in range(4):
                    s.upload(capi.FIELD_VELOCITY, v)
                    fn(); s.synchronize()
                    s.upload(capi.FIELD_VELOCITY, v)
                    s.timer_start(); fn(); best = min(best, s.timer_stop())
                res[op] = best * 1e3
            print(f"{name:8s} kernel {k} threads {os.environ.get('SFL_ADVECT_THREADS', 'default')}: " + "  ".join(f"{a} {b:7.1f} us" for a, b in res.items()), flush=True)
PY
python /tmp/adv_ab.py 1 2
SFL_ADVECT_THREADS=256 python /tmp/adv_ab.py 2
SFL_ADVECT_THREADS=512 python /tmp/adv_ab.py 2
for k in 1 2; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 --sim-steps 3 --advect-kernel $k > $O/bench_k$k.json 2>> $O/bench.err
  python - <<PY
import json
d = json.load(open("$O/bench_k$k.json"))
print("kernel $k sim steps/s", d["sim_steps_per_sec"], {a: round(b["us"], 1) for a, b in d["sim_step_per_operator"].items()})
PY
done
