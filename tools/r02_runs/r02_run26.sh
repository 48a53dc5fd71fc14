#!/bin/bash
# advection tile variants (tile height, window margin, XCD remap): operator and step times
set -u
export TMPDIR=/tmp
cat > /tmp/adv_var.py <<'PY'
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
    s.set_option(capi.OPT_ADVECT_KERNEL, 2)
    for name, v in fields.items():
        res = {}
        ops = (("advect_velocity", lambda: s.advect_velocity(dt, True)), ("advect_color", lambda: s.advect_color(dt, False)),
               ("step2", lambda: s.step(dt, 1.0, 2, 1.96)))
        for op, fn in ops:
            best = 1e9
            for rep in range(4):
                s.upload(capi.FIELD_VELOCITY, v)
                fn(); s.synchronize()
                s.upload(capi.FIELD_VELOCITY, v)
                s.timer_start(); fn(); best = min(best, s.timer_stop())
            res[op] = best * 1e3
        print(f"{sys.argv[1]:8s} {name:8s}: " + "  ".join(f"{a} {b:7.1f} us" for a, b in res.items()), flush=True)
PY
python /tmp/adv_var.py default
for n in ty16 ty64 r2 r6 noremap; do SFL_LIB=$PWD/esp32-fluid-simulation_amd/lib/variants/libsfl_hip_$n.so python /tmp/adv_var.py $n; done
python /tmp/adv_var.py default
