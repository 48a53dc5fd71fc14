#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run7
mkdir -p $O
python - <<'PY'
import importlib, numpy as np, sys
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from oracle import loader
orc = loader.port()
om = np.float32(1.96)
for dim_x, dim_y in ((640, 500), (2048, 700)):
    d = (np.random.default_rng(3).standard_normal((dim_y, dim_x)) * 0.1).astype(np.float32)
    for fuse, iters in ((18, 9), (20, 23), (22, 11), (22, 25), (24, 12), (24, 30)):
        with sfl.Solver(dim_x, dim_y) as s:
            s.set_option(sfl.capi.OPT_SOR_FUSE, fuse)
            s.set_option(sfl.capi.OPT_SOR_LANE_CELLS, 2)
            s.upload(sfl.capi.FIELD_DIVERGENCE, d)
            s.poisson_solve(1.0, iters, om); s.synchronize()
            info = s.last_solve_info()
            ok = np.array_equal(s.download(sfl.capi.FIELD_PRESSURE).view(np.uint32), orc.poisson_solve(d, 1.0, iters, om).view(np.uint32))
        print("parity", dim_x, dim_y, fuse, iters, info, "OK" if ok else "MISMATCH")
PY
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5"
run() { # name args...
  n=$1; shift
  $B "$@" > $O/$n.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-34s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch  %.3e" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"], d["value"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for fuse in 16 18 20 22 24; do run full_l2_f$fuse --lane-cells 2 --fuse $fuse; done
for fuse in 16 20 24; do run slab1024_l2_f$fuse --dim-y 1024 --lane-cells 2 --fuse $fuse; run slab2048_l2_f$fuse --dim-y 2048 --lane-cells 2 --fuse $fuse; done
for fuse in 20 24; do run c5_16384_l2_f$fuse --size 16384 --iters 200 --steps 5 --warmup 2 --lane-cells 2 --fuse $fuse; done
run c5_16384_l2_f16 --size 16384 --iters 200 --steps 5 --warmup 2 --lane-cells 2 --fuse 16
