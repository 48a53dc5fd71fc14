#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run5
mkdir -p $O
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
run full_l2_f16 --lane-cells 2 --fuse 16
for fuse in 14 16 18 20; do run full_l22_f$fuse --lane-cells 22 --fuse $fuse; done
for rows in 100 130 150 170 190 220; do run full_l22_f16_r$rows --lane-cells 22 --fuse 16 --sor-rows $rows; done
for rows in 130 170 220; do run full_l22_f20_r$rows --lane-cells 22 --fuse 20 --sor-rows $rows; done
for fuse in 12 14 16; do run slab1024_l22_f$fuse --dim-y 1024 --lane-cells 22 --fuse $fuse; done
for rows in 20 24 30 40; do run slab1024_l22_f16_r$rows --dim-y 1024 --lane-cells 22 --fuse 16 --sor-rows $rows; done
